"""FITC sparse GP and the Nystrom eigen-basis on the GPU, through the C ABI, against the vectors the
reference produced (tests/golden/make_golden.py:fitc_case) and the oracle's restatement."""
import numpy as np
import pytest

from oracle import gpexp_oracle as orc

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


def test_fitc_gp_matches_the_reference(golden):
    from gpExp.kernels import KernelSquaredExponential
    from gpExp.gp import GP
    from gpExp.gp_kernel_utilities import calculateCovarianceMatrixFITC
    c = "fitc"
    ix = golden.index[c]
    k = KernelSquaredExponential(list(ix["kernel"]["cl"]), ix["kernel"]["signalSize"], ix["kernel"]["d"])
    X, y, Z = golden(c, "X"), golden(c, "y"), golden(c, "Z")
    np.random.seed(ix["seed"])
    g = GP(k, ix["noise"], FITC=ix["fitc"])
    g.train(X, y)
    assert np.array_equal(g.fitcnodes, golden(c, "fitcnodes"))      # same permutation draw as the reference
    assert rel(g.coeff, golden(c, "coeff")) <= 1e-9
    mean, var = g.evaluate(Z, compvar=1)
    assert rel(mean, golden(c, "mean")) <= 1e-9
    assert rel(var, golden(c, "var")) <= 1e-9
    assert rel(g.evaluateVariance(Z), golden(c, "var_signed")) <= 1e-9
    assert rel(g.covarianceMatrix, golden(c, "cov")) <= 1e-11
    assert rel(g.precisionMatrix, golden(c, "prec")) <= 1e-9
    assert rel(g.evaluate(Z), golden(c, "mean")) <= 1e-9
    # a second operation on the same instance keeps the inducing points (the reference raises here)
    g.train(X, y)
    assert rel(g.coeff, golden(c, "coeff")) <= 1e-9
    np.random.seed(ix["seed"])
    g2 = GP(k, ix["noise"], FITC=ix["fitc"])
    assert g2.computeLogLike(X, y) == pytest.approx(float(golden(c, "loglike")), rel=1e-9)
    cov, prec, sn = calculateCovarianceMatrixFITC(k, X, ix["noise"], golden(c, "fitcnodes").copy(), returnCov=True)
    assert rel(cov, golden(c, "util_cov")) <= 1e-11 and rel(prec, golden(c, "util_prec")) <= 1e-9
    prec2, sn2 = calculateCovarianceMatrixFITC(k, X, ix["noise"], golden(c, "fitcnodes").copy())
    assert np.array_equal(prec2, prec) and np.array_equal(sn2, golden(c, "fitcnodes"))
    # posterior covariance (compvar=2) is consistent with the variances
    _, covz = g.evaluate(Z, compvar=2)
    assert rel(np.diag(covz), golden(c, "var_signed")) <= 1e-8


@pytest.mark.parametrize("n,nu,d,kind", [(700, 130, 3, "se"), (1500, 300, 2, "matern32"), (300, 257, 4, "se")])
def test_fitc_against_the_oracle_beyond_one_tile(n, nu, d, kind):
    from gpexp_amd import device as dev
    ctx = dev.context()
    rng = np.random.default_rng(n + nu)
    X = rng.uniform(-1, 1, (n, d))
    S = X[rng.permutation(n)[:nu]].copy()
    y = np.sin(X.sum(1)) + 0.1 * rng.standard_normal(n)
    Z = rng.uniform(-1, 1, (333, d))
    noise = 0.05
    if kind == "se":
        cl = list(0.6 + 0.1 * np.arange(d))
        sp, s = dev.KernelSpec(dev.K_SE, d, cl + [1.1]), dict(kind="se", cl=cl, signalSize=1.1, d=d)
    else:
        sp, s = dev.KernelSpec(dev.K_MATERN32, d, [0.9, 1.3]), dict(kind="matern32", rho=0.9, signalSize=1.3, d=d)
    Xd = dev.points(ctx, X)
    m = dev.FitcModel(ctx, sp, Xd, dev.points(ctx, S), noise)
    ref = orc.fitc_fit(s, X, y, noise, S)
    # Two references.  (a) The oracle follows the reference's dense formula -- pinv(Quu), inv(Quu + Kuf G^-1 Kfu) with a
    # condition number ~ N s / noise ~ 1e5-1e6 here, then the cancelling difference G^-1 - (...): good to 1e-6..1e-4 at
    # these sizes (parity with the reference itself is pinned at N=60 above, 1e-9).  (b) The same quantities from the dense
    # Q + G by a direct solve (cond ~ 1e4): what the Woodbury pieces must reproduce to working precision.
    cov_ref = ref["cov"]
    coeff_acc = np.linalg.solve(cov_ref, y)
    coeff, quad = m.solve(y)
    assert rel(coeff, coeff_acc) <= 1e-8 and rel(coeff, ref["coeff"]) <= 1e-3
    assert quad == pytest.approx(float(y @ coeff_acc), rel=1e-10)
    assert m.logdet() == pytest.approx(np.linalg.slogdet(cov_ref)[1], rel=1e-9, abs=1e-8)
    mean, var = m.posterior(coeff, dev.points(ctx, Z))
    kv = orc.cross_matrix(s, Z, X)
    var_acc = orc.kernel_diag(s, Z) - np.einsum("ij,ji->i", kv, np.linalg.solve(cov_ref, kv.T))
    assert rel(mean, kv @ coeff_acc) <= 1e-8 and rel(var, var_acc) <= 1e-8
    rmean, rvar = orc.posterior(s, ref, Z, compvar=1)
    assert rel(mean, rmean) <= 1e-3 and rel(var, rvar) <= 1e-3   # the dense formula's own round-off (see (a))
    cov, prec = m.dense()
    assert rel(cov, cov_ref) <= 1e-11 and rel(prec, ref["prec"]) <= 1e-3
    assert rel(prec @ cov_ref, np.eye(n)) <= 1e-7            # P (Q + G) = I up to the 1e-12 the reference adds to g
    assert -0.5 * quad - 0.5 * m.logdet() - n / 2 * np.log(2 * np.pi) == pytest.approx(
        -0.5 * float(y @ coeff_acc) - 0.5 * np.linalg.slogdet(cov_ref)[1] - n / 2 * np.log(2 * np.pi), rel=1e-10)


def test_fitc_at_a_size_where_the_woodbury_product_runs_in_slices():
    """N = 8192, nu = 1024: the nu x nu x N product of the fit (Quu + Kuf G^-1 Kfu) is a small C under a long k range and runs
    as slices of the k range (launch_gemm_ksplit, fitc.hip); Lu^-1 Kuf goes out of place through Lu's block inverses.  Against
    the dense Q + G solved directly on the host (the reference's own dense formula loses digits at this size, see above)."""
    import scipy.linalg as sl
    from gpexp_amd import device as dev
    ctx = dev.context()
    n, nu, d, noise = 8192, 1024, 4, 0.05
    rng = np.random.default_rng(n + nu)
    X = rng.uniform(-1, 1, (n, d))
    S = X[rng.permutation(n)[:nu]].copy()
    y = np.sin(X.sum(1)) + 0.1 * rng.standard_normal(n)
    cl = [0.6, 0.7, 0.8, 0.9]
    sp, s = dev.KernelSpec(dev.K_SE, d, cl + [1.1]), dict(kind="se", cl=cl, signalSize=1.1, d=d)
    Quu = orc.cov_matrix(s, S, noise, row_loop=False)
    Kuf = orc.cross_matrix(s, S, X)
    W = sl.solve_triangular(np.linalg.cholesky(Quu), Kuf, lower=True, check_finite=False)
    g = orc.kernel_diag(s, X) + noise - np.sum(W * W, axis=0)
    cov = W.T @ W
    cov[np.diag_indices(n)] += g
    m = dev.FitcModel(ctx, sp, dev.points(ctx, X), dev.points(ctx, S), noise)
    coeff, quad = m.solve(y)
    c = sl.cho_factor(cov, lower=True, check_finite=False)
    coeff_acc = sl.cho_solve(c, y, check_finite=False)
    assert rel(coeff, coeff_acc) <= 1e-7
    assert quad == pytest.approx(float(y @ coeff_acc), rel=1e-9)
    assert m.logdet() == pytest.approx(2.0 * float(np.sum(np.log(np.diag(c[0])))), rel=1e-9, abs=1e-7)
    Z = rng.uniform(-1, 1, (257, d))
    mean, var = m.posterior(coeff, dev.points(ctx, Z))
    kv = orc.cross_matrix(s, Z, X)
    assert rel(mean, kv @ coeff_acc) <= 1e-7
    assert rel(var, orc.kernel_diag(s, Z) - np.einsum("ij,ji->i", kv, sl.cho_solve(c, kv.T, check_finite=False))) <= 1e-7


def test_nystrom_basis_and_operator(golden):
    from gpExp.kernels import KernelSquaredExponential
    from gpExp.gp_kernel_utilities import calculateKernelBasisFunctionsMC, covTimesV, calculateCovarianceMatrix
    c = "fitc"
    ix = golden.index[c]
    k = KernelSquaredExponential(list(ix["kernel"]["cl"]), ix["kernel"]["signalSize"], ix["kernel"]["d"])
    mc = golden(c, "nys_mc")
    b = np.random.default_rng(2).standard_normal(len(mc))
    assert rel(covTimesV(b, k, mc), calculateCovarianceMatrix(k, mc, 0.0) @ b) <= 1e-13
    # the resident covariance is keyed on the CONTENT of the point set: an in-place edit and a fresh array of
    # the same shape (object ids are recycled) must both give the new operator
    mc2 = mc.copy()
    assert rel(covTimesV(b, k, mc2), calculateCovarianceMatrix(k, mc, 0.0) @ b) <= 1e-13
    mc2[:] = mc2[::-1] * 0.9
    assert rel(covTimesV(b, k, mc2), calculateCovarianceMatrix(k, mc2, 0.0) @ b) <= 1e-13
    for _ in range(3):
        fresh = np.random.default_rng(_).uniform(-1, 1, mc.shape)
        assert rel(covTimesV(b, k, fresh), calculateCovarianceMatrix(k, fresh, 0.0) @ b) <= 1e-13
        del fresh
    ev, evec = calculateKernelBasisFunctionsMC(k, 6, mc)
    assert rel(ev, golden(c, "nys_eigv")) <= 1e-9
    assert rel(np.abs(evec), np.abs(golden(c, "nys_eigve"))) <= 1e-6      # ARPACK eigenvectors, up to sign


def test_point_derivatives_on_a_fitc_model_match_the_reference(golden):
    """Round 6 (VERDICT r5 missing 4): the reference computes evaluateVarianceDerivWRTnewpt / evaluateVarianceDerivative from
    whatever `precisionMatrix` holds -- for a FITC model the Woodbury precision (gp.py:194-206, 275, 322).  Through
    gpx_fitc_var_grad_newpt / gpx_fitc_var_grad (beta = P K(X, Z) from the model's factors, no N x N matrix) against the
    reference's outputs on the `fitc` case's inputs, heteroscedastic-noise form included."""
    from gpExp.gp import GP
    from gpExp.kernels import KernelSquaredExponential
    c = "fitc_deriv"
    ix = golden.index[c]
    X, y, Z = golden(c, "X"), golden(c, "y"), golden(c, "Z")
    k = KernelSquaredExponential(ix["kernel"]["cl"], ix["kernel"]["signalSize"], ix["kernel"]["d"])
    np.random.seed(ix["seed"])
    g = GP(k, ix["noise"], FITC=ix["fitc"])
    g.train(X, y)
    assert np.array_equal(g.fitcnodes, golden(c, "fitcnodes"))
    assert rel(g.evaluateVarianceDerivWRTnewpt(Z), golden(c, "dvar_dnewpt")) <= 1e-8
    assert rel(g.evaluateVarianceDerivative(Z), golden(c, "dvar_dpts")) <= 1e-8

    class Noise:
        def __call__(self, p):
            return 0.02 + 0.01 * np.sum(p ** 2.0, axis=1)

        def deriv(self, p):
            return 0.02 * p

    assert rel(g.evaluateVarianceDerivative(Z, noiseFunc=Noise()), golden(c, "dvar_dpts_noisefunc")) <= 1e-8
    # against the dense algebra on the build's own dense precision (what the reference's formula does with P), 1e-10
    P = g.precisionMatrix
    kv = np.array([k.evaluate(Z, X[j:j + 1]) for j in range(len(X))])          # (N, M)
    beta = P @ kv
    dk = np.stack([k.derivative(Z, X[j:j + 1]) for j in range(len(X))], axis=2)  # (M, d, N)
    want = (-2.0 * np.einsum("mdn,nm->md", dk, beta)).reshape(-1)
    assert rel(g.evaluateVarianceDerivWRTnewpt(Z), want) <= 1e-9
