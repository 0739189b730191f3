"""SURVEY.md 8 f1 on the device (-m gpu): gradients of the posterior variance w.r.t. point locations --
GP.evaluateVarianceDerivative (gp.py:282-341), GP.evaluateVarianceDerivWRTnewpt (gp.py:261-280) and
costFunctionGP_IVAR.derivative (experimentalDesign.py:168-179) -- for the squared-exponential and the 1-D Mehler kernel,
with and without a heteroscedastic noise model, against the reference's own vectors (tests/golden/make_golden*.py) and the
oracle's restatement at larger sizes.  Tolerance 1e-9: the quantities are built on the explicit inverse in the reference
(pinv) and on two triangular solves here."""
import os

import numpy as np
import pytest

from oracle import gpexp_oracle as orc
from helpers import NoiseFunc, rel

pytestmark = pytest.mark.gpu


def kernel_of(s):
    from gpExp.kernels import KernelSquaredExponential, KernelMehler1D
    if s["kind"] == "se":
        return KernelSquaredExponential(list(s["cl"]), s["signalSize"], s["d"])
    return KernelMehler1D(s["t"][0], 1)


def space_of(d, nf=None):
    from gpExp.approximation import Space
    return Space(d, lambda size: np.random.rand(size[0], size[1]) * 2 - 1, lambda p: np.ones(len(p)) / 2 ** d, noise=nf)


@pytest.mark.parametrize("case", ["varderiv", "varderiv_mehler1d"])
def test_point_derivatives_vs_reference(golden, case):
    from gpExp.gp import GP
    s = golden.index[case]["kernel"]
    X, Z = golden(case, "X"), golden(case, "Z")
    g = GP(kernel_of(s), golden.index[case]["noise"])
    g.addNodesAndComputeCovariance(X)
    D = g.evaluateVarianceDerivative(Z)
    assert D.shape == (X.size, len(Z))
    assert rel(D, golden(case, "dvar_dpts")) <= 1e-9
    assert rel(g.evaluateVarianceDerivWRTnewpt(Z), golden(case, "dvar_dnew")) <= 1e-9
    assert rel(g.kernel.derivative(Z, X[:1]) if case.endswith("1d") else g.kernel.derivative(X, Z[:1]),
               golden(case, "kernel_derivative")) <= 1e-12


def test_heteroscedastic_derivatives_vs_reference(golden):
    """noiseFunc terms of gp.py:314-317 and the IVAR gradient of experimentalDesign.py:173-177."""
    from gpExp.gp import GP
    from gpExp.experimentalDesign import costFunctionGP_IVAR
    c = "varderiv_nf"
    s = golden.index[c]["kernel"]
    X, Z = golden(c, "X"), golden(c, "Z")
    nf = NoiseFunc(2)
    g = GP(kernel_of(s), 1e-3)
    g.addNodesAndComputeCovariance(X, noiseIn=nf(X))
    assert rel(g.evaluateVarianceDerivative(Z, noiseFunc=nf), golden(c, "dvar_dpts")) <= 1e-9
    assert rel(g.evaluateVarianceDerivWRTnewpt(Z), golden(c, "dvar_dnew")) <= 1e-9
    cf = costFunctionGP_IVAR(GP(kernel_of(s), 1e-3), len(X), space_of(2, nf), mcPoints=Z)
    assert cf.evaluate(X) == pytest.approx(float(golden(c, "ivar")), rel=1e-10)
    assert rel(cf.derivative(X), golden(c, "ivar_grad")) <= 1e-9
    cf0 = costFunctionGP_IVAR(GP(kernel_of(s), 1e-3), len(X), space_of(2), mcPoints=Z)
    assert cf0.evaluate(X) == pytest.approx(float(golden(c, "ivar_homo")), rel=1e-10)
    assert rel(cf0.derivative(X), golden(c, "ivar_grad_homo")) <= 1e-9
    g0 = GP(kernel_of(s), 1e-3)
    g0.addNodesAndComputeCovariance(X)
    assert rel(g0.evaluateVarianceDerivative(Z), golden(c, "dvar_dpts_homo")) <= 1e-9


def test_whole_set_coincidence_branch_vs_reference(golden):
    """ONE evaluation point equal to a training point + noiseFunc: the branch at gp.py:318-320."""
    from gpExp.gp import GP
    c = "varderiv_single"
    X, Z = golden(c, "X"), golden(c, "Z")
    nf = NoiseFunc(2)
    g = GP(kernel_of(golden.index[c]["kernel"]), 1e-3)
    g.addNodesAndComputeCovariance(X, noiseIn=nf(X))
    assert rel(g.evaluateVarianceDerivative(Z, noiseFunc=nf), golden(c, "dvar_dpts")) <= 1e-9


def test_mehler1d_ivar_gradient_vs_reference(golden):
    from gpExp.gp import GP
    from gpExp.experimentalDesign import costFunctionGP_IVAR
    c = "varderiv_mehler1d"
    X, Z = golden(c, "X"), golden(c, "Z")
    cf = costFunctionGP_IVAR(GP(kernel_of(golden.index[c]["kernel"]), golden.index[c]["noise"]), len(X), space_of(1),
                             mcPoints=Z)
    assert cf.evaluate(X) == pytest.approx(float(golden(c, "ivar")), rel=1e-10)
    assert rel(cf.derivative(X), golden(c, "ivar_grad")) <= 1e-9


@pytest.mark.parametrize("kind,with_nf,dups", [("se", False, False), ("se", True, False), ("se", True, True),
                                               ("mehler1d", False, False), ("mehler1d", True, True)])
def test_derivatives_vs_oracle_larger(kind, with_nf, dups):
    """n = 150 training points (ragged against the 128 tiles), M = 333 evaluation points, evaluation chunks of 128
    forced through GPX_CROSS_BYTES; duplicated training points exercise the coincidence mask of gp.py:308."""
    from gpExp.gp import GP
    from gpExp.experimentalDesign import costFunctionGP_IVAR
    rng = np.random.default_rng(5 + len(kind))
    d = 1 if kind == "mehler1d" else 3
    n, m = 150, 333
    X = rng.uniform(-1, 1, (n, d))
    if dups:
        X[17] = X[3]
        X[140] = X[3]
    Z = rng.uniform(-1, 1, (m, d))
    s = dict(kind="se", cl=[0.5, 0.7, 0.9], signalSize=1.3, d=3) if kind == "se" else dict(kind="mehler1d", t=[0.6], d=1)
    nf = NoiseFunc(d) if with_nf else None
    noise = 0.05
    nug = nf(X) if with_nf else noise
    model = orc.fit(s, X, None, nug)
    want = orc.variance_derivative(s, model, Z, nf)
    g = GP(kernel_of(s), noise)
    g.addNodesAndComputeCovariance(X, noiseIn=(nug if with_nf else None))
    old = os.environ.get("GPX_CROSS_BYTES")
    os.environ["GPX_CROSS_BYTES"] = str(3 * 256 * 8 * 128)     # 128 evaluation points per chunk
    try:
        got = g.evaluateVarianceDerivative(Z, noiseFunc=nf)
        gnew = g.evaluateVarianceDerivWRTnewpt(Z)
    finally:
        if old is None:
            os.environ.pop("GPX_CROSS_BYTES", None)
        else:
            os.environ["GPX_CROSS_BYTES"] = old
    assert rel(got, want) <= 1e-9
    assert rel(gnew, orc.variance_deriv_wrt_newpt(s, model, Z)) <= 1e-9
    assert rel(g.evaluateVarianceDerivative(Z, noiseFunc=nf), got) <= 1e-13      # one chunk == several chunks
    cf = costFunctionGP_IVAR(GP(kernel_of(s), noise), n, space_of(d, nf), mcPoints=Z)
    assert rel(cf.derivative(X), want.sum(axis=1) / m) <= 1e-9
    assert rel(cf.derivative(X), orc.ivar_grad(s, X, Z, noise, nf)) <= 1e-9


def _se_gradients_numpy(cl, sig, X, Z, nug):
    """The two squared-exponential gradients as dense NumPy algebra (the oracle's loops are O(N^2 d) Python iterations):
    dk(u, p)[l] = -s (u_l - p_l) / cl_l^2 k(u, p) is linear in the difference, so both collapse to products with the
    coordinate arrays.  Checked against the oracle at n = 150 below before it is trusted at n = 4100."""
    cl = np.asarray(cl, dtype=float)
    def K(A, B):
        D = (A[:, None, :] - B[None, :, :]) / cl
        return sig * np.exp(-0.5 * np.sum(D * D, axis=2))
    K0, Kxz = K(X, X), K(X, Z)
    beta = np.linalg.solve(K0 + nug * np.eye(len(X)), Kxz)
    c = -sig / cl ** 2
    Q = beta * Kxz                                   # N x M
    g1 = 2.0 * c * (Q @ Z - X * Q.sum(axis=1, keepdims=True))
    R = K0 * (beta @ beta.T)
    g2 = 2.0 * c * (X * R.sum(axis=1, keepdims=True) - R @ X)
    ivar_grad = ((g1 + g2) / Z.shape[0]).reshape(-1)
    newpt = (-2.0 * c * (Z * Q.sum(axis=0)[:, None] - Q.T @ X)).reshape(-1)
    # the (N d) x M matrix of gp.py:282-341 for the first evaluation points
    mf = min(300, Z.shape[0])
    b, kz = beta[:, :mf], Kxz[:, :mf]
    K0b = K0 @ b
    full = np.zeros((len(X), X.shape[1], mf))
    for l in range(X.shape[1]):
        t1 = c[l] * (Z[None, :mf, l] - X[:, None, l]) * kz
        t2 = c[l] * (X[:, None, l] * K0b - K0 @ (X[:, None, l] * b))
        full[:, l, :] = b * (2.0 * t1 + 2.0 * t2)
    return ivar_grad, newpt, full.reshape(len(X) * X.shape[1], mf)


def test_large_factor_path_of_the_gradients():
    """From np >= 4096 both solves for beta go through the factor's block inverses, S = beta beta^T runs as slices of its k
    range, and the row kernels are the squared exponential's linear-difference forms (grad.hip).  n = 4100 (ragged against the
    128 tiles), M = 8200, d = 5; the dense NumPy algebra is first checked against the oracle at n = 150."""
    from gpexp_amd import device as dev
    ctx = dev.context()
    d, sig, nug = 5, 1.3, 0.05
    cl = [0.5, 0.6, 0.7, 0.8, 0.9]
    s = dict(kind="se", cl=cl, signalSize=sig, d=d)
    sp = dev.KernelSpec(dev.K_SE, d, cl + [sig])
    for n, m, check_oracle in ((150, 333, True), (4100, 8200, False)):
        rng = np.random.default_rng(n)
        X, Z = rng.uniform(-1, 1, (n, d)), rng.uniform(-1, 1, (m, d))
        want_g, want_n, want_f = _se_gradients_numpy(cl, sig, X, Z, nug)
        if check_oracle:
            model = orc.fit(s, X, None, nug)
            assert rel(want_g, orc.ivar_grad(s, X, Z, nug)) <= 1e-10
            assert rel(want_n, orc.variance_deriv_wrt_newpt(s, model, Z)) <= 1e-10
            assert rel(want_f, orc.variance_derivative(s, model, Z[:300])) <= 1e-10
        Xd, Zd = dev.points(ctx, X), dev.points(ctx, Z)
        L = dev.potrf(ctx, dev.kfill(ctx, sp, Xd, nugget=nug))
        assert rel(np.asarray(dev.ivar_grad(ctx, sp, L, Xd, Zd)).reshape(-1), want_g) <= 1e-9
        assert rel(np.asarray(dev.var_grad_newpt(ctx, sp, L, Xd, Zd)).reshape(-1), want_n) <= 1e-9
        assert rel(dev.var_grad(ctx, sp, L, Xd, dev.points(ctx, Z[:300])), want_f) <= 1e-9


@pytest.mark.parametrize("n,m", [(1100, 8300), (4100, 4500)])
def test_cost_then_gradient_at_the_same_design_share_the_forward_solve(n, m):
    """An optimiser evaluates the IVAR cost and then its gradient at one design (experimentalDesign.py:471-489).  From 1024
    design points the cost keeps W = L^-1 K(X, Z) on the device (gpx_ivar_keep) and the gradient starts from it
    (gpx_ivar_grad_w): same cost, same gradient as the stand-alone calls -- below and above the order from which the solves go
    through the block inverses -- and through the class API the second refit is the kept factor."""
    from gpexp_amd import device as dev
    from gpExp.gp import GP
    from gpExp.experimentalDesign import costFunctionGP_IVAR
    ctx = dev.context()
    d = 3
    rng = np.random.default_rng(n + m)
    X, Z = rng.uniform(-1, 1, (n, d)), rng.uniform(-1, 1, (m, d))
    s = dict(kind="se", cl=[0.5, 0.7, 0.9], signalSize=1.3, d=d)
    sp = dev.KernelSpec(dev.K_SE, d, [0.5, 0.7, 0.9, 1.3])
    Xd, Zd = dev.points(ctx, X), dev.points(ctx, Z)
    L = dev.potrf(ctx, dev.kfill(ctx, sp, Xd, nugget=0.05))
    cost, W = dev.ivar(ctx, sp, L, Xd, Zd, keep=True)
    assert W is not None and cost == dev.ivar(ctx, sp, L, Xd, Zd)
    g0 = dev.ivar_grad(ctx, sp, L, Xd, Zd)
    assert rel(dev.ivar_grad(ctx, sp, L, Xd, Zd, W=W), g0) <= 1e-12
    del W
    cf = costFunctionGP_IVAR(GP(kernel_of(s), 0.05), n, space_of(d, None), mcPoints=Z)
    c1 = cf.evaluate(X)
    assert cf._w_kept is not None and c1 == pytest.approx(abs(cost), rel=1e-13)
    kept_factor = cf.gaussianProcess._L
    g1 = cf.derivative(X)
    assert cf.gaussianProcess._L is kept_factor and cf._w_kept[0] is kept_factor    # no refit; the kept solve stays (read only)
    assert rel(g1, g0) <= 1e-12
    assert cf.evaluate(X) == c1                                                # the very same fit again: the kept cost
    cf._w_kept = None
    assert rel(cf.derivative(X), g0) <= 1e-12                                  # without a kept solve: the stand-alone path
    X2 = X.copy(); X2[-1] += 0.01
    cf.evaluate(X)
    g2 = cf.derivative(X2)                                                     # another design: the kept solve must not be used
    fresh = costFunctionGP_IVAR(GP(kernel_of(s), 0.05), n, space_of(d, None), mcPoints=Z)
    assert rel(g2, fresh.derivative(X2)) <= 1e-12
    # a batch loop moves the LAST points only (experimentalDesign.py:694-751): the refit keeps the leading rows of the factor and
    # the kept solve keeps its leading rows -- only the trailing rows are solved again (gpx_ivar_update); twice in a row, then the
    # gradient from the updated solve
    cf.evaluate(X)
    for step in range(2):
        X3 = X.copy(); X3[-150:] = rng.uniform(-1, 1, (150, d))
        W_before = cf._w_kept[1]
        c3 = cf.evaluate(X3)
        assert cf.gaussianProcess._last_refit is not None and cf._w_kept[1] is W_before     # refit of the rows + update in place
        f3 = costFunctionGP_IVAR(GP(kernel_of(s), 0.05), n, space_of(d, None), mcPoints=Z)
        f3.gaussianProcess.reuseFactor = False
        assert c3 == pytest.approx(f3.evaluate(X3), rel=1e-12)
        g_full = f3.derivative(X3)
        assert rel(cf.derivative(X3), g_full) <= 1e-11
        # the batch driver pins the leading points (equal bounds): the gradient of the free ones alone (gpx_ivar_grad_rows),
        # zeros for the pinned entries
        cf.pinnedPoints = n - 150
        r0 = ((n - 150) // 128) * 128
        g_free = cf.derivative(X3)
        assert np.all(g_free[:r0 * d] == 0.0) and rel(g_free[r0 * d:], g_full[r0 * d:]) <= 1e-11
        cf.pinnedPoints = 0


def test_unsupported_kernels_raise():
    from gpExp.gp import GP
    from gpExp.kernels import KernelIsoMatern, KernelMehlerND
    X = np.random.default_rng(0).uniform(-1, 1, (9, 2))
    g = GP(KernelIsoMatern(0.5, 1.0, 2), 0.01)
    g.addNodesAndComputeCovariance(X)
    with pytest.raises(AttributeError):        # the reference's KernelIsoMatern has no derivative() either
        g.evaluateVarianceDerivative(X[:3])
    g2 = GP(KernelMehlerND([0.5, 0.3], 2), 0.01)
    g2.addNodesAndComputeCovariance(X)
    with pytest.raises(AttributeError):        # kernels.py:246: "derivative of KernelMehlerND not yet implemented"
        g2.evaluateVarianceDerivWRTnewpt(X[:3])


def test_ivar_cost_with_fitc_model(golden):
    """costFunctionGP_IVAR.evaluate on a GP built with FITC=... goes through the Woodbury variances
    (gp.py:246-255) as the reference does, instead of dereferencing a dense factor that does not exist."""
    from gpExp.gp import GP
    from gpExp.experimentalDesign import costFunctionGP_IVAR, greedyIVARStep
    c = "fitc"
    s = golden.index[c]["kernel"]
    X, Z = golden(c, "X"), golden(c, "Z")
    np.random.seed(golden.index[c]["seed"])
    g = GP(kernel_of(s), golden.index[c]["noise"], FITC=golden.index[c]["fitc"])
    cf = costFunctionGP_IVAR(g, len(X), space_of(2), mcPoints=Z)
    cost = cf.evaluate(X)
    assert np.array_equal(cf.gaussianProcess.fitcnodes, golden(c, "fitcnodes"))
    assert cost == pytest.approx(abs(np.mean(golden(c, "var_signed"))), rel=1e-9)
    # round 6: the gradient too -- the reference sums evaluateVarianceDerivative (Woodbury precision, gp.py:322) over the MC points
    # (experimentalDesign.py:171-177; fixture fitc_deriv: the reference's (N d, M) matrix on the same X, Z and inducing points)
    want = np.sum(golden("fitc_deriv", "dvar_dpts"), axis=1) / float(len(Z))
    got = cf.derivative(X)
    assert got.shape == want.shape and np.max(np.abs(got - want)) <= 1e-8 * np.max(np.abs(want))
    with pytest.raises(NotImplementedError):
        greedyIVARStep(cf.gaussianProcess, X[:5], Z)
