"""GPU parity tests (-m gpu): the HIP path, called through the C ABI, against
(1) the golden vectors produced by the reference and (2) the CPU oracle on seeded inputs.

Tolerances (fp64):
  kernel-matrix entries      1e-13 relative (max-norm): same formula, different rounding order/exp
  posterior mean / variance  1e-10 relative (BASELINE.json north_star), max-norm
  log-marginal likelihood    1e-10 relative
  index selections           exact
The reference factorises with pinv (SVD), the GPU with Cholesky; all fixtures have cond(K) < 1e4
("pinv != Cholesky").
"""
import numpy as np
import pytest

from oracle import gpexp_oracle as orc
from helpers import elementwise, c4_lite_inputs

pytestmark = pytest.mark.gpu

KIND = {"se": 0, "matern32": 1, "matern52": 2, "mehler": 3}
GP_CASES = ["kat1_demo", "kat2_matern32", "kat3_mehler", "se_iso_d3_n96", "se_ard_d8_n130",
            "matern32_d8_n200", "mehler_d3_n64", "se_ard_d2_n77_ppnoise", "se_iso_d3_n300"]


def rel(a, b):
    a = np.asarray(a, dtype=float)
    b = np.asarray(b, dtype=float)
    return np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)


@pytest.fixture(scope="module")
def dev():
    from gpexp_amd import device
    return device


@pytest.fixture(scope="module")
def ctx(dev):
    return dev.context()


def spec_of(dev, s):
    k = s["kind"]
    d = s["d"]
    if k == "se":
        cl = np.asarray(s["cl"], dtype=float)
        if cl.size == 1:
            cl = np.tile(cl, d)
        return dev.KernelSpec(KIND[k], d, list(cl) + [s["signalSize"]])
    if k in ("matern32", "matern52"):
        return dev.KernelSpec(KIND[k], d, [s["rho"], s["signalSize"]])
    return dev.KernelSpec(KIND[k], d, list(s["t"]))


def test_library_is_native(ctx):
    info = ctx.info()
    assert "gfx950" in info["name"], info
    assert info["cus"] == 256


@pytest.mark.parametrize("case", GP_CASES)
def test_kfill_vs_reference(dev, ctx, golden, case):
    s = golden.index[case]["kernel"]
    X = dev.points(ctx, golden(case, "X"))
    K = dev.kfill(ctx, spec_of(dev, s), X, nugget=golden.noise(case)).to_host()
    assert rel(K, golden(case, "K")) <= 1e-13
    np.testing.assert_allclose(K, golden(case, "K"), rtol=2e-12, atol=1e-300)


@pytest.mark.parametrize("kind,d", [("se", 5), ("se", 8), ("se", 10), ("se", 17), ("matern32", 8), ("matern52", 8),
                                    ("matern52", 3), ("mehler", 4), ("mehler", 8)])
def test_kfill_rect_vs_oracle(dev, ctx, kind, d):
    rng = np.random.default_rng(11 + d)
    X = rng.uniform(-1, 1, (150, d))
    Z = rng.uniform(-1, 1, (70, d))
    if kind == "se":
        s = dict(kind="se", cl=list(0.4 + 0.05 * np.arange(d)), signalSize=1.7, d=d)
    elif kind == "mehler":
        s = dict(kind="mehler", t=list(0.2 + 0.05 * np.arange(d)), d=d)
    else:
        s = dict(kind=kind, rho=0.8, signalSize=1.2, d=d)
    Kxz = dev.kfill(ctx, spec_of(dev, s), dev.points(ctx, X), Z=dev.points(ctx, Z)).to_host()
    want = orc.cross_matrix(s, Z, X).T  # oracle builds (M,N); the device keeps (N,M)
    assert Kxz.shape == (150, 70)
    assert rel(Kxz, want) <= 1e-13
    kd = dev.kdiag(ctx, spec_of(dev, s), dev.points(ctx, Z))
    assert rel(kd, orc.kernel_diag(s, Z)) <= 1e-13


@pytest.mark.parametrize("bt", [0, 1])
@pytest.mark.parametrize("acc", [0, 1])
def test_gemm_mfma_vs_numpy(dev, ctx, bt, acc):
    rng = np.random.default_rng(5)
    m, n, k = 256, 384, 128 + 16 * 3
    A = rng.standard_normal((m, k))
    B = rng.standard_normal((n, k) if bt else (k, n))
    C0 = rng.standard_normal((m, n))
    dA = dev.DeviceMatrix.from_host(ctx, A, pad=False)
    dB = dev.DeviceMatrix.from_host(ctx, B, pad=False)
    dC = dev.DeviceMatrix.from_host(ctx, C0, pad=False)
    dev.dbg_gemm(ctx, dA, dB, dC, bt, acc)
    prod = A @ (B.T if bt else B)
    want = C0 - prod if acc else prod
    assert rel(dC.to_host(), want) <= 1e-13


@pytest.mark.parametrize("tri", [1, 2, 3])
def test_gemm_triangular_operands(dev, ctx, tri):
    """The k range of every tile is cut to the operand's non-zero part: same result as the dense product of the
    explicitly zero-filled operand."""
    rng = np.random.default_rng(70 + tri)
    if tri == 1:      # A lower triangular (m x m), C = A B  / C -= A B
        m, n = 384, 256
        A = np.tril(rng.standard_normal((m, m))); B = rng.standard_normal((m, n))
        for acc in (0, 1):
            C0 = rng.standard_normal((m, n))
            dC = dev.DeviceMatrix.from_host(ctx, C0, pad=False)
            dev.dbg_gemm_tri(ctx, dev.DeviceMatrix.from_host(ctx, A, pad=False), dev.DeviceMatrix.from_host(ctx, B, pad=False),
                             dC, 0, acc, 1)
            assert rel(dC.to_host(), C0 - A @ B if acc else A @ B) <= 1e-13
    elif tri == 2:    # B lower triangular (n x n) used transposed, C = A B^T
        m, n = 640, 384
        A = rng.standard_normal((m, n)); B = np.tril(rng.standard_normal((n, n)))
        dC = dev.DeviceMatrix.zeros(ctx, m, n)
        dev.dbg_gemm_tri(ctx, dev.DeviceMatrix.from_host(ctx, A, pad=False), dev.DeviceMatrix.from_host(ctx, B, pad=False), dC, 1, 0, 2)
        assert rel(dC.to_host()[:m, :n], A @ B.T) <= 1e-13
    else:             # lower C = U U^T, U upper triangular
        m = 512
        U = np.triu(rng.standard_normal((m, m)))
        dU = dev.DeviceMatrix.from_host(ctx, U, pad=False)
        dC = dev.DeviceMatrix.zeros(ctx, m, m)
        dev.dbg_gemm_tri(ctx, dU, dU, dC, 1, 0, 3)
        il = np.tril_indices(m)
        assert rel(dC.to_host()[il], (U @ U.T)[il]) <= 1e-13


@pytest.mark.parametrize("tri,m,n", [(2, 16384, 1024), (2, 2304, 1152), (2, 20480, 512), (4, 16384, 1024), (4, 1920, 640),
                                     (1, 1024, 16384), (1, 1152, 2304)])
def test_gemm_triangular_operands_longest_first_tile_order(dev, ctx, tri, m, n):
    """Round 5: inside a super-block the tiles of a triangular-operand launch are enumerated longest k range first (tile_of,
    gemm_f64.hip).  The enumeration must still visit every tile exactly once -- for 128-tiles over several super-blocks and XCD
    rounds (>= 1024 tiles), ragged super-blocks, and the 64-tile form: same result as the dense product of the zero-filled
    operand, every entry."""
    rng = np.random.default_rng(1000 * tri + m % 97 + n)
    if tri == 1:      # A (m x m) lower triangular, C = A B
        A = np.tril(rng.standard_normal((m, m))); B = rng.standard_normal((m, n)); bt = 0; want = A @ B
    elif tri == 2:    # B (n x n) lower triangular used transposed, C = A B^T
        A = rng.standard_normal((m, n)); B = np.tril(rng.standard_normal((n, n))); bt = 1; want = A @ B.T
    else:             # B (n x n) lower triangular used as it is, C = A B
        A = rng.standard_normal((m, n)); B = np.tril(rng.standard_normal((n, n))); bt = 0; want = A @ B
    dC = dev.DeviceMatrix.from_host(ctx, np.full(want.shape, np.nan), pad=False)     # an unvisited tile would stay NaN
    dev.dbg_gemm_tri(ctx, dev.DeviceMatrix.from_host(ctx, A, pad=False), dev.DeviceMatrix.from_host(ctx, B, pad=False), dC, bt, 0, tri)
    got = dC.to_host()
    assert np.all(np.isfinite(got)) and rel(got, want) <= 1e-13


@pytest.mark.parametrize("mode,m,n,k,parts", [(0, 256, 384, 4096, 4), (1, 1152, 1152, 2048, 2), (1, 4096, 4096, 512, 4),
                                               (2, 512, 1024, 1024, 4), (2, 192, 2048, 2048, 2), (3, 512, 1024, 1024, 4)])
def test_gemm_in_slices_of_the_k_range(dev, ctx, mode, m, n, k, parts):
    """Round 5: a small C under a long k range (FITC's nu x nu x N product; the few-row products of gpx_refit_rows) is computed as
    slices of the k range whose partials are summed in slice order.  Every tile of every slice exactly once (the jobs are dealt
    to the XCDs like the super-blocks of the main kernel); lower: only the diagonal tiles and below are touched; mode 3 writes
    the product over its own left operand."""
    rng = np.random.default_rng(31 * mode + m + k)
    A = rng.standard_normal((m, k))
    B = rng.standard_normal((n, k))
    C0 = rng.standard_normal((m, n))
    dA = dev.DeviceMatrix.from_host(ctx, A, pad=False)
    dB = dev.DeviceMatrix.from_host(ctx, B, pad=False)
    if mode == 3:
        assert n == k
        dev.dbg_gemm_ksplit(ctx, dA, dB, dA, mode, parts)
        assert rel(dA.to_host(), A @ B.T) <= 1e-13
        return
    dC = dev.DeviceMatrix.from_host(ctx, C0, pad=False)
    dev.dbg_gemm_ksplit(ctx, dA, dB, dC, mode, parts)
    got, want = dC.to_host(), C0 - A @ B.T
    if mode == 1:
        i, j = np.indices((m, n))
        low = j <= (i | 127)
        assert rel(got[low], want[low]) <= 1e-13 and np.array_equal(got[~low], C0[~low])
    else:
        assert rel(got, want) <= 1e-13


def test_gemm_lower_only(dev, ctx):
    rng = np.random.default_rng(6)
    A = rng.standard_normal((384, 64))
    C0 = rng.standard_normal((384, 384))
    dA = dev.DeviceMatrix.from_host(ctx, A)
    dC = dev.DeviceMatrix.from_host(ctx, C0)
    dev.dbg_gemm(ctx, dA, dA, dC, 1, 1, lower=True)
    got = dC.to_host()
    want = C0 - A @ A.T
    # contract: every entry on/below the diagonal is updated; entries above it are either updated or untouched
    # (tile granularity -- 128 or 64 -- is the kernel's business), never anything else
    il = np.tril_indices(384)
    assert rel(got[il], want[il]) <= 1e-13
    iu = np.triu_indices(384, 1)
    touched = np.isclose(got[iu], want[iu], rtol=1e-12, atol=0)
    untouched = got[iu] == C0[iu]
    assert np.all(touched | untouched)
    assert untouched.sum() >= 128 * 128 * 3 - 1  # the three strictly-upper 128-blocks are never touched


@pytest.mark.parametrize("n", [1, 5, 128, 129, 300, 1000])
def test_potrf_potrs_logdet_vs_lapack(dev, ctx, n):
    rng = np.random.default_rng(n)
    d = 3
    X = rng.uniform(-1, 1, (n, d))
    s = dict(kind="se", cl=[0.3], signalSize=1.0, d=d)
    K = orc.cov_matrix(s, X, 0.05, row_loop=False)
    y = rng.standard_normal(n)
    dK = dev.kfill(ctx, spec_of(dev, s), dev.points(ctx, X), nugget=0.05)
    dev.potrf(ctx, dK)
    L = dK.to_host(tri=1)
    Lref = np.linalg.cholesky(K)
    assert rel(L, Lref) <= 1e-11
    alpha = dev.potrs(ctx, dK, y)
    assert rel(alpha, np.linalg.solve(K, y)) <= 1e-10
    assert dev.logdet(ctx, dK) == pytest.approx(np.linalg.slogdet(K)[1], rel=1e-11, abs=1e-11)


def test_potrf_reports_non_positive_pivot(dev, ctx):
    from gpexp_amd._lib import NotPositiveDefinite
    X = np.array([[0.1], [0.1], [0.5]])  # duplicate point, nugget 0 -> singular
    s = dict(kind="se", cl=[0.3], signalSize=1.0, d=1)
    dK = dev.kfill(ctx, spec_of(dev, s), dev.points(ctx, X), nugget=0.0)
    with pytest.raises(NotPositiveDefinite) as e:
        dev.potrf(ctx, dK)
    assert e.value.pivot == 2


@pytest.mark.parametrize("case", GP_CASES)
def test_fit_posterior_loglike_vs_reference(dev, ctx, golden, case):
    s = golden.index[case]["kernel"]
    sp = spec_of(dev, s)
    Xh, y, Zh = golden(case, "X"), golden(case, "y"), golden(case, "Z")
    X, Z = dev.points(ctx, Xh), dev.points(ctx, Zh)
    L = dev.potrf(ctx, dev.kfill(ctx, sp, X, nugget=golden.noise(case)))
    alpha = dev.potrs(ctx, L, y)
    assert rel(alpha, golden(case, "coeff")) <= 1e-10
    mean, var = dev.posterior(ctx, sp, L, X, alpha, Z)
    assert rel(mean, golden(case, "mean")) <= 1e-10
    assert rel(var, golden(case, "var")) <= 1e-10
    # element-wise: every variance to a relative bound of ITS OWN size, down to entries 1e-3 of the
    # largest (helpers.elementwise) -- 1e-9: the reference's pinv carries ~cond(K) eps of absolute error itself
    assert elementwise(var, golden(case, "var")) <= 1e-9
    n = len(y)
    ll = -0.5 * y @ alpha - 0.5 * dev.logdet(ctx, L) - n / 2.0 * np.log(2 * np.pi)
    assert ll == pytest.approx(float(golden(case, "loglike")), rel=1e-10)
    nc = golden(case, "cov").shape[0]
    cov = dev.posterior_cov(ctx, sp, L, X, dev.points(ctx, Zh[:nc]))
    assert rel(cov, golden(case, "cov")) <= 1e-10
    P = dev.potri(ctx, L).to_host(tri=2)   # the C ABI's contract: lower triangle valid (mirrored here)
    assert rel(P, golden(case, "precision")) <= 1e-9


def test_ivar_vs_reference(dev, ctx, golden):
    c = "kat4_ivar"
    sp = spec_of(dev, golden.index[c]["kernel"])
    X, Z = dev.points(ctx, golden(c, "X")), dev.points(ctx, golden(c, "mc"))
    L = dev.potrf(ctx, dev.kfill(ctx, sp, X, nugget=1e-3))
    assert abs(dev.ivar(ctx, sp, L, X, Z)) == pytest.approx(float(golden(c, "ivar")), rel=1e-10)


def test_greedy_var_indices_bit_exact(dev, ctx, golden):
    c = "kat5_greedy"
    sp = spec_of(dev, golden.index[c]["kernel"])
    C = dev.points(ctx, golden(c, "C"))
    np.testing.assert_array_equal(dev.greedy_var(ctx, sp, C, 8, keep=[0]), golden(c, "gvar_idx"))
    np.testing.assert_array_equal(dev.greedy_var(ctx, sp, C, 10, keep=[7]), golden(c, "gvar_idx_from7"))
    np.testing.assert_array_equal(dev.greedy_var(ctx, sp, C, 9, keep=[3, 11], weights=golden(c, "weights")),
                                  golden(c, "gvar_idx_weighted"))
    assert dev.greedy_var(ctx, sp, C, 3)[0] == 0  # empty start: arg-max of the prior variance


def test_greedy_ivar_indices_bit_exact(dev, ctx, golden):
    c = "kat5_greedy"
    sp = spec_of(dev, golden.index[c]["kernel"])
    Ch, Zh = golden(c, "C"), golden(c, "Z")
    Xh = golden(c, "X0").copy()
    C, Z = dev.points(ctx, Ch), dev.points(ctx, Zh)
    for step in range(4):
        X = dev.points(ctx, Xh)
        L = dev.potrf(ctx, dev.kfill(ctx, sp, X, nugget=1e-3))
        best, costs = dev.greedy_ivar_step(ctx, sp, L, X, C, Z, 1e-3)
        assert best == golden(c, "givar_idx")[step]
        assert rel(costs, golden(c, "givar_allcosts")[step]) <= 1e-10
        assert costs[best] == pytest.approx(golden(c, "givar_cost")[step], rel=1e-10)
        Xh = np.vstack((Xh, Ch[best:best + 1]))


def test_greedy_ivar_multi_pick_resident_state_bit_exact(dev, ctx, golden):
    """gpx_greedy_ivar: the reference-pinned KAT5 four-pick sequence from ONE call that keeps L^-1 K(X, C)
    and cov(Z, C | design) resident and conditions them on each pick by a rank-one update -- winners exact, the winner's cost and
    EVERY candidate's cost at every pick 1e-10 against the reference's refits (experimentalDesign.py:79-117 per SURVEY 8c)."""
    c = "kat5_greedy"
    sp = spec_of(dev, golden.index[c]["kernel"])
    X = dev.points(ctx, golden(c, "X0"))
    L = dev.potrf(ctx, dev.kfill(ctx, sp, X, nugget=1e-3))
    C, Z = dev.points(ctx, golden(c, "C")), dev.points(ctx, golden(c, "Z"))
    idx, cost, allc = dev.greedy_ivar(ctx, sp, L, X, C, Z, 1e-3, 4, want_all=True)
    np.testing.assert_array_equal(idx, golden(c, "givar_idx")[:4])
    assert rel(cost, golden(c, "givar_cost")[:4]) <= 1e-10
    for t in range(4):
        assert rel(allc[t], golden(c, "givar_allcosts")[t]) <= 1e-10
    i2, c2 = dev.greedy_ivar(ctx, sp, L, X, C, Z, 1e-3, 4)
    assert np.array_equal(i2, idx) and np.array_equal(c2, cost)          # deterministic
    # the class-API form, and one pick = gpx_greedy_ivar_step
    from gpExp.experimentalDesign import performGreedyIVARExperimentalDesign, greedyIVARStep
    from gpExp.gp import GP
    from gpExp.kernels import KernelSquaredExponential
    s = golden.index[c]["kernel"]
    g = GP(KernelSquaredExponential(list(s["cl"]), s["signalSize"], s["d"]), 1e-3)
    g.addNodesAndComputeCovariance(golden(c, "X0"))
    pts = performGreedyIVARExperimentalDesign(g, golden(c, "C"), golden(c, "Z"), 4)
    np.testing.assert_array_equal(pts, golden(c, "C")[list(golden(c, "givar_idx")[:4])])
    b1, c1 = greedyIVARStep(g, golden(c, "C"), golden(c, "Z"))
    assert b1 == idx[0] and np.array_equal(c1, allc[0])                  # the set-up IS the step function's arithmetic


@pytest.mark.parametrize("seed", range(6))
def test_greedy_ivar_multi_pick_equals_refit_loop(dev, ctx, seed):
    """Random configurations (kernel kind, d, ragged N / M / nMC, per-point training noise): k picks from the resident state ==
    k rounds of gpx_greedy_ivar_step + an actual refit on the winner -- winners equal wherever the step's winner is clear
    (gap > 1e-9 relative), costs 1e-9."""
    rng = np.random.default_rng(100 + seed)
    d = int(rng.integers(1, 7))
    n, m, nmc, k = int(rng.integers(5, 400)), int(rng.integers(3, 700)), int(rng.integers(2, 300)), int(rng.integers(2, 9))
    kind = ["se", "matern32", "matern52"][seed % 3]
    s = dict(kind="se", cl=list(rng.uniform(0.3, 0.9, d)), signalSize=1.2, d=d) if kind == "se" else \
        dict(kind=kind, rho=float(rng.uniform(0.4, 0.9)), signalSize=1.1, d=d)
    sp = spec_of(dev, s)
    noise = float(rng.uniform(0.01, 0.2))
    Xh, Ch, Zh = rng.uniform(-1, 1, (n, d)), rng.uniform(-1, 1, (m, d)), rng.uniform(-1, 1, (nmc, d))
    X, C, Z = dev.points(ctx, Xh), dev.points(ctx, Ch), dev.points(ctx, Zh)
    L = dev.potrf(ctx, dev.kfill(ctx, sp, X, nugget=noise))
    idx, cost, allc = dev.greedy_ivar(ctx, sp, L, X, C, Z, noise, k, want_all=True)
    Xc = Xh.copy()
    for t in range(k):
        Xd = dev.points(ctx, Xc)
        Lt = dev.potrf(ctx, dev.kfill(ctx, sp, Xd, nugget=noise))
        best, costs = dev.greedy_ivar_step(ctx, sp, Lt, Xd, C, Z, noise)
        assert rel(allc[t], costs) <= 1e-9, (seed, t)
        srt = np.sort(costs)
        if len(srt) < 2 or srt[1] - srt[0] > 1e-9 * abs(srt[0]):
            assert idx[t] == best, (seed, t, idx[t], best)
        Xc = np.vstack((Xc, Ch[idx[t]:idx[t] + 1]))


def test_posterior_chunked_equals_unchunked(dev, ctx, monkeypatch):
    """ragged sizes + forced chunking of the evaluation set (size-independent property)."""
    rng = np.random.default_rng(9)
    s = dict(kind="matern52", rho=0.6, signalSize=1.1, d=4)
    sp = spec_of(dev, s)
    Xh = rng.uniform(-1, 1, (333, 4))
    Zh = rng.uniform(-1, 1, (517, 4))
    y = rng.standard_normal(333)
    X, Z = dev.points(ctx, Xh), dev.points(ctx, Zh)
    L = dev.potrf(ctx, dev.kfill(ctx, sp, X, nugget=0.1))
    alpha = dev.potrs(ctx, L, y)
    m0, v0 = dev.posterior(ctx, sp, L, X, alpha, Z)
    monkeypatch.setenv("GPX_CROSS_BYTES", str(384 * 128 * 8))  # -> 128-column chunks
    m1, v1 = dev.posterior(ctx, sp, L, X, alpha, Z)
    np.testing.assert_array_equal(m0, m1)
    np.testing.assert_array_equal(v0, v1)
    # oracle (pinv) on the same inputs
    model = orc.fit(s, Xh, y, 0.1)
    mo, vo = orc.posterior(s, model, Zh)
    assert rel(m0, mo) <= 1e-10 and rel(v0, vo) <= 1e-10


def test_empty_evaluation_set(dev, ctx):
    s = dict(kind="se", cl=[0.3], signalSize=1.0, d=2)
    sp = spec_of(dev, s)
    X = dev.points(ctx, np.random.default_rng(1).uniform(-1, 1, (10, 2)))
    L = dev.potrf(ctx, dev.kfill(ctx, sp, X, nugget=0.1))
    m, v = dev.posterior(ctx, sp, L, X, np.zeros(10), dev.points(ctx, np.zeros((0, 2))))
    assert m.shape == (0,) and v.shape == (0,)


@pytest.mark.gpu
def test_c2_full_size_against_reference(dev, ctx, golden):
    """BASELINE config C2 at full size (N=4096, d=3, iso-SE, cond(K) = 1.2e3): Cholesky on the GPU against the reference's
    pinv / slogdet results (north_star tolerance 1e-10 on mean, variance and log-marginal)."""
    from gpExp.kernels import KernelSquaredExponential
    from gpExp.gp import GP
    c = "c2_full"
    ix = golden.index[c]
    rng = np.random.default_rng(ix["seed"])
    N, M, d = ix["N"], ix["M"], ix["kernel"]["d"]
    X = rng.uniform(-1, 1, (N, d))
    y = np.sin(2 * np.pi * X.sum(1) / d) + np.sqrt(ix["noise"]) * rng.standard_normal(N)
    Z = rng.uniform(-1, 1, (M, d))
    g = GP(KernelSquaredExponential(list(ix["kernel"]["cl"]), ix["kernel"]["signalSize"], d), ix["noise"])
    g.train(X, y)
    assert rel(g.coeff, golden(c, "coeff")) <= 1e-10
    mean, var = g.evaluate(Z[:256], compvar=1)
    assert rel(mean, golden(c, "mean256")) <= 1e-10
    assert rel(var, golden(c, "var256")) <= 1e-10
    assert g.computeLogLike(X, y) == pytest.approx(float(golden(c, "loglike")), rel=1e-10)
    assert float(y @ g.coeff) == pytest.approx(float(golden(c, "ytalpha")), rel=1e-10)


def test_c4_lite_blocked_factorisation_against_reference(dev, ctx, golden):
    """`c4_lite` (tests/golden/make_golden_r4.py: the REFERENCE at N=8192, d=8, Matern-3/2, rho=0.5, noise=0.1, 256 evaluation
    points, cond(K) = 5e2): the smallest size that takes the blocked look-ahead factorisation (potrf_blocked, N >= 8192), its
    1024-order block inverses and the out-of-place posterior solve (VERDICT r3 missing 3).  Through the class API, against
    gp.py:76-145 (train / evaluate), gp.py:213-259 (evaluateVariance) and gp.py:373-440 (computeLogLike): 1e-10 max-norm on
    coeff / mean / variance / log-marginal, and 5e-10 ELEMENT-WISE on the variances."""
    from gpExp.kernels import KernelIsoMatern
    from gpExp.gp import GP
    c = "c4_lite"
    ix = golden.index[c]
    X, y, Z = c4_lite_inputs(ix)
    k = ix["kernel"]
    g = GP(KernelIsoMatern(k["rho"], k["signalSize"], k["d"]), ix["noise"])
    g.train(X, y)
    assert rel(g.coeff, golden(c, "coeff")) <= 1e-10
    mean, var = g.evaluate(Z, compvar=1)
    assert rel(mean, golden(c, "mean256")) <= 1e-10
    assert rel(var, golden(c, "var256")) <= 1e-10
    assert elementwise(var, golden(c, "var256")) <= 5e-10
    signed = g.evaluateVariance(Z)
    assert elementwise(signed, golden(c, "varsigned256")) <= 5e-10
    assert elementwise(mean, golden(c, "mean256"), floor=1e-2) <= 1e-9
    assert g.computeLogLike(X, y) == pytest.approx(float(golden(c, "loglike")), rel=1e-10)
    assert float(y @ g.coeff) == pytest.approx(float(golden(c, "ytalpha")), rel=1e-10)
