"""GPU edge cases (-m gpu): maximum dimension, 1-D Mehler, per-point noise through loglikeParams(noiseIn=...),
posterior covariance larger than one tile, ragged sizes around the 64/128 tile edges, heavy chunking, tiny problems."""
import numpy as np
import pytest

from oracle import gpexp_oracle as orc

pytestmark = pytest.mark.gpu


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


def test_max_dimension_32(dev, ctx):
    rng = np.random.default_rng(32)
    d = 32
    X = rng.uniform(-1, 1, (100, d))
    Z = rng.uniform(-1, 1, (37, d))
    for s, sp in [(dict(kind="se", cl=list(1.0 + 0.02 * np.arange(d)), signalSize=0.9, d=d),
                   dev.KernelSpec(dev.K_SE, d, list(1.0 + 0.02 * np.arange(d)) + [0.9])),
                  (dict(kind="mehler", t=list(0.1 + 0.01 * np.arange(d)), d=d),
                   dev.KernelSpec(dev.K_MEHLER, d, list(0.1 + 0.01 * np.arange(d)))),
                  # Matern: the rectangular fill stages 1 + 4 operand images per workgroup, 97 KB of LDS at d = 32
                  (dict(kind="matern52", rho=6.0, signalSize=1.1, d=d), dev.KernelSpec(dev.K_MATERN52, d, [6.0, 1.1]))]:
        K = dev.kfill(ctx, sp, dev.points(ctx, X), nugget=0.0).to_host()
        assert rel(K, orc.cov_matrix(s, X, 0.0, row_loop=False)) <= 1e-12
        Kxz = dev.kfill(ctx, sp, dev.points(ctx, X), Z=dev.points(ctx, Z)).to_host()
        assert rel(Kxz, orc.cross_matrix(s, Z, X).T) <= 1e-12
    with pytest.raises(Exception):
        dev.kfill(ctx, dev.KernelSpec(dev.K_SE, 33, [1.0] * 34), dev.points(ctx, np.zeros((4, 33))))


@pytest.mark.parametrize("n", [63, 64, 65, 127, 128, 129, 191, 257])
def test_symmetric_fill_and_fit_at_tile_edges(dev, ctx, n):
    """sizes straddling the 64-point fill tile and the 128 padding: K is exactly symmetric, diagonal exact, fit agrees"""
    rng = np.random.default_rng(n)
    X = rng.uniform(-1, 1, (n, 3))
    s = dict(kind="matern52", rho=0.7, signalSize=1.3, d=3)
    sp = dev.KernelSpec(dev.K_MATERN52, 3, [0.7, 1.3])
    K = dev.kfill(ctx, sp, dev.points(ctx, X), nugget=0.05).to_host()
    np.testing.assert_array_equal(K, K.T)                      # mirror-written tiles
    np.testing.assert_array_equal(np.diag(K), np.full(n, 1.3 + 0.05))  # exact zero distance on the diagonal
    assert rel(K, orc.cov_matrix(s, X, 0.05, row_loop=False)) <= 1e-13
    y = rng.standard_normal(n)
    L = dev.potrf(ctx, dev.kfill(ctx, sp, dev.points(ctx, X), nugget=0.05))
    assert rel(dev.potrs(ctx, L, y), np.linalg.solve(orc.cov_matrix(s, X, 0.05, row_loop=False), y)) <= 1e-10


def test_mehler_1d_kernel_class(golden):
    from gpExp.kernels import KernelMehler1D, KernelMehlerND
    rng = np.random.default_rng(1)
    a, b = rng.uniform(-1, 1, (20, 1)), rng.uniform(-1, 1, (20, 1))
    k1 = KernelMehler1D(0.4, 1)
    want = orc.kernel_eval(dict(kind="mehler", t=[0.4], d=1), a, b)
    assert rel(k1.evaluate(a, b), want) <= 1e-13
    assert rel(KernelMehlerND([0.4], 1).evaluate(a, b[:1]), orc.kernel_eval(dict(kind="mehler", t=[0.4], d=1), a, b[:1])) <= 1e-13
    with pytest.raises(AssertionError):
        k1.evaluate(np.zeros((3, 2)), np.zeros((3, 2)))


def test_loglike_with_per_point_noise():
    """loglikeParams(noiseIn=array): broken in the reference (gp.py:429-430 passes an unknown keyword); implemented with the
    per-point nugget semantics of calculateCovarianceMatrix (gp_kernel_utilities.py:64-65) and checked against the oracle."""
    from gpExp.kernels import KernelSquaredExponential
    from gpExp.gp import GP
    rng = np.random.default_rng(7)
    X = rng.uniform(-1, 1, (90, 2))
    y = rng.standard_normal(90)
    nz = 0.02 * (1 + rng.uniform(0, 1, 90))
    g = GP(KernelSquaredExponential([0.4, 0.9], 2.0, 2), 0.5)
    s = dict(kind="se", cl=[0.4, 0.9], signalSize=2.0, d=2)
    assert g.loglikeParams(X, y, noiseIn=nz) == pytest.approx(orc.loglike(s, X, y, nz), rel=1e-10)
    assert g.pts is None  # loglikeParams does not touch the trained state


def test_posterior_cov_larger_than_a_tile(dev, ctx):
    rng = np.random.default_rng(8)
    s = dict(kind="se", cl=[0.5, 0.6, 0.7], signalSize=1.1, d=3)
    sp = dev.KernelSpec(dev.K_SE, 3, [0.5, 0.6, 0.7, 1.1])
    Xh, Zh = rng.uniform(-1, 1, (210, 3)), rng.uniform(-1, 1, (150, 3))
    X = dev.points(ctx, Xh)
    L = dev.potrf(ctx, dev.kfill(ctx, sp, X, nugget=0.05))
    cov = dev.posterior_cov(ctx, sp, L, X, dev.points(ctx, Zh))
    model = orc.fit(s, Xh, None, 0.05)
    _, want = orc.posterior(s, model, Zh, compvar=2)
    assert cov.shape == (150, 150) and rel(cov, want) <= 1e-10
    _, var = dev.posterior(ctx, sp, L, X, None, dev.points(ctx, Zh), want_mean=False)
    assert rel(np.diag(cov), var) <= 1e-12


def test_many_small_chunks_and_tiny_training_set(dev, ctx, monkeypatch):
    rng = np.random.default_rng(9)
    s = dict(kind="matern32", rho=0.8, signalSize=1.0, d=2)
    sp = dev.KernelSpec(dev.K_MATERN32, 2, [0.8, 1.0])
    Xh = rng.uniform(-1, 1, (3, 2))      # 3 training points
    y = rng.standard_normal(3)
    Zh = rng.uniform(-1, 1, (1000, 2))
    X, Z = dev.points(ctx, Xh), dev.points(ctx, Zh)
    L = dev.potrf(ctx, dev.kfill(ctx, sp, X, nugget=0.01))
    alpha = dev.potrs(ctx, L, y)
    monkeypatch.setenv("GPX_CROSS_BYTES", "1")  # smallest possible chunk: 128 evaluation points at a time (8 chunks)
    m, v = dev.posterior(ctx, sp, L, X, alpha, Z)
    model = orc.fit(s, Xh, y, 0.01)
    mo, vo = orc.posterior(s, model, Zh)
    assert rel(m, mo) <= 1e-10 and rel(v, vo) <= 1e-10
    iv = dev.ivar(ctx, sp, L, X, Z)
    assert iv == pytest.approx(vo.mean(), rel=1e-10)
    best, costs = dev.greedy_ivar_step(ctx, sp, L, X, dev.points(ctx, Zh[:300]), dev.points(ctx, Zh[300:]), 0.01)
    want = [orc.ivar(s, np.vstack((Xh, Zh[j:j + 1])), Zh[300:], 0.01) for j in range(0, 300, 37)]
    assert rel(costs[0:300:37], want) <= 1e-10


def test_single_candidate_and_single_selection(dev, ctx):
    sp = dev.KernelSpec(dev.K_SE, 1, [0.3, 1.0])
    C = dev.points(ctx, np.array([[0.25]]))
    assert list(dev.greedy_var(ctx, sp, C, 1)) == [0]
    idx, ratios = dev.mi_greedy(ctx, sp, dev.points(ctx, np.array([[0.1], [0.5], [-0.7]])), 0.1, 1, 2)
    assert list(idx) == [2] and ratios.size == 0


def test_wide_and_narrow_leaf_multiply_strips_agree(dev, ctx, monkeypatch):
    """The in-place leaf multiplies use 64-wide strips beyond 8192 rows/columns and 32-wide ones below: the same evaluation
    in one 9000-point chunk (wide strips) and in 2048-point chunks (narrow strips) must agree bit for bit, and the factor
    of a 8448-row panel (wide right-multiply strips) must still solve K alpha = y."""
    rng = np.random.default_rng(77)
    n, d, m = 700, 3, 9000
    Xh, Zh = rng.uniform(-1, 1, (n, d)), rng.uniform(-1, 1, (m, d))
    sp = dev.KernelSpec(dev.K_SE, d, [0.5, 0.6, 0.7, 1.2])
    X, Z = dev.points(ctx, Xh), dev.points(ctx, Zh)
    L = dev.potrf(ctx, dev.kfill(ctx, sp, X, nugget=0.05))
    y = rng.standard_normal(n)
    alpha = dev.potrs(ctx, L, y)
    m1, v1 = dev.posterior(ctx, sp, L, X, alpha, Z)
    monkeypatch.setenv("GPX_CROSS_BYTES", str(768 * 8 * 2048))   # padded N = 768 -> chunks of 2048 evaluation points
    m2, v2 = dev.posterior(ctx, sp, L, X, alpha, Z)
    assert np.array_equal(m1, m2) and np.array_equal(v1, v2)
    monkeypatch.delenv("GPX_CROSS_BYTES")
    s = dict(kind="se", cl=[0.5, 0.6, 0.7], signalSize=1.2, d=d)
    mo, vo = orc.posterior(s, orc.fit(s, Xh, y, 0.05), Zh[:200])
    assert rel(m1[:200], mo) <= 1e-10 and rel(v1[:200], vo) <= 1e-10
    # tall panel: N = 16896 rows (> 32 * 256) under a 128-wide leaf exercises the 64-row right strips
    N2 = 16896
    X2h = rng.uniform(-1, 1, (N2, d))
    y2 = rng.standard_normal(N2)
    K2 = dev.kfill(ctx, sp, dev.points(ctx, X2h), nugget=0.1)
    rows = rng.choice(N2, 4, replace=False)
    Krows = np.stack([dev.kernel_eval(ctx, sp, X2h, X2h[r:r + 1]) for r in rows])
    Krows[np.arange(4), rows] += 0.1
    dev.potrf(ctx, K2)
    a2 = dev.potrs(ctx, K2, y2)
    assert np.max(np.abs(Krows @ a2 - y2[rows])) <= 1e-9 * np.max(np.abs(y2))


@pytest.mark.parametrize("n", [2047, 2177, 4000, 8200, 9000, 12929])
def test_round2_solver_paths_at_ragged_sizes(dev, ctx, n):
    """Sizes that are not multiples of the 1024-order inverse blocks or of the 4096-wide look-ahead panels (padded
    2048 / 2304 / 4096 / 8320 / 9088 / 13056; 2048 and 4096 = 128 * 2^q take the level-batched triangular inverse of gpx_potri): blocked look-ahead factorisation (N >= 8192), block-inverse potrs (ragged last
    block), out-of-place posterior solve (N >= 2048) -- against LAPACK on the host."""
    import scipy.linalg as sl
    rng = np.random.default_rng(n)
    d = 4
    X = rng.uniform(-1, 1, (n, d))
    y = np.sin(X.sum(1)) + 0.1 * rng.standard_normal(n)
    Z = rng.uniform(-1, 1, (333, d))
    sp = dev.KernelSpec(dev.K_MATERN52, d, [0.6, 1.1])
    dX, dZ = dev.points(ctx, X), dev.points(ctx, Z)
    Kd = dev.kfill(ctx, sp, dX, nugget=0.05)
    K = Kd.to_host()
    L = dev.potrf(ctx, Kd)
    c = sl.cho_factor(K, lower=True, check_finite=False)
    Lh = L.to_host(tri=1)
    assert rel(Lh, np.tril(c[0])) <= 1e-12
    alpha = dev.potrs(ctx, L, y)
    assert rel(alpha, sl.cho_solve(c, y, check_finite=False)) <= 1e-10
    assert rel(dev.potrs(ctx, L, y), alpha) == 0.0                      # second solve: cached block inverses, same bits
    Kxz = dev.kfill(ctx, sp, dX, Z=dZ).to_host()
    W = sl.solve_triangular(np.tril(c[0]), Kxz, lower=True, check_finite=False)
    mean, var = dev.posterior(ctx, sp, L, dX, alpha, dZ)
    assert rel(mean, Kxz.T @ alpha) <= 1e-10
    assert rel(var, 1.1 - np.sum(W * W, axis=0)) <= 1e-10
    assert abs(dev.logdet(ctx, L) - 2 * np.sum(np.log(np.diag(c[0])))) <= 1e-11 * n
    P = dev.potri(ctx, L).to_host(tri=2)                                  # triangular-aware explicit inverse
    assert rel(P @ K[:, :7], np.eye(n)[:, :7]) <= 1e-9
    del Kd, L
    ctx.trim()


def test_refactor_in_place_invalidates_block_inverses(dev, ctx):
    """The bench refills and refactors one matrix every step: the cached block inverses of the previous factor must not
    leak into the next solve."""
    rng = np.random.default_rng(99)
    n, d = 2500, 3
    X = dev.points(ctx, rng.uniform(-1, 1, (n, d)))
    y = rng.standard_normal(n)
    K = dev.DeviceMatrix.zeros(ctx, n, n)
    out = []
    for rho in (0.5, 0.9, 0.5):
        sp = dev.KernelSpec(dev.K_MATERN32, d, [rho, 1.0])
        dev.kfill_into(ctx, sp, X, K, nugget=0.1)
        dev.potrf(ctx, K)
        out.append(dev.potrs(ctx, K, y))
    assert np.array_equal(out[0], out[2]) and rel(out[1], out[0]) > 1e-3


@pytest.mark.parametrize("n,m", [(700, 300), (9000, 2500)])
def test_fit_ivar_equals_fit_then_ivar(dev, ctx, n, m):
    """gpx_fit_ivar (one call: factor in place, then the evaluation solve) against gpx_potrf followed by gpx_ivar on the same
    inputs: same factor and block inverses bit for bit, same IVAR, repeatable.  (The streamed variant of rounds 2-5 was
    removed in round 6: measured slower on one GPU.)"""
    rng = np.random.default_rng(n + m)
    d = 5
    X = dev.points(ctx, rng.uniform(-1, 1, (n, d)))
    Z = dev.points(ctx, rng.uniform(-1, 1, (m, d)))
    sp = dev.KernelSpec(dev.K_MATERN52, d, [0.6, 1.0])
    K1 = dev.potrf(ctx, dev.kfill(ctx, sp, X, nugget=0.05))
    iv1 = dev.ivar(ctx, sp, K1, X, Z)
    K2 = dev.kfill(ctx, sp, X, nugget=0.05)
    iv2 = dev.fit_ivar(ctx, sp, K2, X, Z)
    assert iv2 == iv1
    assert dev.logdet(ctx, K2) == dev.logdet(ctx, K1)
    y = rng.standard_normal(n)
    assert np.array_equal(dev.potrs(ctx, K2, y), dev.potrs(ctx, K1, y))    # factor and block inverses identical
    dev.kfill_into(ctx, sp, X, K2, nugget=0.05)
    assert dev.fit_ivar(ctx, sp, K2, X, Z) == iv2                           # repeatable bit for bit
    del K1, K2
    ctx.trim()


def test_fit_ivar_reports_non_positive_definite(dev, ctx):
    from gpexp_amd._lib import NotPositiveDefinite
    rng = np.random.default_rng(5)
    n, d = 8300, 3
    Xh = rng.uniform(-1, 1, (n, d))
    Xh[8000] = Xh[17]
    X = dev.points(ctx, Xh)
    Z = dev.points(ctx, rng.uniform(-1, 1, (256, d)))
    sp = dev.KernelSpec(dev.K_SE, d, [0.5, 0.5, 0.5, 1.0])
    K = dev.kfill(ctx, sp, X, nugget=0.0)
    dev.potrf_policy(ctx, 1e-13, False)
    try:
        with pytest.raises(NotPositiveDefinite):
            dev.fit_ivar(ctx, sp, K, X, Z)
    finally:
        dev.potrf_policy(ctx, 0.0, False)


def test_recorded_program_refuses_a_freed_matrix():
    """the rows of a recorded program carry raw matrix handles; a matrix freed after the recording must make the replay
    fail with an error (the context keeps the set of live handles), not run a kernel on released memory."""
    from gpexp_amd import device as dev, dist
    from gpexp_amd._lib import GpxError
    ctx = dev.context()
    a, b = dev.alloc_vector(ctx, 256), dev.alloc_vector(ctx, 256)
    prog = dist.Program()
    prog.emit(dist.OP["COPY"], (a, b), (0, 0, 128))
    prog.run(ctx)
    ctx.sync()
    b.free()
    with pytest.raises(GpxError, match="no longer alive"):
        prog.run(ctx)


def test_distributed_scratches_are_reserved_by_the_constructor():
    """the panel-solve and streamed-evaluation scratches are sized when the runner is built (gpx_dist2_reserve), so a
    step never reallocates them with work in flight; a second reserve with smaller sizes is a no-op."""
    from gpexp_amd import device as dev
    ctx = dev.context()
    assert ctx.lib.gpx_dist2_reserve(ctx.h, 512, 4, 4096) == 0
    assert ctx.lib.gpx_dist2_reserve(ctx.h, 256, 2, 1024) == 0
    assert ctx.lib.gpx_dist2_reserve(ctx.h, 100, 2, 1024) < 0     # nb must be a multiple of 128


def test_stamp_and_spin_until_pace_a_stream():
    """The paced replay's two debug entries: a spin that ends `us` after a wall-clock stamp taken on the stream -- at once when the
    moment has passed, and never a launch for us <= 0 -- is what holds a foreign delivery back (scripts/replay_comm.py)."""
    import time
    from gpexp_amd import device as dev
    ctx = dev.context()
    lib = ctx.lib
    assert lib.gpx_dbg_stamp(ctx.h, 5) == 0 and lib.gpx_dbg_spin_until(ctx.h, 5, 0) == 0
    ctx.sync()
    t0 = time.perf_counter()
    assert lib.gpx_dbg_stamp(ctx.h, 7) == 0
    assert lib.gpx_dbg_spin_until(ctx.h, 7, 20000) == 0        # 20 ms after the stamp
    ctx.sync()
    first = time.perf_counter() - t0
    assert 0.019 < first < 0.2, first
    t0 = time.perf_counter()
    assert lib.gpx_dbg_spin_until(ctx.h, 7, 20000) == 0        # that moment has passed: returns at once
    ctx.sync()
    assert time.perf_counter() - t0 < 0.01
    assert lib.gpx_dbg_stamp(ctx.h, 1024) < 0 and lib.gpx_dbg_spin_until(ctx.h, -1, 5) < 0
