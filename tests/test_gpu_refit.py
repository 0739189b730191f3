"""refits that only change the trailing design points (gpx_refit_rows), through the C ABI, against the
full assemble + factor path and the oracle."""
import numpy as np
import pytest

from oracle import gpexp_oracle as orc

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


@pytest.fixture(scope="module")
def dev():
    from gpexp_amd import device
    return device


@pytest.fixture(scope="module")
def ctx(dev):
    return dev.context()


@pytest.mark.parametrize("n_old,n_new,keep", [(300, 300, 128), (300, 300, 256), (700, 700, 512), (500, 650, 384),
                                              (650, 500, 384), (1025, 1025, 1024), (400, 400, 0),
                                              (512, 512, 512), (600, 512, 512)])
@pytest.mark.parametrize("kind", ["se", "matern52"])
def test_refit_rows_equals_full_factorisation(dev, ctx, n_old, n_new, keep, kind):
    rng = np.random.default_rng(n_old + 7 * n_new + keep)
    d = 3
    Xo = rng.uniform(-1, 1, (n_old, d))
    Xn = rng.uniform(-1, 1, (n_new, d))
    Xn[:keep] = Xo[:keep]                       # shared leading points; everything after them differs
    if kind == "se":
        sp = dev.KernelSpec(dev.K_SE, d, [0.5, 0.7, 0.9, 1.2])
        s = dict(kind="se", cl=[0.5, 0.7, 0.9], signalSize=1.2, d=d)
    else:
        sp = dev.KernelSpec(dev.K_MATERN52, d, [0.8, 1.1])
        s = dict(kind="matern52", rho=0.8, signalSize=1.1, d=d)
    nug = 0.05
    L_old = dev.potrf(ctx, dev.kfill(ctx, sp, dev.points(ctx, Xo), nugget=nug))
    Xd = dev.points(ctx, Xn)
    L_ref = dev.potrf(ctx, dev.kfill(ctx, sp, Xd, nugget=nug))
    L_new = dev.refit_rows(ctx, sp, Xd, nug, L_old, keep)
    a, b = np.tril(L_new.to_host()), np.tril(L_ref.to_host())
    assert rel(a, b) <= 1e-12
    K = orc.cov_matrix(s, Xn, nug, row_loop=False)
    assert rel(a @ a.T, K) <= 1e-13                                     # it factors the right matrix
    y = rng.standard_normal(n_new)
    assert rel(dev.potrs(ctx, L_new, y), dev.potrs(ctx, L_ref, y)) <= 1e-10
    assert dev.logdet(ctx, L_new) == pytest.approx(dev.logdet(ctx, L_ref), rel=1e-12)
    # the old factor is untouched (shallow GP copies may still hold it)
    assert rel(np.tril(L_old.to_host()), np.linalg.cholesky(orc.cov_matrix(s, Xo, nug, row_loop=False))) <= 1e-11


@pytest.mark.parametrize("n_old,n_new,keep", [(5000, 5000, 4608), (6200, 6100, 5120), (9000, 9000, 8064)])
def test_refit_rows_few_rows_against_a_large_kept_factor(dev, ctx, n_old, n_new, keep):
    """keep >= 4096: the strip solve goes through the kept factor's 1024-order block inverses, its few-row products as slices of
    the k range (chol.hip trsm_few_rows), the trailing update as 1024-wide batches, and the copy of the kept rows runs on a side
    stream underneath.  Same factor as a full factorisation; the kept rows arrive complete (the copy is joined)."""
    rng = np.random.default_rng(n_old + keep)
    d = 4
    Xo = rng.uniform(-1, 1, (n_old, d))
    Xn = rng.uniform(-1, 1, (n_new, d))
    Xn[:keep] = Xo[:keep]
    sp = dev.KernelSpec(dev.K_SE, d, [0.5, 0.6, 0.7, 0.8, 1.1])
    nug = 0.1
    L_old = dev.potrf(ctx, dev.kfill(ctx, sp, dev.points(ctx, Xo), nugget=nug))
    old_before = np.tril(L_old.to_host())
    Xd = dev.points(ctx, Xn)
    L_ref = dev.potrf(ctx, dev.kfill(ctx, sp, Xd, nugget=nug))
    for _ in range(2):          # (second call: the block inverses of the old factor are cached by then)
        L_new = dev.refit_rows(ctx, sp, Xd, nug, L_old, keep)
        a, b = np.tril(L_new.to_host()), np.tril(L_ref.to_host())
        assert np.array_equal(a[:keep], old_before[:keep, :n_new])       # the kept rows: copied bit for bit
        assert rel(a, b) <= 1e-11
        y = rng.standard_normal(n_new)
        assert rel(dev.potrs(ctx, L_new, y), dev.potrs(ctx, L_ref, y)) <= 1e-9
        assert dev.logdet(ctx, L_new) == pytest.approx(dev.logdet(ctx, L_ref), rel=1e-12)
    assert np.array_equal(np.tril(L_old.to_host()), old_before)


def test_refit_rows_with_per_point_nugget_and_rejects_bad_keep(dev, ctx):
    rng = np.random.default_rng(5)
    n, d, keep = 520, 2, 256
    X = rng.uniform(-1, 1, (n, d))
    nug = 0.01 + 0.05 * rng.uniform(size=n)
    sp = dev.KernelSpec(dev.K_SE, d, [0.6, 0.8, 1.0])
    Xd = dev.points(ctx, X)
    L_old = dev.potrf(ctx, dev.kfill(ctx, sp, Xd, nugget=nug))
    X2 = X.copy(); X2[keep:] = rng.uniform(-1, 1, (n - keep, d))
    nug2 = nug.copy(); nug2[keep:] *= 2.0
    X2d = dev.points(ctx, X2)
    L_ref = dev.potrf(ctx, dev.kfill(ctx, sp, X2d, nugget=nug2))
    L_new = dev.refit_rows(ctx, sp, X2d, nug2, L_old, keep)
    assert rel(np.tril(L_new.to_host()), np.tril(L_ref.to_host())) <= 1e-12
    with pytest.raises(Exception):
        dev.refit_rows(ctx, sp, X2d, nug2, L_old, 100)       # not a multiple of 128
    with pytest.raises(Exception):
        dev.refit_rows(ctx, sp, X2d, nug2, L_old, 1024)      # beyond both point sets


def test_gp_reuses_the_leading_factor_transparently(dev, ctx):
    from gpExp.kernels import KernelSquaredExponential
    from gpExp.gp import GP
    rng = np.random.default_rng(11)
    d, n, p = 2, 800, 650
    X1 = rng.uniform(-1, 1, (n, d))
    X2 = X1.copy(); X2[p:] = rng.uniform(-1, 1, (n - p, d))
    Z = rng.uniform(-1, 1, (300, d))
    k = KernelSquaredExponential([0.4, 0.6], 1.3, d)
    g = GP(k, 1e-2)
    g.addNodesAndComputeCovariance(X1)
    assert g._reusable_rows(X2, 1e-2, g.kernel._spec()) == 640
    g.addNodesAndComputeCovariance(X2)                         # takes the gpx_refit_rows path
    fresh = GP(k, 1e-2); fresh.reuseFactor = False
    fresh.addNodesAndComputeCovariance(X2)
    assert rel(g.evaluateVariance(Z), fresh.evaluateVariance(Z)) <= 1e-10
    # growing design across a padding boundary
    X3 = np.vstack((X2, rng.uniform(-1, 1, (150, d))))
    g.addNodesAndComputeCovariance(X3)
    fresh.addNodesAndComputeCovariance(X3)
    assert rel(g.evaluateVariance(Z), fresh.evaluateVariance(Z)) <= 1e-10
    # a hyper-parameter change invalidates the cached factor
    g.kernel.updateHyperParameters({"cl0": 0.55, "cl1": 0.6, "signalSize": 1.3})   # (replaces the whole dict, kernels.py:44-47)
    fresh.kernel.updateHyperParameters({"cl0": 0.55, "cl1": 0.6, "signalSize": 1.3})
    assert g._reusable_rows(X3, 1e-2, g.kernel._spec()) == 0
    g.addNodesAndComputeCovariance(X3); fresh.addNodesAndComputeCovariance(X3)
    assert rel(g.evaluateVariance(Z), fresh.evaluateVariance(Z)) <= 1e-10
    # the kept factor itself when the very same fit is asked for again -- also by a likelihood evaluation, which never stores one
    y3 = np.sin(X3.sum(1))
    g.train(X3, y3)
    kept = g._fcache[3]
    assert g._cached_factor(X3, 1e-2, g.kernel._spec()) is kept and g._cached_factor(X3.copy(), 1e-2, g.kernel._spec()) is kept
    assert g._cached_factor(X3[:-1], 1e-2, g.kernel._spec()) is None and g._cached_factor(X3 + 1e-16, 2e-2, g.kernel._spec()) is None
    fresh.train(X3, y3)
    assert g.computeLogLike(X3, y3) == pytest.approx(fresh.computeLogLike(X3, y3), rel=1e-12)
    assert g._fcache[3] is kept                                   # consulted, not replaced
    y4 = y3 + 0.1                                                  # other data, same points: still the kept factor
    assert g.computeLogLike(X3, y4) == pytest.approx(fresh.computeLogLike(X3, y4), rel=1e-12)
    # so does a different noise
    g.noise = 2e-2
    assert g._reusable_rows(X3, g.noise, g.kernel._spec()) == 0 and g._cached_factor(X3, g.noise, g.kernel._spec()) is None


def test_greedy_with_derivatives_driver_pins_earlier_batches(dev, ctx, capsys):
    """experimentalDesign.py:694-751 (no continuation): batches of new points, earlier ones pinned by bounds."""
    from gpExp.kernels import KernelSquaredExponential
    from gpExp.gp import GP
    from gpExp.approximation import Space
    from gpExp.experimentalDesign import costFunctionGP_IVAR, ExperimentalDesignGreedyWithDerivatives
    rng = np.random.default_rng(3)
    d = 2
    mc = rng.uniform(-1, 1, (400, d))
    space = Space(d, lambda size: rng.uniform(-1, 1, size), lambda p: np.all(np.abs(p) < 1.0, axis=1) * 0.25)
    gp = GP(KernelSquaredExponential([0.5, 0.5], 1.0, d), 1e-4)
    cf = costFunctionGP_IVAR(gp, 6, space, mcPoints=mc)
    des = ExperimentalDesignGreedyWithDerivatives(cf, 6, 3, d)
    first = ExperimentalDesignGreedyWithDerivatives(costFunctionGP_IVAR(gp, 3, space, mcPoints=mc), 3, 3, d).begin()
    pts = des.begin()
    capsys.readouterr()
    assert pts.shape == (6, d) and first.shape == (3, d)
    assert rel(pts[:3], first) <= 1e-9                       # the first batch stays where its own optimisation left it
    assert np.all(np.abs(pts) <= 1.0 + 1e-9)
    c3 = costFunctionGP_IVAR(gp, 3, space, mcPoints=mc).evaluate(first)
    c6 = costFunctionGP_IVAR(gp, 6, space, mcPoints=mc).evaluate(pts)
    assert c6 < c3


def cf_last_cost(des, pts, mc, space, kern):
    """The cost of a design through a cost object that DOES reuse (kept factor / kept solve): what the driver's optimiser saw."""
    from gpExp.gp import GP
    from gpExp.experimentalDesign import costFunctionGP_IVAR
    c = costFunctionGP_IVAR(GP(kern(), 1e-2), len(pts), space, mcPoints=mc)
    c.evaluate(pts)                                     # full solve
    moved = pts.copy(); moved[-3:] *= 0.99
    c.evaluate(moved)                                   # incremental
    return c.evaluate(pts)                              # incremental back to the design


def test_batch_driver_over_the_free_points_only(dev, ctx, capsys):
    """`freeVariablesOnly` (opt-in): the pinned points leave SLSQP's variable vector; cost and gradient still see the whole design.
    Small: the same design as the reference's all-variables run.  Large (1152 pinned + 128 new points): every cost evaluation of
    the batch re-solves the moved rows of the kept forward solve only (gpx_ivar_update), every gradient is the free points' alone
    (gpx_ivar_grad_rows); the pinned points stay, the cost goes down from the greedy start."""
    from gpExp.kernels import KernelSquaredExponential
    from gpExp.gp import GP
    from gpExp.approximation import Space
    from gpExp.experimentalDesign import costFunctionGP_IVAR, ExperimentalDesignGreedyWithDerivatives
    rng = np.random.default_rng(5)
    d = 2
    mc = rng.uniform(-1, 1, (400, d))
    space = Space(d, lambda size: rng.uniform(-1, 1, size), lambda p: np.all(np.abs(p) < 1.0, axis=1) * 0.25)
    gp = GP(KernelSquaredExponential([0.5, 0.5], 1.0, d), 1e-4)
    start = ExperimentalDesignGreedyWithDerivatives(costFunctionGP_IVAR(gp, 3, space, mcPoints=mc), 3, 3, d).begin()
    runs = []
    for free_only in (False, True):
        des = ExperimentalDesignGreedyWithDerivatives(costFunctionGP_IVAR(gp, 6, space, mcPoints=mc), 6, 3, d)
        des.freeVariablesOnly = free_only
        runs.append(des.begin(startValues=start))
    capsys.readouterr()
    assert np.array_equal(runs[1][:3], start) and rel(runs[0][:3], start) <= 1e-12
    c = [costFunctionGP_IVAR(gp, 6, space, mcPoints=mc).evaluate(r) for r in runs]
    assert c[1] == pytest.approx(c[0], rel=1e-4)                                    # SLSQP stops at acc = 1e-6 either way
    # large (d = 5: in two dimensions 1152 points leave SLSQP nothing to do)
    n0, nb, d = 1152, 128, 5
    space = Space(d, lambda size: rng.uniform(-1, 1, size), lambda p: np.all(np.abs(p) < 1.0, axis=1) * 0.25)
    mc = rng.uniform(-1, 1, (4096, d))
    pinned = rng.uniform(-1, 1, (n0, d))
    kern = lambda: KernelSquaredExponential([0.5] * d, 1.0, d)
    gp = GP(kern(), 1e-2)
    cf = costFunctionGP_IVAR(gp, n0 + nb, space, mcPoints=mc)
    des = ExperimentalDesignGreedyWithDerivatives(cf, n0 + nb, nb, d)
    des.freeVariablesOnly = True
    calls = {"update": 0, "rows": 0}
    upd, rows = dev.ivar_update, dev.ivar_grad_rows
    dev.ivar_update = lambda *a, **k: (calls.__setitem__("update", calls["update"] + 1), upd(*a, **k))[1]
    dev.ivar_grad_rows = lambda *a, **k: (calls.__setitem__("rows", calls["rows"] + 1), rows(*a, **k))[1]
    try:
        pts = des.begin(startValues=pinned)
    finally:
        dev.ivar_update, dev.ivar_grad_rows = upd, rows
    capsys.readouterr()
    assert pts.shape == (n0 + nb, d) and np.array_equal(pts[:n0], pinned)
    assert calls["update"] >= 5 and calls["rows"] >= 2                         # SLSQP iterates here: line searches + gradients
    fresh = costFunctionGP_IVAR(GP(kern(), 1e-2), n0 + nb, space, mcPoints=mc)
    fresh.gaussianProcess.reuseFactor = False
    c_end = fresh.evaluate(pts)
    assert c_end == pytest.approx(cf_last_cost(des, pts, mc, space, kern), rel=1e-10)
    c_pinned_only = costFunctionGP_IVAR(GP(kern(), 1e-2), n0, space, mcPoints=mc).evaluate(pinned)
    assert c_end < c_pinned_only


def test_refit_factor_with_nan_poisoned_upper_part_feeds_every_consumer(dev):
    """gpx_refit_rows copies the old factor's LOWER triangle only, so the new factor's strict upper part is whatever the
    pool handed out.  Under GPX_ALLOC_GUARD=2 every pooled block starts NaN-filled: the refit factor's upper part IS NaN here,
    and every consumer of a factor -- solve, log det, posterior, the L^-1 gradient form, the explicit inverse, a further refit --
    must give what it gives on a factor from gpx_potrf (include/gpx.h states the invariant)."""
    import os
    old = os.environ.get("GPX_ALLOC_GUARD")
    os.environ["GPX_ALLOC_GUARD"] = "2"
    try:
        ctx = dev.Context(0)                   # the guard mode is read when a context is created
    finally:
        if old is None:
            os.environ.pop("GPX_ALLOC_GUARD", None)
        else:
            os.environ["GPX_ALLOC_GUARD"] = old
    try:
        rng = np.random.default_rng(99)
        n, d, keep = 1700, 3, 1024
        sp = dev.KernelSpec(dev.K_SE, d, [0.5, 0.7, 0.9, 1.2])
        Xo = rng.uniform(-1, 1, (n, d))
        Xn = Xo.copy()
        Xn[keep:] = rng.uniform(-1, 1, (n - keep, d))
        y = rng.standard_normal(n)
        Z = dev.points(ctx, rng.uniform(-1, 1, (257, d)))
        L_old = dev.potrf(ctx, dev.kfill(ctx, sp, dev.points(ctx, Xo), nugget=0.05))
        Xd = dev.points(ctx, Xn)
        L_ref = dev.potrf(ctx, dev.kfill(ctx, sp, Xd, nugget=0.05))
        L_new = dev.refit_rows(ctx, sp, Xd, 0.05, L_old, keep)
        assert np.isnan(np.triu(L_new.to_host()[:n, :n], 600)).any(), "the guard mode did not poison the refit factor's upper part"
        a_new, a_ref = dev.potrs(ctx, L_new, y), dev.potrs(ctx, L_ref, y)
        assert np.all(np.isfinite(a_new)) and rel(a_new, a_ref) <= 1e-10
        assert dev.logdet(ctx, L_new) == pytest.approx(dev.logdet(ctx, L_ref), rel=1e-12)
        m1, v1 = dev.posterior(ctx, sp, L_new, Xd, a_new, Z)
        m0, v0 = dev.posterior(ctx, sp, L_ref, Xd, a_ref, Z)
        assert rel(m1, m0) <= 1e-10 and rel(v1, v0) <= 1e-9
        g1, g0 = dev.lml_grad_linv(ctx, sp, L_new, Xd, a_new), dev.lml_grad_linv(ctx, sp, L_ref, Xd, a_ref)
        assert np.all(np.isfinite(g1)) and rel(g1, g0) <= 1e-9
        P1, P0 = dev.potri(ctx, L_new).to_host(tri=2)[:n, :n], dev.potri(ctx, L_ref).to_host(tri=2)[:n, :n]
        assert np.all(np.isfinite(P1)) and rel(P1, P0) <= 1e-9
        Xn2 = Xn.copy()
        Xn2[1536:] = rng.uniform(-1, 1, (n - 1536, d))
        X2 = dev.points(ctx, Xn2)
        L2 = dev.refit_rows(ctx, sp, X2, 0.05, L_new, 1536)          # a further refit on the refit factor
        L2_ref = dev.potrf(ctx, dev.kfill(ctx, sp, X2, nugget=0.05))
        assert rel(np.tril(L2.to_host()[:n, :n]), np.tril(L2_ref.to_host()[:n, :n])) <= 1e-11
        assert ctx.guard_violations() == 0
    finally:
        ctx.close()
