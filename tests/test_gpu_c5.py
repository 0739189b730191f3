"""BASELINE config C5 building blocks on the GPU (-m gpu), through the C ABI: the slab form of the log-marginal gradient
(gpx_lml_grad_slab: the unit the multi-GPU gradient shards by) and the row-sharded greedy MI state (gpx_mi_*).  The multi-rank
runs of both are in tests/test_dist.py (gloo doubles on CPU, shared-GPU ranks on the GPU)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from gpexp_amd import device as dev
    return dev.context()


@pytest.mark.parametrize("n,d,parts", [(1500, 4, 1), (1500, 4, 3), (2100, 10, 5), (700, 2, 8), (130, 3, 2), (8300, 3, 3), (9216, 2, 4)])
def test_lml_grad_slabs_sum_to_the_full_gradient(ctx, n, d, parts):
    """Slabs of a partition of the rows add up to gpx_lml_grad (which forms the whole inverse): 1e-10, any partition; the
    work-balanced boundaries are multiples of 128 that cover the padded order (gp.py:444-466)."""
    from gpexp_amd import device as dev
    rng = np.random.default_rng(n + d)
    Xh = rng.uniform(-1, 1, (n, d))
    y = np.sin(2 * np.pi * Xh.sum(1) / d) + 0.3 * rng.standard_normal(n)
    spec = dev.KernelSpec(dev.K_SE, d, list(0.5 + 0.03 * np.arange(d)) + [1.2])
    X = dev.points(ctx, Xh)
    L = dev.potrf(ctx, dev.kfill(ctx, spec, X, nugget=0.1))
    alpha = dev.potrs(ctx, L, y)
    ref = dev.lml_grad(ctx, spec, L, X, alpha)
    b = dev.lml_grad_slab_bounds(n, parts)
    assert b[0] == 0 and b[-1] == (n + 127) // 128 * 128 and all(x % 128 == 0 for x in b) and b == sorted(b)
    sums = np.zeros(d + 2)
    for r0, r1 in zip(b[:-1], b[1:]):
        if r1 > r0:
            sums += dev.lml_grad_slab(ctx, spec, L, X, alpha, r0, r1)
    got = dev.lml_grad_from_sums(spec, sums)
    assert np.max(np.abs(got - ref)) <= 1e-10 * np.max(np.abs(ref)), (got, ref)
    # the single-GPU large-N form: one explicit L^-1, the lower triangle of K^-1 written over it (gpx_lml_grad_linv)
    lin = dev.lml_grad_from_sums(spec, dev.lml_grad_linv(ctx, spec, L, X, alpha))
    assert np.max(np.abs(lin - ref)) <= 1e-10 * np.max(np.abs(ref)), (lin, ref)
    # the ROWS form (round 5, gpx_lml_grad_rows: rows of L^-1, one solve + one accumulated SYRK + one trace per range): ranges
    # balanced on their actual cost, each worked in sub-slabs of equal height, add up to the same gradient -- and so does ONE range
    # over all rows, the single-GPU fallback between the linv form and the slab loop
    for nsub in (1, 3):
        rb = dev.lml_grad_rows_bounds(n, parts, nsub)
        assert rb[0] == 0 and rb[-1] == (n + 127) // 128 * 128 and all(x % 128 == 0 for x in rb) and rb == sorted(rb)
        rs = sum(dev.lml_grad_rows(ctx, spec, L, X, alpha, a, c, nsub) for a, c in zip(rb[:-1], rb[1:]) if c > a)
        rows = dev.lml_grad_from_sums(spec, rs)
        assert np.max(np.abs(rows - ref)) <= 1e-10 * np.max(np.abs(ref)), (nsub, rows, ref)
    one = dev.lml_grad_from_sums(spec, dev.lml_grad_rows(ctx, spec, L, X, alpha, 0, (n + 127) // 128 * 128, 4))
    assert np.max(np.abs(one - ref)) <= 1e-10 * np.max(np.abs(ref))
    # an uneven hand-made partition gives the same
    cuts = [0, 128, (n + 127) // 128 * 128]
    sums2 = sum(dev.lml_grad_slab(ctx, spec, L, X, alpha, a, c) for a, c in zip(cuts[:-1], cuts[1:]) if c > a)
    assert np.max(np.abs(dev.lml_grad_from_sums(spec, sums2) - ref)) <= 1e-10 * np.max(np.abs(ref))


def test_rows_bounds_balance_the_actual_cost():
    """A sub-slab [c0, c1) of the rows form costs (c1 - c0) c1^2: with equal-height sub-slabs inside cost-balanced ranges the ranks'
    shares agree to a few per cent and the total stays within 13 % of the integral at 8 ranges x 2 (10 % at 1 x 16)."""
    from gpexp_amd import device as dev
    n = 65536
    for parts, nsub, over in ((8, 2, 1.13), (1, 16, 1.10), (4, 4, 1.12)):
        b = dev.lml_grad_rows_bounds(n, parts, nsub)
        cost = []
        for a, c in zip(b[:-1], b[1:]):
            h = (c - a) / nsub
            cost.append(sum(h * (a + (j + 1) * h) ** 2 for j in range(nsub)))
        assert max(cost) / min(cost) < 1.06, (parts, nsub, cost)
        assert sum(cost) / (n ** 3 / 3.0) < over


def test_slab_bounds_balance_the_work():
    from gpexp_amd import device as dev
    n, parts = 65536, 8
    b = dev.lml_grad_slab_bounds(n, parts)
    work = [((n - r0) ** 3 - (n - r1) ** 3) / 3.0 for r0, r1 in zip(b[:-1], b[1:])]
    assert max(work) / (sum(work) / parts) < 1.10      # within 10 % of the mean at C5's size (bounds sit on 1024-row
    #                                                     block-inverse boundaries there)


@pytest.mark.parametrize("m,nsel,slices", [(300, 8, 1), (300, 8, 3), (517, 6, 4), (64, 5, 7)])
def test_row_sharded_mi_state_reproduces_mi_greedy(ctx, m, nsel, slices):
    """gpx_mi_* with the rows of the inverse cut into `slices` ranges (one state per range, as one per rank; the picked row
    travels through a buffer, the winners are merged first-max) == gpx_mi_greedy: identical picks AND identical ratios, bit
    for bit -- the down-date of a row uses only that row and the pivot row (experimentalDesign.py:259-285, 753-785)."""
    from gpexp_amd import device as dev, dist
    rng = np.random.default_rng(m)
    d = 3
    Ch = rng.uniform(-1, 1, (m, d))
    spec = dev.KernelSpec(dev.K_SE, d, [0.4, 0.6, 0.8, 2.0])
    Cp = dev.points(ctx, Ch)
    ref_idx, ref_ratio = dev.mi_greedy(ctx, spec, Cp, 1e-3, nsel)
    ranges = [dist.eval_slice(m, r, slices) for r in range(slices)]
    states = [dev.MiState(ctx, spec, Cp, 1e-3, nsel, 0, lo, hi) for lo, hi in ranges]
    row = dev.alloc_vector(ctx, m)
    picks, ratios = [0], []
    for cur in range(nsel - 1):
        for st in states:
            st.row(cur, row)            # only the owner of picks[-1] writes the buffer
        ctx.sync()
        res = [st.score(cur, row) for st in states]
        best = dist.merge_argmax([v for v, _ in res], [i for _, i in res])
        for st in states:
            st.select(cur + 1, best[1])
        picks.append(best[1])
        ratios.append(best[0])
    assert picks == list(ref_idx)
    assert ratios == list(ref_ratio)
