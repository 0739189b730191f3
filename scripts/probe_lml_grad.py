"""gpx_lml_grad (potri + fused traces over the N x N inverse) against the slab form (row slabs of K^-1 by two triangular solves
against the trailing factor, no N x N inverse): warm times per N.  Usage: probe_lml_grad.py [N ...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpexp_amd import device as dev
ctx = dev.context()
for N in [int(a) for a in sys.argv[1:]] or [8192, 16384, 32768, 65536]:
    d = 10
    rng = np.random.default_rng(N)
    Xh = rng.uniform(-1, 1, (N, d)); y = np.sin(2 * np.pi * Xh.sum(1) / d) + 0.3 * rng.standard_normal(N)
    sp = dev.KernelSpec(dev.K_SE, d, [0.5 + 0.03 * k for k in range(d)] + [1.0])
    X = dev.points(ctx, Xh)
    L = dev.potrf(ctx, dev.kfill(ctx, sp, X, nugget=0.1))
    alpha = dev.potrs(ctx, L, y)
    res = {}
    for name, fn in (("full", lambda: dev.lml_grad_full(ctx, sp, L, X, alpha)),) + tuple(
            ("slabs%d" % s, (lambda s=s: dev.lml_grad(ctx, sp, L, X, alpha, slabs=s))) for s in (4, 8, 16, 32)):
        ts = []
        for it in range(3):
            ctx.sync(); t0 = time.perf_counter(); g = fn(); ctx.sync(); ts.append(time.perf_counter() - t0)
        res[name] = (ts[0], min(ts[1:]), g)
    ref = res["full"][2]
    print("N=%d  " % N + "  ".join("%s: first %.3f s warm %.3f s (rel diff %.1e)" % (k, v[0], v[1], np.max(np.abs(v[2] - ref)) / np.max(np.abs(ref)))
                                   for k, v in res.items()), flush=True)
    del L; ctx.trim()
