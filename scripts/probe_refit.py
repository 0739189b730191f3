"""f2: full factorisation vs prefix-reusing refit when only the last b design points change."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gpexp_amd import device as dev
ctx = dev.context()
for n, b in ((4096, 64), (8192, 64), (16384, 128), (16384, 512), (16384, 1024)):
    rng = np.random.default_rng(n)
    d = 8
    X = rng.uniform(-1, 1, (n, d))
    sp = dev.KernelSpec(dev.K_SE, d, list(0.4 + 0.05 * np.arange(d)) + [1.0])
    Xd = dev.points(ctx, X)
    L = dev.potrf(ctx, dev.kfill(ctx, sp, Xd, nugget=0.1))
    X2 = X.copy(); X2[n - b:] = rng.uniform(-1, 1, (b, d)); X2d = dev.points(ctx, X2)
    keep = ((n - b) // 128) * 128
    for it in range(2):
        ctx.sync(); t0 = time.perf_counter()
        Lf = dev.potrf(ctx, dev.kfill(ctx, sp, X2d, nugget=0.1)); ctx.sync(); t1 = time.perf_counter()
        Lr = dev.refit_rows(ctx, sp, X2d, 0.1, L, keep); ctx.sync(); t2 = time.perf_counter()
    print("N=%d, last %d points changed (keep %d): full fit %.2f ms, refit %.2f ms (%.1fx)" % (n, b, keep, 1e3 * (t1 - t0), 1e3 * (t2 - t1), (t1 - t0) / (t2 - t1)), flush=True)
