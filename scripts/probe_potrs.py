"""potrs timing: first call after a factorisation (builds the block inverses) and steady state, device-vector form;
residual check against exact rows of K.  Usage: probe_potrs.py N [N ...]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gpexp_amd import device as dev
ctx = dev.context()
for N in [int(a) for a in sys.argv[1:]] or [4096, 8192, 32768]:
    rng = np.random.default_rng(N)
    Xh = rng.uniform(-1, 1, (N, 8))
    X = dev.points(ctx, Xh)
    y = rng.standard_normal(N)
    sp = dev.KernelSpec(dev.K_MATERN52, 8, [0.5, 1.0])
    K = dev.potrf(ctx, dev.kfill(ctx, sp, X, nugget=0.1))
    yd = dev.padded_vector(ctx, y)
    ad = dev.padded_vector(ctx, np.zeros(N))
    ctx.sync()
    t0 = time.perf_counter(); dev.potrs_dev(ctx, K, yd, ad); ctx.sync(); first = time.perf_counter() - t0
    ts = []
    for it in range(5):
        t0 = time.perf_counter(); dev.potrs_dev(ctx, K, yd, ad); ctx.sync(); ts.append(time.perf_counter() - t0)
    a = ad.to_host()[:N, 0]
    rows = rng.choice(N, 4, replace=False)
    Kr = np.stack([dev.kernel_eval(ctx, sp, Xh, Xh[r:r + 1]) for r in rows]); Kr[np.arange(4), rows] += 0.1
    res = np.max(np.abs(Kr @ a - y[rows]))
    best = min(ts)
    print("potrs N=%d: first %.3f ms (incl. block inverses), steady %.3f ms = %.2f TB/s (4 N^2 B), residual %.1e"
          % (N, 1e3 * first, 1e3 * best, 4.0 * N * N / best / 1e12, res), flush=True)
    del K
    ctx.trim()
