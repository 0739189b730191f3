import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gpexp_amd import device as dev
ctx = dev.context()
N, d = 32768, 8
rng = np.random.default_rng(N)
X = dev.points(ctx, rng.uniform(-1, 1, (N, d)))
sp = dev.KernelSpec(2, d, [0.5, 1.0])
L = dev.potrf(ctx, dev.kfill(ctx, sp, X, nugget=0.1))
for m in (4096, 8192, 16384):
    Z = dev.points(ctx, rng.uniform(-1, 1, (m, d)))
    dev.posterior(ctx, sp, L, X, None, Z, want_mean=False)
    ts = []
    for _ in range(3):
        ctx.sync(); t0 = time.perf_counter(); dev.posterior(ctx, sp, L, X, None, Z, want_mean=False); ctx.sync(); ts.append(1e3 * (time.perf_counter() - t0))
    print("IVAR slice M=%d at N=%d: %.1f ms (%.1f TF/s)" % (m, N, min(ts), N * N * m / min(ts) / 1e9))
