import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gpexp_amd import device as dev
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
M = int(sys.argv[2]) if len(sys.argv) > 2 else N
ctx = dev.context()
rng = np.random.default_rng(N)
X = dev.points(ctx, rng.uniform(-1, 1, (N, 8)))
Z = dev.points(ctx, rng.uniform(-1, 1, (M, 8)))
sp = dev.KernelSpec(2, 8, [0.5, 1.0])
K = dev.kfill(ctx, sp, X, nugget=0.1)
dev.potrf(ctx, K)
for it in range(2):
    t0 = time.perf_counter(); iv = dev.ivar(ctx, sp, K, X, Z); t1 = time.perf_counter()
    print("ivar N=%d M=%d: %.1f ms  %.1f TF/s  value %.8g" % (N, M, 1e3 * (t1 - t0), N * N * M / (t1 - t0) / 1e12, iv), flush=True)
