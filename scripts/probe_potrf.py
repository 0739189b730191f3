import sys, time
import numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpexp_amd import device as dev
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
ctx = dev.context()
rng = np.random.default_rng(N)
X = dev.points(ctx, rng.uniform(-1, 1, (N, 8)))
sp = dev.KernelSpec(2, 8, [0.5, 1.0])
K = dev.DeviceMatrix.zeros(ctx, N, N)
for it in range(2):
    dev.kfill_into(ctx, sp, X, K, nugget=0.1); ctx.sync()
    t0 = time.perf_counter(); dev.potrf(ctx, K); t1 = time.perf_counter()
    print("potrf N=%d: %.1f ms  %.1f TF/s" % (N, 1e3 * (t1 - t0), N**3 / 3 / (t1 - t0) / 1e12), flush=True)
