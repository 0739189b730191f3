import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gpexp_amd import device as dev
ctx = dev.context()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
rng = np.random.default_rng(N)
X = dev.points(ctx, rng.uniform(-1, 1, (N, 8)))
y = rng.standard_normal(N)
sp = dev.KernelSpec(dev.K_MATERN52, 8, [0.5, 1.0])
K = dev.potrf(ctx, dev.kfill(ctx, sp, X, nugget=0.1))
for it in range(3):
    t0 = time.perf_counter(); a = dev.potrs(ctx, K, y); t1 = time.perf_counter()
print("potrs N=%d: %.3f ms" % (N, 1e3 * (t1 - t0)))
