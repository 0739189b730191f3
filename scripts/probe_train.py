"""GP.train breakdown (kfill / potrf / potrs / logdet) at several sizes."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gpexp_amd import device as dev
ctx = dev.context()
for N in (1024, 4096, 8192, 16384, 32768):
    rng = np.random.default_rng(N)
    d = 8
    X = dev.points(ctx, rng.uniform(-1, 1, (N, d)))
    y = rng.standard_normal(N)
    sp = dev.KernelSpec(dev.K_MATERN52, d, [0.5, 1.0])
    K = dev.DeviceMatrix.zeros(ctx, N, N)
    for it in range(3):
        ctx.sync(); t0 = time.perf_counter()
        dev.kfill_into(ctx, sp, X, K, nugget=0.1); ctx.sync(); t1 = time.perf_counter()
        dev.potrf(ctx, K); ctx.sync(); t2 = time.perf_counter()
        a = dev.potrs(ctx, K, y); t3 = time.perf_counter()
        ld = dev.logdet(ctx, K); t4 = time.perf_counter()
    print("N=%6d: kfill %7.3f ms  potrf %8.3f ms  potrs %7.3f ms  logdet %6.3f ms" % (N, 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2), 1e3 * (t4 - t3)), flush=True)
