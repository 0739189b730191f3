"""Kernel durations of the two assembly kernels in different neighbourhoods (run under rocprofv3 --kernel-trace and read
the trace in launch order): back to back, after a host sync, after the small kernels that end a fit."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gpexp_amd import device as dev
ctx = dev.context()
N = 32768
rng = np.random.default_rng(N)
X = dev.points(ctx, rng.uniform(-1, 1, (N, 8))); Z = dev.points(ctx, rng.uniform(-1, 1, (N, 8)))
y = dev.padded_vector(ctx, rng.standard_normal(N)); a = dev.padded_vector(ctx, np.zeros(N))
sp = dev.KernelSpec(dev.K_MATERN52, 8, [0.5, 1.0])
K = dev.DeviceMatrix.zeros(ctx, N, N); K2 = dev.DeviceMatrix.zeros(ctx, N, N)
sym = lambda: dev.kfill_into(ctx, sp, X, K, nugget=0.1)
cross = lambda: dev.kfill_into(ctx, sp, X, K2, Z=Z)
sym(); cross(); ctx.sync()
print("A: 4 symmetric back to back"); [sym() for _ in range(4)]; ctx.sync(); time.sleep(0.05)
print("B: 4 cross back to back"); [cross() for _ in range(4)]; ctx.sync(); time.sleep(0.05)
print("C: sym, sync, sleep 2 ms, x4")
for _ in range(4): sym(); ctx.sync(); time.sleep(0.002)
print("D: cross, sync, sleep 2 ms, x4")
for _ in range(4): cross(); ctx.sync(); time.sleep(0.002)
print("E: (sym, potrf, potrs, logdet) x3 without sync")
for _ in range(3): sym(); dev.potrf(ctx, K); dev.potrs_dev(ctx, K, y, a); dev.logdet(ctx, K)
ctx.sync(); time.sleep(0.05)
print("F: sym cross sym cross"); sym(); cross(); sym(); cross(); ctx.sync()
