"""Same launches timed three ways: HIP-event span (the library's profiler), host wall clock around launch + sync, and -- when
run under rocprofv3 --kernel-trace -- the kernel's own begin/end (scripts/fill_sequences_report.py)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gpexp_amd import device as dev
ctx = dev.context()
N = 32768
rng = np.random.default_rng(N)
X = dev.points(ctx, rng.uniform(-1, 1, (N, 8))); Z = dev.points(ctx, rng.uniform(-1, 1, (N, 8)))
sp = dev.KernelSpec(dev.K_MATERN52, 8, [0.5, 1.0])
K = dev.DeviceMatrix.zeros(ctx, N, N)
fills = {"kfill": lambda: dev.kfill_into(ctx, sp, X, K, nugget=0.1), "kcross": lambda: dev.kfill_into(ctx, sp, X, K, Z=Z)}
for f in fills.values(): f()
ctx.sync()
for name, f in fills.items():
    for rep in range(4):
        ctx.sync(); time.sleep(0.003)
        ctx.profile(True); ctx.profile_reset()
        t0 = time.perf_counter(); f(); t1 = time.perf_counter(); ctx.sync(); t2 = time.perf_counter()
        p = ctx.profile_get()[name]; ctx.profile(False)
        print("%-6s event span %.3f ms   launch call %.3f ms   launch+sync wall %.3f ms" % (name, p["ms"], 1e3 * (t1 - t0), 1e3 * (t2 - t0)), flush=True)
