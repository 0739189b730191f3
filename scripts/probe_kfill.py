import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gpexp_amd import device as dev
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
d = int(sys.argv[2]) if len(sys.argv) > 2 else 8
kind = int(sys.argv[3]) if len(sys.argv) > 3 else 2
ctx = dev.context()
rng = np.random.default_rng(N)
X = dev.points(ctx, rng.uniform(-1, 1, (N, d)))
Z = dev.points(ctx, rng.uniform(-1, 1, (N, d)))
hyp = {0: list(0.4 + 0.05 * np.arange(d)) + [1.0], 1: [0.5, 1.0], 2: [0.5, 1.0], 3: list(0.2 + 0.02 * np.arange(d))}[kind]
sp = dev.KernelSpec(kind, d, hyp)
K = dev.DeviceMatrix.zeros(ctx, N, N)
for sym in (True, False):
    for it in range(3):
        ctx.profile(True); ctx.profile_reset()
        dev.kfill_into(ctx, sp, X, K, Z=None if sym else Z, nugget=0.1 if sym else 0.0)
        p = ctx.profile_get()["kfill" if sym else "kcross"]; ctx.profile(False)
    print("kfill kind=%d d=%d N=%d %s: %.3f ms  %.2f TB/s" % (kind, d, N, "symmetric" if sym else "rect", p["ms"], p["bytes"] / p["ms"] / 1e9), flush=True)
