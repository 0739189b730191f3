"""All kernel kinds, symmetric and rectangular, each launch after a sync + 2 ms pause; run under rocprofv3 --kernel-trace
(durations by template instance: scripts/fill_sequences_report.py)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gpexp_amd import device as dev
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
d = 8
ctx = dev.context()
rng = np.random.default_rng(N)
X = dev.points(ctx, rng.uniform(-1, 1, (N, d))); Z = dev.points(ctx, rng.uniform(-1, 1, (N, d)))
K = dev.DeviceMatrix.zeros(ctx, N, N)
for kind in (0, 1, 2, 3):
    hyp = {0: list(0.4 + 0.05 * np.arange(d)) + [1.0], 1: [0.5, 1.0], 2: [0.5, 1.0], 3: list(0.2 + 0.02 * np.arange(d))}[kind]
    sp = dev.KernelSpec(kind, d, hyp)
    for sym in (True, False):
        for it in range(3):
            dev.kfill_into(ctx, sp, X, K, Z=None if sym else Z, nugget=0.1 if sym else 0.0); ctx.sync(); time.sleep(0.002)
