"""potrf timing (best of 4 after a warm-up; includes the block-inverse build and the device sync) + logdet for a
bit-level comparison between variants.  Usage: probe_potrf.py [N ...]; variants through GPX_POTRF_* in the environment."""
import sys, time
import numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpexp_amd import device as dev
ctx = dev.context()
for N in [int(a) for a in sys.argv[1:]] or [32768]:
    rng = np.random.default_rng(N)
    X = dev.points(ctx, rng.uniform(-1, 1, (N, 8)))
    sp = dev.KernelSpec(2, 8, [0.5, 1.0])
    K = dev.DeviceMatrix.zeros(ctx, N, N)
    ts = []
    for it in range(5):
        dev.kfill_into(ctx, sp, X, K, nugget=0.1); ctx.sync()
        t0 = time.perf_counter(); dev.potrf(ctx, K); ctx.sync(); ts.append(time.perf_counter() - t0)
    best = min(ts[1:])
    yv = np.sin(np.arange(N) * 0.37)
    print("potrf N=%d: %.1f ms  %.1f TF/s  logdet=%.15g  yTa=%.15g  [%s]" % (N, 1e3 * best, N**3 / 3 / best / 1e12, dev.logdet(ctx, K),
          float(yv @ dev.potrs(ctx, K, yv)),
          " ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items()) if k.startswith("GPX_POTRF"))), flush=True)
    del K
    ctx.trim()
