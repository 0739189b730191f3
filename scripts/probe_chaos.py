"""Results of the multi-stream paths must not depend on how the streams drift: the same factorisation / step with and
without GPX_CHAOS (random 0.1-3 ms spins at every launch site).  Usage: GPX_CHAOS=<seed> python scripts/probe_chaos.py"""
import sys, os, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gpexp_amd import device as dev
ctx = dev.context()
for N in (8192, 16384, 32768):
    rng = np.random.default_rng(N)
    Xh = rng.uniform(-1, 1, (N, 8)); Zh = rng.uniform(-1, 1, (4096, 8)); y = rng.standard_normal(N)
    X, Z = dev.points(ctx, Xh), dev.points(ctx, Zh)
    sp = dev.KernelSpec(dev.K_MATERN52, 8, [0.5, 1.0])
    K = dev.DeviceMatrix.zeros(ctx, N, N)
    yd = dev.padded_vector(ctx, y); ad = dev.padded_vector(ctx, np.zeros(N))
    out = []
    for rep in range(2):
        dev.kfill_into(ctx, sp, X, K, nugget=0.1)
        dev.potrf(ctx, K)
        ctx.stream(1); dev.potrs_dev(ctx, K, yd, ad); ctx.stream(0)
        iv = dev.ivar(ctx, sp, K, X, Z)
        ld = dev.logdet(ctx, K)
        ctx.sync()
        a = ad.to_host()[:N, 0]
        out.append((ld, iv, hashlib.sha1(a.tobytes()).hexdigest()[:12]))
    print("N=%d chaos=%s: logdet %.17g ivar %.17g alpha %s  (repeat identical: %s)" % (N, os.environ.get("GPX_CHAOS", "-"), out[0][0], out[0][1], out[0][2], out[0] == out[1]), flush=True)
    del K; ctx.trim()
