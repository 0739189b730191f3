"""Does a power-of-two leading dimension of the B operand (n x k, used transposed) or of C slow the GEMM down?
NT product m x n x k with ld(B), ld(C) either exactly n / k (pad=False) or skewed (+16, pad=True)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gpexp_amd import device as dev
ctx = dev.context()
m, n, k = 28672, 4096, 4096
rng = np.random.default_rng(1)
A = dev.DeviceMatrix.from_host(ctx, rng.standard_normal((m, k)), pad=True)
Bh = rng.standard_normal((n, k))
for padB in (False, True):
    for padC in (False, True):
        B = dev.DeviceMatrix.from_host(ctx, Bh, pad=padB)
        Cm = dev.DeviceMatrix.zeros(ctx, m, n, pad=padC)
        ts = []
        for it in range(6):
            ctx.sync(); t0 = time.perf_counter(); dev.dbg_gemm(ctx, A, B, Cm, 1, 0); ctx.sync(); ts.append(time.perf_counter() - t0)
        t = min(ts[1:])
        print("ldB %s ldC %s: %.3f ms  %.1f TF/s" % ("4112" if padB else "4096", "4112" if padC else "4096", 1e3 * t, 2.0 * m * n * k / t / 1e12), flush=True)
