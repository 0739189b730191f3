"""The triangular-operand product of the panel solves, X (m x n) times the transposed n-order block inverse, three ways:
dense GEMM, triangular-operand GEMM (per-tile k cut), and their cost on the useful (triangular) flop count m n (n + 128).
Usage: probe_gemm_tri.py [m,n ...]   (GPX_LD_SKEW=0 in the environment: row strides that are exact powers of two)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gpexp_amd import device as dev
ctx = dev.context()
rng = np.random.default_rng(1)
shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or \
    [(4096, 1024), (8192, 1024), (16384, 1024), (24576, 1024), (24576, 2048), (24576, 512), (24576, 4096), (8192, 4096)]
for (m, n) in shapes:
    Ah = rng.standard_normal((m, n)); Bh = np.tril(rng.standard_normal((n, n)))
    A = dev.DeviceMatrix.from_host(ctx, Ah, pad=True); B = dev.DeviceMatrix.from_host(ctx, Bh, pad=True)
    Cm = dev.DeviceMatrix.zeros(ctx, m, n)
    res = {}
    for name, fn in (("dense", lambda: dev.dbg_gemm(ctx, A, B, Cm, 1, 0)), ("tri", lambda: dev.dbg_gemm_tri(ctx, A, B, Cm, 1, 0, 2))):
        ts = []
        for it in range(6):
            ctx.sync(); t0 = time.perf_counter(); fn(); ctx.sync(); ts.append(time.perf_counter() - t0)
        res[name] = min(ts[1:])
    useful = float(m) * n * (n + 128)
    print("m=%5d n=%4d ld=%d: dense %.3f ms = %.1f TF/s (2mnk)  |  tri %.3f ms = %.1f TF/s on the useful flops (dense: %.1f)"
          % (m, n, int(os.environ.get("GPX_LD_SKEW", "16")) + n, 1e3 * res["dense"], 2.0 * m * n * n / res["dense"] / 1e12, 1e3 * res["tri"],
             useful / res["tri"] / 1e12, useful / res["dense"] / 1e12), flush=True)
