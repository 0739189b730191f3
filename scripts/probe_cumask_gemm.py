"""How fast is a large GEMM on the CU-masked stream, for different sets of reserved CUs per XCD?  (The look-ahead chunk of
potrf ran ~30 % slower than its CU share with CUs 0-3 reserved.)  One context per pattern (GPX_CUMASK_RESERVE is read at
gpx_create).  Usage: probe_cumask_gemm.py [pattern ...]   pattern = comma-separated CU indices, '-' = none"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gpexp_amd import device as dev

pats = sys.argv[1:] or ["0,1,2,3", "0,8,16,24", "0,2,4,6", "0,1", "0,16", "0,1,2,3,4,5,6,7", "0,4,8,12,16,20,24,28", "28,29,30,31"]
shapes = [(8192, 8192, 4096, 1), (16384, 16384, 4096, 1), (20480, 4096, 4096, 0)]
rng = np.random.default_rng(1)
blk = rng.standard_normal((1024, 4096))
for pat in pats:
    os.environ["GPX_CUMASK_RESERVE"] = "999" if pat == "-" else pat
    ctx = dev.Context(0) if hasattr(dev, "Context") else dev.context()
    for (m, n, k, low) in shapes:
        A = dev.DeviceMatrix.from_host(ctx, np.tile(blk, (m // 1024, 1)))
        B = A if n == m else dev.DeviceMatrix.from_host(ctx, np.tile(blk, (n // 1024, 1)))
        C = dev.DeviceMatrix.zeros(ctx, m, n)
        res = {}
        for which in (0, 3):
            ctx.stream(which)
            ts = []
            for rep in range(4):
                ctx.sync(); t0 = time.perf_counter(); dev.dbg_gemm(ctx, A, B, C, 1, 1, low); ctx.sync(); ts.append(time.perf_counter() - t0)
            res[which] = min(ts[1:])
        ctx.stream(0)
        fl = (m * n * k if low else 2.0 * m * n * k)
        print("reserve [%s] m=%d n=%d k=%d lower=%d: main %.3f ms %.1f TF/s | masked %.3f ms %.1f TF/s  ratio %.3f"
              % (pat, m, n, k, low, 1e3 * res[0], fl / res[0] / 1e12, 1e3 * res[3], fl / res[3] / 1e12, res[0] / res[3]), flush=True)
        del A, B, C
    ctx.trim()
    del ctx
