"""Triangular-operand GEMM modes against the dense kernel on the same shapes: time and numerics."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gpexp_amd import device as dev
ctx = dev.context()
rng = np.random.default_rng(1)
for (m, n) in ((4096, 4096), (28672, 4096), (28672, 1024), (1024, 32768)):
    tri = 1 if m < n else 2
    k = m if tri == 1 else n
    Ah = rng.standard_normal((m, k)); Bh = rng.standard_normal((n, k) if tri == 2 else (k, n))
    if tri == 2: Bh = np.tril(Bh)
    else: Ah = np.tril(Ah)
    A = dev.DeviceMatrix.from_host(ctx, Ah, pad=True); B = dev.DeviceMatrix.from_host(ctx, Bh, pad=True)
    Cm = dev.DeviceMatrix.zeros(ctx, m, n)
    res = {}
    for name, fn in (("dense", lambda: dev.dbg_gemm(ctx, A, B, Cm, 1 if tri == 2 else 0, 0)),
                     ("tri", lambda: dev.dbg_gemm_tri(ctx, A, B, Cm, 1 if tri == 2 else 0, 0, tri))):
        ts = []
        for it in range(5):
            ctx.sync(); t0 = time.perf_counter(); fn(); ctx.sync(); ts.append(time.perf_counter() - t0)
        res[name] = (min(ts[1:]), Cm.to_host()[:256, :].copy() if m * n <= 1 << 27 else None)
    err = None if res["tri"][1] is None else float(np.max(np.abs(res["tri"][1] - res["dense"][1])))
    print("m=%d n=%d k=%d tri=%d: dense %.3f ms, tri %.3f ms (%.2fx)  max|diff| %s" % (m, n, k, tri, 1e3 * res["dense"][0],
          1e3 * res["tri"][0], res["dense"][0] / res["tri"][0], err), flush=True)
