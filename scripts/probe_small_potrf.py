"""Panel-sized factorisations and panel solves: the per-step chain costs of the multi-GPU time model."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gpexp_amd import device as dev
ctx = dev.context()
rng = np.random.default_rng(0)
sp = dev.KernelSpec(2, 8, [0.5, 1.0])
for nb in (256, 512, 1024, 2048):
    X = dev.points(ctx, rng.uniform(-1, 1, (nb, 8)))
    K = dev.DeviceMatrix.zeros(ctx, nb, nb)
    ts = []
    for it in range(12):
        dev.kfill_into(ctx, sp, X, K, nugget=0.1); ctx.sync()
        t0 = time.perf_counter(); dev.potrf(ctx, K); ts.append(time.perf_counter() - t0)
    print("potrf(%d): %.3f ms (best of 12, includes one host sync and, from 2048, the block-inverse build)" % (nb, 1e3 * min(ts[2:])), flush=True)
    if nb <= 1024 and os.environ.get("GPX_POTRF_COOP", "1") != "0":
        import ctypes as C
        st = (C.c_int64 * 24)()
        if ctx.lib.gpx_dbg_coop_stamps(ctx.h, st) == 0:
            tw, tl, td = list(st[0:8]), list(st[8:16]), list(st[16:24])
            nl = nb // 128
            print("   chain (us): " + " | ".join("p%d wait %.1f leaf+flag %.1f" % (p, (tl[p] - tw[p]) / 100.0, (td[p] - tl[p]) / 100.0)
                                                  for p in range(nl)) + " | total %.1f" % ((td[nl - 1] - tw[0]) / 100.0), flush=True)
# trailing-update GEMM rates at K = nb (m x m x nb lower): what a rank's share of the update runs at
for nb in (256, 512, 1024):
    m = 16384
    A = dev.DeviceMatrix.from_host(ctx, rng.standard_normal((m, nb)), pad=True)
    Cm = dev.DeviceMatrix.zeros(ctx, m, m)
    ts = []
    for it in range(5):
        ctx.sync(); t0 = time.perf_counter(); dev.dbg_gemm(ctx, A, A, Cm, 1, 1, lower=True); ctx.sync(); ts.append(time.perf_counter() - t0)
    t = min(ts[1:])
    print("update %d x %d (lower) K=%d: %.3f ms = %.1f TF/s" % (m, m, nb, 1e3 * t, m * (m + 128) * nb / t / 1e12), flush=True)
