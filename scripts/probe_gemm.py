"""Isolated GEMM timing: python probe_gemm.py m,n,k,bt,acc,lower [...]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gpexp_amd import device as dev
ctx = dev.context()
for a in sys.argv[1:]:
    m, n, k, bt, acc, low = map(int, a.split(","))
    A = dev.DeviceMatrix.zeros(ctx, m, k)
    B = dev.DeviceMatrix.zeros(ctx, n, k) if bt else dev.DeviceMatrix.zeros(ctx, k, n)
    C = dev.DeviceMatrix.zeros(ctx, m, n)
    rng = np.random.default_rng(1)
    # random data (zeros run at a higher clock): fill via a small host block tiled by kfill is overkill; upload rows
    blk = rng.standard_normal((min(m, 1024), k))
    A2 = dev.DeviceMatrix.from_host(ctx, np.tile(blk, (m // blk.shape[0], 1)))
    blkb = rng.standard_normal((min(n, 1024), k)) if bt else rng.standard_normal((k, min(n, 1024)))
    B2 = dev.DeviceMatrix.from_host(ctx, np.tile(blkb, (n // blkb.shape[0], 1)) if bt else np.tile(blkb, (1, n // blkb.shape[1])))
    for rep in range(3):
        ctx.profile(True); ctx.profile_reset()
        dev.dbg_gemm(ctx, A2, B2, C, bt, acc, low)
        p = ctx.profile_get()["gemm"]; ctx.profile(False)
        print("gemm m=%d n=%d k=%d bt=%d acc=%d lower=%d: %.3f ms  %.2f TF/s" % (m, n, k, bt, acc, low, p["ms"], p["flops"] / p["ms"] / 1e9), flush=True)
