"""One launch of the segmented trailing update (gpx_dist2_update_multi, 1 x 1 grid) against the plain lower SYRK launch of
the same shape: what does the segmented / trapezoid-skipping kernel cost by itself?  usage: probe_update_multi.py [N nb nseg J0]"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpexp_amd import device as dev, dist  # noqa: E402
from gpexp_amd._lib import check, c_i64, c_vp  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 512
nseg = int(sys.argv[3]) if len(sys.argv) > 3 else 4
J0 = int(sys.argv[4]) if len(sys.argv) > 4 else 8
ctx = dev.context()
geo = dist.Grid2D(N, nb, 1, 1, 0)
A = dev.DeviceMatrix.zeros(ctx, N, N)
rng = np.random.default_rng(0)
G = []
for s in range(nseg):
    g = dev.DeviceMatrix.zeros(ctx, geo.buf_elems(), 1, pad=False)
    blk = rng.standard_normal(1 << 20)
    for off in range(0, geo.buf_elems(), blk.size):
        cnt = min(blk.size, geo.buf_elems() - off)
        check(ctx.lib.gpx_mat_write(ctx.h, g.h, off, cnt, blk.ctypes.data_as(C.POINTER(C.c_double))))
    G.append(g)
m = N - J0 * nb
hs = (c_vp * nseg)(*[g.h for g in G])
ks = (c_i64 * nseg)(*range(nseg))
tiles = (m // 128) * (m // 128 + 1) / 2
for below in (0,):
    ts = []
    for it in range(4):
        ctx.sync()
        t0 = time.perf_counter()
        check(ctx.lib.gpx_dist2_update_multi(ctx.h, A.h, J0 * nb, m, J0 * nb, m, nb, 1, 1, 0, 0, geo.piece_stride, nseg, hs, ks, below))
        ctx.sync()
        ts.append(time.perf_counter() - t0)
    t = min(ts[1:])
    # the kernel computes whole nb-blocks on the diagonal: count the tiles it really does
    nblk = m // nb
    tiles_done = nblk * (nblk + 1) / 2 * (nb // 128) ** 2
    print("update_multi m=%d K=%d x %d: %.3f ms  %.1f TF/s (tiles computed) %.1f TF/s (lower-triangle tiles)" %
          (m, nb, nseg, 1e3 * t, tiles_done * 2 * 128 * 128 * nb * nseg / t / 1e12, tiles * 2 * 128 * 128 * nb * nseg / t / 1e12))
# plain lower SYRK of the same shape: C (m x m) -= P P^T, P m x K
K = nb * nseg
P = dev.DeviceMatrix.from_host(ctx, rng.standard_normal((m, K)), pad=True)   # random like the packed buffers: MFMA power
Cm = dev.DeviceMatrix.zeros(ctx, m, m)                                          # (and with it the clock) depends on the data
ts = []
for it in range(4):
    ctx.sync()
    t0 = time.perf_counter()
    dev.dbg_gemm(ctx, P, P, Cm, 1, 1, lower=True)
    ctx.sync()
    ts.append(time.perf_counter() - t0)
t = min(ts[1:])
print("plain lower SYRK m=%d K=%d: %.3f ms  %.1f TF/s" % (m, K, 1e3 * t, tiles * 2 * 128 * 128 * K / t / 1e12))
# E4: a fully active rectangle (all tiles below the diagonal): rows [N/2, N) x cols [J0 nb, J0 nb + N/2 - J0 nb)
mr = N // 2
nc = N // 2 - J0 * nb
ts = []
for it in range(4):
    ctx.sync()
    t0 = time.perf_counter()
    check(ctx.lib.gpx_dist2_update_multi(ctx.h, A.h, N // 2, mr, J0 * nb, nc, nb, 1, 1, 0, 0, geo.piece_stride, nseg, hs, ks, 0))
    ctx.sync()
    ts.append(time.perf_counter() - t0)
t = min(ts[1:])
print("update_multi RECT (all tiles active) m=%d n=%d K=%d x %d: %.3f ms  %.1f TF/s" % (mr, nc, nb, nseg, 1e3 * t, 2.0 * mr * nc * nb * nseg / t / 1e12))
P2 = dev.DeviceMatrix.from_host(ctx, rng.standard_normal((mr, K)), pad=True)
Q2 = dev.DeviceMatrix.from_host(ctx, rng.standard_normal((nc, K)), pad=True)
C2 = dev.DeviceMatrix.zeros(ctx, mr, nc)
ts = []
for it in range(4):
    ctx.sync()
    t0 = time.perf_counter()
    dev.dbg_gemm(ctx, P2, Q2, C2, 1, 1, lower=False)
    ctx.sync()
    ts.append(time.perf_counter() - t0)
t = min(ts[1:])
print("plain RECT m=%d n=%d K=%d: %.3f ms  %.1f TF/s" % (mr, nc, K, 1e3 * t, 2.0 * mr * nc * K / t / 1e12))
