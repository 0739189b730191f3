#!/usr/bin/env python3
"""Per-step TIMELINE of the replayed 2-D distributed factorisation without a profiler in the way: the pipeline's own events
(gpx_event_record ids of gpexp_amd.dist: E_DFACT, E_PIECE, E_ARRIVED, E_COLREADY, E_UPD, E_BULK ...) carry time stamps
(GPX_EVENT_TIMING=1) and are read back after a step with gpx_dbg_event_elapsed.  rocprofv3's kernel trace makes the step
host-bound (20 us per launch), so its timeline is not the un-instrumented one.

    GPX_EVENT_TIMING=1 python scripts/dist_timeline.py [--grid 2x4 --rank 0 --n 32768 --nb 512 --stream]
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

os.environ.setdefault("GPX_EVENT_TIMING", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from gpexp_amd import device as dev, dist  # noqa: E402
from replay_comm import ReplayComm  # noqa: E402

KINDS = ["COLREADY", "DFACT", "DBC", "PIECE", "ARRIVED", "STORED", "UPD", "DIAGREADY", "EARLYSOLVED", "EARLY", "COL2", "PANELDONE",
         "BULK", "IVAR"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", default="2x4")
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--n", type=int, default=32768)
    ap.add_argument("--m", type=int, default=1024)
    ap.add_argument("--d", type=int, default=8)
    ap.add_argument("--nb", type=int, default=512)
    ap.add_argument("--stream", action="store_true")
    args = ap.parse_args()
    ctx = dev.Context(0)
    dev._ctx = ctx
    rng = np.random.default_rng(args.n)
    Xh = rng.uniform(-1, 1, (args.n, args.d))
    yh = np.sin(2 * np.pi * Xh.sum(1) / args.d) + np.sqrt(0.1) * rng.standard_normal(args.n)
    Zh = rng.uniform(-1, 1, (args.m, args.d))
    spec = dev.KernelSpec(dev.K_MATERN52, args.d, [0.5, 1.0])
    X = dev.points(ctx, Xh)
    Lref = dev.potrf(ctx, dev.kfill(ctx, spec, X, nugget=0.1))
    Pr, Pc = (int(v) for v in args.grid.split("x"))
    comm = ReplayComm(ctx, Pr * Pc, args.rank, Lref)
    run = dist.DistFitIvar2D(ctx, comm, spec, Xh, yh, Zh, 0.1, nb=args.nb, grid=(Pr, Pc), streamed=args.stream, fit_only=True)
    for _ in range(3):
        run.step()
        ctx.sync()
    geo = run.geo

    def at(kind, k):
        ms = C.c_double()
        rc = ctx.lib.gpx_dbg_event_elapsed(ctx.h, dist.EV_FORK, dist._ev2(kind, k), C.byref(ms))
        return ms.value if rc == 0 else None

    print("# rank %d of %s, N=%d nb=%d: ms after the step's fork at which each event of step k completed" % (args.rank, args.grid, args.n, args.nb))
    print("k    " + " ".join("%9s" % s for s in ("DFACT", "PIECE", "ARRIVED", "COLREADY", "COL2", "UPD", "BULK", "STORED")))
    rows = []
    for k in range(geo.nblk):
        vals = [at(getattr(dist, "E_" + s), k) for s in ("DFACT", "PIECE", "ARRIVED", "COLREADY", "COL2", "UPD", "BULK", "STORED")]
        rows.append(vals)
        print("%-4d " % k + " ".join("%9.3f" % v if v is not None else "        -" for v in vals))
    arr = [r[2] for r in rows if r[2] is not None]
    d = np.diff(arr)
    print("# ARRIVED(k+1) - ARRIVED(k): mean %.3f ms, first half %.3f, second half %.3f; last ARRIVED at %.2f ms"
          % (d.mean(), d[:len(d) // 2].mean(), d[len(d) // 2:].mean(), arr[-1]))
    upd = [r[5] for r in rows if r[5] is not None]
    print("# last UPD at %.2f ms" % upd[-1])
    ctx.close()


if __name__ == "__main__":
    main()
