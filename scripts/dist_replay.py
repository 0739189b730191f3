#!/usr/bin/env python3
"""Single-rank REPLAY of the 2-D distributed fit on ONE GPU.

    python scripts/dist_replay.py [--grids 1x2,2x2,2x4] [--ranks 0,last] [--n 32768 --m 32768 --d 8 --nb 512 --agg 4]

One process plays rank (pr, pc) of a Pr x Pc grid: it runs that rank's exact kernel sequence of `dist2_potrf` + the streamed
IVAR solve (the recorded program bench.py replays on a real node), and every receive of the panel loop is a device copy of
the same bytes out of a complete factor resident on the GPU (scripts/replay_comm.py).  It measures what one GPU can
measure -- the rank's GPU time per step, the per-class kernel time, and the HOST time spent issuing the step -- and not what
it cannot: xGMI transfer time and the waiting for peers.  The variance sum of the rank's evaluation slice is checked against
the single-GPU path, so the replayed rank demonstrably computed its share of the factor.

Prints one JSON object per (grid, rank) and a table; scripts/dist_time_model.py reads the JSON
(profiles/r03_dist_replay_fit_only.json: `--no-stream --m 1024`; profiles/r03_dist_replay_fit_ivar.json: defaults).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

sys.path.insert(0, os.path.join(ROOT, "scripts"))

from gpexp_amd import device as dev, dist  # noqa: E402
from replay_comm import ReplayComm  # noqa: E402


def replay_rank(ctx, spec, Xh, yh, Zh, noise, Lref, X, grid, rank, nb=512, agg=None, streamed=True, steps=3, profile=True,
                single_potrf_ms=None):
    """One process plays `rank` of the Pr x Pc `grid` against the complete factor `Lref` resident on this GPU: records the
    rank's program, checks what it computed against the single-GPU path, times `steps` steps.  Returns the result dict."""
    Pr, Pc = grid
    world = Pr * Pc
    gs = "%dx%d" % (Pr, Pc)
    n, m = Xh.shape[0], Zh.shape[0]
    agg = dist.default_agg() if agg is None else agg
    comm = ReplayComm(ctx, world, rank, Lref)
    run = dist.DistFitIvar2D(ctx, comm, spec, Xh, yh, Zh, noise, nb=nb, grid=(Pr, Pc), agg=agg, streamed=streamed, fit_only=True)
    _, part = run.step()          # records the program, first run
    ctx.sync()
    check = 0.0
    if not run.window:
        # the replicated factor this rank assembled from its own solves and the staged pieces must BE the factor
        Zc = dev.points(ctx, Zh[:512])
        _, v1 = dev.posterior(ctx, spec, run.L, X, None, Zc, want_mean=False)
        _, v0 = dev.posterior(ctx, spec, Lref, X, None, Zc, want_mean=False)
        check = float(np.max(np.abs(v1 - v0)) / np.max(np.abs(v0)))
        assert check < 1e-10, (gs, rank, check)
    if run.B is not None:     # streamed evaluation of the rank's slice against the single-GPU path
        lo, hi = dist.eval_slice(m, rank, world)
        _, var = dev.posterior(ctx, spec, Lref, X, None, dev.points(ctx, Zh[lo:hi]), want_mean=False)
        ref = float(np.sum(var))
        assert abs(part - ref) <= 1e-10 * abs(ref), (gs, rank, part, ref)
        check = max(check, abs(part - ref) / abs(ref))
    ts, host = [], []
    for _ in range(steps):
        ctx.sync()
        t0 = time.perf_counter()
        run.step()
        ctx.sync()
        ts.append(1e3 * (time.perf_counter() - t0))
        host.append(run.host_ms.get("factor", 0.0))
    geo = run.geo
    res = dict(grid=gs, rank=rank, pr=geo.pr, pc=geo.pc, N=n, M=m, nb=nb, agg=agg, steps_k=geo.nblk,
               streamed_ivar=bool(streamed), factor_window_panels=run.window, ms_per_step=float(np.median(ts)), ms_all=ts,
               host_issue_ms_per_fit=float(np.median(host)), host_issue_us_per_panel_step=1e3 * float(np.median(host)) / geo.nblk,
               program_rows=len(run.programs["factor"]), rows_per_panel_step=len(run.programs["factor"]) / geo.nblk,
               issue_mode="hipGraph (one launch per step)" if run.use_graph else "rows (one HIP call per row)",
               graph_nodes=run.programs["factor"].graph_nodes, bytes_received_per_fit=comm.bytes_in,
               variance_check_rel=check, single_gpu_potrf_ms=single_potrf_ms)
    if profile:
        # per-class kernel time of one more, instrumented, step (row by row: profiler events cannot live inside a graph)
        run.force_interpret = True
        ctx.profile(True)
        ctx.profile_reset()
        run.step()
        ctx.sync()
        prof = ctx.profile_get()
        ctx.profile(False)
        res["host_issue_ms_row_by_row"] = run.host_ms.get("factor", 0.0)
        run.force_interpret = False
        res["class_ms"] = {k: round(v["ms"], 3) for k, v in prof.items() if v["launches"]}
        res["class_launches"] = {k: v["launches"] for k, v in prof.items() if v["launches"]}
        res["gemm_flops"] = prof["gemm"]["flops"]
    del run, comm
    ctx.trim()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grids", default="1x1,1x2,2x2,2x4")
    ap.add_argument("--ranks", default="0,last")
    ap.add_argument("--n", type=int, default=32768)
    ap.add_argument("--m", type=int, default=32768)
    ap.add_argument("--d", type=int, default=8)
    ap.add_argument("--nb", type=int, default=512)
    ap.add_argument("--agg", type=int, default=dist.default_agg())
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--no-stream", action="store_true", help="factorisation alone (no evaluation streamed underneath, as the C5 fit)")
    ap.add_argument("--out", default="")
    args = ap.parse_args()

    ctx = dev.Context(0)
    dev._ctx = ctx
    rng = np.random.default_rng(args.n)
    noise = 0.1
    Xh = rng.uniform(-1, 1, (args.n, args.d))
    yh = np.sin(2 * np.pi * Xh.sum(1) / args.d) + np.sqrt(noise) * rng.standard_normal(args.n)
    Zh = rng.uniform(-1, 1, (args.m, args.d))
    spec = dev.KernelSpec(dev.K_MATERN52, args.d, [0.5, 1.0])

    # the complete factor every "receive" is copied from, and the single-GPU reference of the check
    X = dev.points(ctx, Xh)
    Lref = dev.kfill(ctx, spec, X, nugget=noise)
    ctx.sync()
    t0 = time.perf_counter()
    dev.potrf(ctx, Lref)
    ctx.sync()
    single_potrf_ms = 1e3 * (time.perf_counter() - t0)

    results = []
    for gs in args.grids.split(","):
        Pr, Pc = (int(v) for v in gs.split("x"))
        world = Pr * Pc
        ranks = sorted({(world - 1 if r == "last" else int(r)) for r in args.ranks.split(",") if r == "last" or int(r) < world})
        for rank in ranks:
            res = replay_rank(ctx, spec, Xh, yh, Zh, noise, Lref, X, (Pr, Pc), rank, nb=args.nb, agg=args.agg,
                              streamed=(world >= 4 and not args.no_stream), steps=args.steps, single_potrf_ms=single_potrf_ms)
            results.append(res)
            print(json.dumps(res), flush=True)
    print("\n%-6s %-5s %10s %14s %16s %10s %12s" % ("grid", "rank", "ms/step", "host ms/fit", "host us/k-step", "rows/k", "GB received"))
    for r in results:
        print("%-6s %-5d %10.2f %14.2f %16.1f %10.1f %12.2f" % (r["grid"], r["rank"], r["ms_per_step"], r["host_issue_ms_per_fit"],
                                                               r["host_issue_us_per_panel_step"], r["rows_per_panel_step"],
                                                               r["bytes_received_per_fit"] / 1e9))
    if args.out:
        with open(args.out, "w") as f:
            json.dump(results, f, indent=1)
    ctx.close()


if __name__ == "__main__":
    main()
