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


def own_step_latencies(ctx, geo):
    """ms between the arrival of panel k-1 and of panel k for the steps k whose panel THIS rank's process column solves, from the
    time stamps of the pipeline's own events (GPX_EVENT_TIMING=1): what the rank adds to the chain across ranks when it is the
    holder -- its near update of column k, the panel solve, the hand-over to the communication stream."""
    import ctypes as C

    def at(kind, k):
        ms = C.c_double()
        rc = ctx.lib.gpx_dbg_event_elapsed(ctx.h, dist.EV_FORK, dist._ev2(kind, k), C.byref(ms))
        return ms.value if rc == 0 else None
    arr = [at(dist.E_ARRIVED, k) for k in range(geo.nblk)]
    if any(a is None for a in arr):
        return None, None
    own = [k for k in range(geo.nblk) if k % geo.Pc == geo.pc]
    lat = {k: arr[k] - (arr[k - 1] if k > 0 else 0.0) for k in own}
    # round 5: the panel travels in two row chunks; the first one's arrivals form a chain of their own (chunk 0 of panel k needs
    # chunk 0 of panel k-1 only)
    arr0 = [at(dist.E_ARR0, k) for k in range(geo.nblk)]
    own_step_latencies.lat0 = (None if any(a is None for a in arr0) else
                               {k: arr0[k] - (arr0[k - 1] if k > 0 else 0.0) for k in own})
    # the diagonal chain's two hand-overs, where THIS rank produces them: the diagonal block factored (owner of (k, k)) and block
    # row k+1 of panel k solved (holder of column k in the process row of block row k+1), ms after the arrival of panel k-1
    own_step_latencies.dfact = {k: at(dist.E_DFACT, k) - (arr[k - 1] if k > 0 else 0.0) for k in own if k % geo.Pr == geo.pr}
    own_step_latencies.early = {k: at(dist.E_EARLYSOLVED, k) - (arr[k - 1] if k > 0 else 0.0)
                                for k in own if k + 1 < geo.nblk and (k + 1) % geo.Pr == geo.pr}
    if os.environ.get("GPX_REPLAY_DETAIL") == "1":   # how far the strands that release a panel buffer run behind its arrival
        for kind, nm in ((dist.E_STORED, "copied into the replica (BACK)"), (dist.E_UPD, "near updates (MAIN)"),
                         (dist.E_BULK, "bulk update of the group (EVAL / BULK)"), (dist.E_PANELDONE, "panel stream")):
            lags = [(at(kind, k) - arr[k], k) for k in range(geo.nblk) if at(kind, k) is not None and at(kind, k) > 0]
            if lags:
                print("# lag behind the panel's arrival, %-40s mean %6.3f ms, max %6.3f ms at k = %d, in steps of the chain: max %.1f"
                      % (nm + ":", sum(v for v, _ in lags) / len(lags), max(lags)[0], max(lags)[1],
                         max(v / max(arr[min(k + 1, geo.nblk - 1)] - arr[k], 1e-3) for v, k in lags if k + 1 < geo.nblk)), file=sys.stderr)
    if os.environ.get("GPX_REPLAY_DETAIL") == "1":   # where an own holder step spends its latency
        print("# own holder steps: k, [ARRIVED(k-1)] -> near update of column k done (COLREADY) -> diagonal factored (DFACT) -> "
              "panel solved (PIECE) -> delivered (ARRIVED); ms relative to ARRIVED(k-1)", file=sys.stderr)
        for k in own[1:]:
            base = arr[k - 1]
            v = [at(kind, k) for kind in (dist.E_COL2, dist.E_COLREADY, dist.E_DIAGREADY, dist.E_DFACT, dist.E_EARLYSOLVED, dist.E_PIECE)]
            print("#  k=%2d  COL2 %7.3f  COLREADY %7.3f  DIAGREADY %7.3f  DFACT %7.3f  EARLYSOLVED %7.3f  PIECE %7.3f  ARRIVED %7.3f" %
                  tuple([k] + [(x - base) if x is not None else float("nan") for x in v] + [arr[k] - base]), file=sys.stderr)
    return lat, arr


def pace_from(lat, nblk):
    """Per-step pacing (us) for the foreign steps: the own-step latencies interpolated over k (they fall as the trailing matrix
    shrinks)."""
    ks = sorted(lat)
    xs = np.arange(nblk)
    return np.maximum(np.interp(xs, ks, [lat[k] for k in ks]) * 1e3, 0.0).astype(np.int64)


def paced_replay(ctx, spec, Xh, yh, Zh, noise, Lref, X, grid, rank, nb=None, agg=None, streamed=False, iters=3, steps=2):
    """The factorisation time of the WHOLE grid estimated on one GPU: replay `rank` with every foreign panel held back by the
    latency this rank shows in its own holder steps, and iterate (the latencies depend on how busy the rank is, which depends
    on the pacing).  Needs GPX_EVENT_TIMING=1 in the environment before the context's first event.  Returns a dict."""
    hist = []
    pace = None
    res = None
    for it in range(iters + 1):
        res = replay_rank(ctx, spec, Xh, yh, Zh, noise, Lref, X, grid, rank, nb=nb, agg=agg, streamed=streamed, steps=steps,
                          profile=False, pace_us=pace, want_latencies=True)
        lat = res.pop("own_latency_ms")
        lat0 = res.pop("own_latency0_ms", None)
        res.pop("own_dfact_ms", None)
        res.pop("own_early_ms", None)
        if lat is None:
            return dict(error="the pipeline's events carry no time stamps (set GPX_EVENT_TIMING=1 before the first use)")
        hist.append(dict(paced=pace is not None, ms_per_step=res["ms_per_step"], own_latency_sum_ms=float(sum(lat.values())),
                         chain_estimate_ms=float(sum(lat.values())) * grid[1]))
        pace = dict(panel=pace_from(lat, res["steps_k"]), panel0=pace_from(lat0 or lat, res["steps_k"]), dfact=None, early=None)
    return dict(grid=res["grid"], rank=rank, nb=res["nb"], streamed_ivar=bool(streamed), iterations=hist,
                unpaced_rank_busy_ms=hist[0]["ms_per_step"], paced_step_ms=hist[-1]["ms_per_step"],
                own_latency_ms_first_last=[round(lat[min(lat)], 3), round(lat[max(lat)], 3)],
                bytes_received_per_fit=res["bytes_received_per_fit"], variance_check_rel=res["variance_check_rel"])


def paced_grid(ctx, spec, Xh, yh, Zh, noise, Lref, X, grid, nb=None, agg=None, streamed=False, iters=5, steps=2, rows=None):
    """The factorisation time of the WHOLE Pr x Pc grid estimated on ONE GPU.  Step k+1's panel solve needs step k's panel, so
    the factorisation time of a real run is the SUM over the steps of what the step's holder column needs from the arrival of
    panel k-1 to the delivery of panel k -- not any rank's busy time (a replay in which every foreign panel arrives at once keeps
    the rank busy all the time and hides exactly that chain).  Here every process column is replayed in turn (its ranks in the
    process rows `rows`, default all), each replay PACED: a foreign panel k arrives at the latency measured for ITS holder
    column (the slowest of that column's replayed ranks); the latencies depend on how busy a rank is, which depends on the
    pacing, so the sweep over the columns is iterated.  Result: the per-step holder latencies, their sum (the chain) and the
    paced step time of every replayed rank (they share one global timeline, so they agree up to the ranks' last updates).
    xGMI transfer time is NOT in it (receives are device copies).  Needs GPX_EVENT_TIMING=1 before the context's first event."""
    Pr, Pc = grid
    rows = list(range(Pr)) if rows is None else list(rows)
    nblk = None
    lat = {}           # step k -> holder latency (ms), max over the replayed ranks of the holder column
    lat0 = {}          # the same for the FIRST row chunk of the panel (round 5)
    dfl, eal = {}, {}  # step k -> ms after the arrival of panel k-1 at which the diagonal block was factored / block row k+1 solved
    hist = []
    last = {}
    diag_paced = os.environ.get("GPX_REPLAY_PACE_DIAG", "1") == "1" and len(rows) == Pr
    for it in range(iters):
        newlat, newdf, newea, newlat0 = {}, {}, {}, {}
        for pc in range(Pc):
            for pr in rows:
                rank = pr * Pc + pc
                pace = None
                if nblk is not None and len(lat) == nblk:
                    us = lambda d: np.array([max(d.get(k, 0.0), 0.0) * 1e3 for k in range(nblk)]).astype(np.int64)  # noqa: E731
                    # (sweep 1 paces the panels alone: the figures of the unpaced sweep 0 are those of a rank that is never idle,
                    # several times the fixed point's, and holding the diagonal chain back by them too keeps them there)
                    dp = diag_paced and it >= 2
                    pace = dict(panel=us(lat), panel0=us(lat0 if len(lat0) == nblk else lat), dfact=us(dfl) if dp else None,
                                early=us(eal) if dp else None)
                res = replay_rank(ctx, spec, Xh, yh, Zh, noise, Lref, X, grid, rank, nb=nb, agg=agg, streamed=streamed, steps=steps,
                                  profile=False, pace_us=pace, want_latencies=True)
                own = res.pop("own_latency_ms")
                if own is None:
                    return dict(error="the pipeline's events carry no time stamps (set GPX_EVENT_TIMING=1 before the first use)")
                nblk = res["steps_k"]
                agg_used, nb = res["agg"], res["nb"]
                for k, v in own.items():
                    newlat[k] = max(newlat.get(k, 0.0), v)
                for k, v in (res.pop("own_latency0_ms", None) or own).items():
                    newlat0[k] = max(newlat0.get(k, 0.0), v)
                newdf.update(res.pop("own_dfact_ms") or {})
                newea.update(res.pop("own_early_ms") or {})
                last[rank] = dict(paced=pace is not None, ms_per_step=res["ms_per_step"], own_latency_sum_ms=float(sum(own.values())),
                                  last_arrived_ms=res.get("last_arrived_ms"), last_step_ms=res.get("last_step_ms"),
                                  foreign_excess_ms=res.get("foreign_excess_ms"), foreign_excess_worst=res.get("foreign_excess_worst"),
                                  bytes_received_per_fit=res["bytes_received_per_fit"], check=res["variance_check_rel"])
        # damped fixed point: a rank's holder latency falls when its foreign panels arrive later (it is less busy), which makes
        # the next sweep's pacing shorter and the latencies rise again -- the plain iteration oscillates (fit + IVAR at 2 x 4:
        # 215, 72, 110, 93, 102 ms of chain); from the second paced sweep on the pacing moves half-way
        damp = lambda old, new: new if it < (3 if diag_paced else 2) else {k: 0.5 * (old.get(k, new[k]) + new[k]) for k in new}  # noqa: E731
        lat, dfl, eal, lat0 = damp(lat, newlat), damp(dfl, newdf), damp(eal, newea), damp(lat0, newlat0)
        hist.append(dict(iteration=it, chain_ms=float(sum(lat.values())),
                         rank_step_ms={str(r): round(v["ms_per_step"], 3) for r, v in last.items()},
                         paced=all(v["paced"] for v in last.values())))
    per_col = [float(sum(v for k, v in lat.items() if k % Pc == pc)) for pc in range(Pc)]
    return dict(grid="%dx%d" % grid, nb=nb, agg=agg_used, streamed_ivar=bool(streamed), diagonal_chain_paced=bool(diag_paced),
                replayed_ranks=sorted(last), iterations=hist, chain_ms=float(sum(lat.values())), chain_ms_by_process_column=per_col,
                paced_step_ms_max=max(v["ms_per_step"] for v in last.values()),
                paced_step_ms={str(r): round(v["ms_per_step"], 3) for r, v in last.items()},
                last_panel_arrived_ms={str(r): (round(v["last_arrived_ms"], 3) if v["last_arrived_ms"] is not None else None)
                                       for r, v in last.items()},
                foreign_excess_ms={str(r): v.get("foreign_excess_ms") for r, v in last.items()},
                foreign_excess_worst={str(r): v.get("foreign_excess_worst") for r, v in last.items()},
                holder_latency_ms_first_mid_last=[round(lat[1], 3), round(lat[nblk // 2], 3), round(lat[nblk - 1], 3)],
                variance_check_rel=max(v["check"] for v in last.values()),
                bytes_received_per_fit=max(v["bytes_received_per_fit"] for v in last.values()))


def replay_rank(ctx, spec, Xh, yh, Zh, noise, Lref, X, grid, rank, nb=None, agg=None, streamed=True, steps=3, profile=True,
                single_potrf_ms=None, pace_us=None, want_latencies=False):
    """One process plays `rank` of the Pr x Pc `grid` against the complete factor `Lref` resident on this GPU: records the
    rank's program, checks what it computed against the single-GPU path, times `steps` steps.  Returns the result dict."""
    Pr, Pc = grid
    world = Pr * Pc
    gs = "%dx%d" % (Pr, Pc)
    n, m = Xh.shape[0], Zh.shape[0]
    comm = ReplayComm(ctx, world, rank, Lref, pace_us=pace_us)
    cyclic = os.environ.get("GPX_REPLAY_CYCLIC") == "1"     # the class API's distributed-factor mode: no replica, re-streamed evaluation
    run = dist.DistFitIvar2D(ctx, comm, spec, Xh, yh, Zh[:0] if cyclic else Zh, noise, nb=nb, grid=(Pr, Pc), agg=agg,
                             streamed=False if cyclic else streamed, fit_only=True, cyclic=cyclic)
    if os.environ.get("GPX_REPLAY_NO_REPLICA") == "1":
        run.cyclic_only = True                              # factorisation without any copy of the finished panels
    _, part = run.step()          # records the program, first run
    ctx.sync()
    check = 0.0
    if not run.window and not run.cyclic_only:
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
    lat, arr_all = own_step_latencies(ctx, geo) if want_latencies else (None, None)
    cyc = None
    if cyclic and pace_us is None:
        # the evaluation of this rank's slice against the block-cyclic factor: panels re-streamed (stand-in copies for the
        # receives), windowed solve -- checked against the single-GPU path, timed; and what the rank holds for the factor
        lo, hi = dist.eval_slice(m, rank, world)
        _, var = run.cyclic_posterior(Zh[lo:hi], None, False, True)
        _, ref = dev.posterior(ctx, spec, Lref, X, None, dev.points(ctx, Zh[lo:hi]), want_mean=False)
        err = float(np.max(np.abs(var - ref)) / np.max(np.abs(ref)))
        assert err < 1e-10, (gs, rank, err)
        tt = []
        for _ in range(2):
            ctx.sync()
            t0 = time.perf_counter()
            run.cyclic_posterior(Zh[lo:hi], None, False, True)
            ctx.sync()
            tt.append(1e3 * (time.perf_counter() - t0))
        A8 = 8 * geo.local_rows(geo.pr) * geo.local_cols(geo.pc)
        cyc = dict(restreamed_ivar_slice_ms=min(tt), slice_points=hi - lo, variance_rel_err=err,
                   resident_factor_bytes=run.resident_bytes(), local_matrix_bytes=A8,
                   ring_bytes=8 * len(run.G) * geo.buf_elems(), window_bytes=8 * geo.np * 2 * run.agg * run.nb,
                   replica_would_be_bytes=8 * geo.np * geo.np, cross_slice_bytes=8 * geo.np * (hi - lo))
    nb = run.nb
    res = dict(grid=gs, rank=rank, pr=geo.pr, pc=geo.pc, N=n, M=m, nb=nb, agg=run.agg, steps_k=geo.nblk,
               streamed_ivar=bool(streamed), factor_window_panels=run.window, ms_per_step=float(np.median(ts)), ms_all=ts,
               host_issue_ms_per_fit=float(np.median(host)), host_issue_us_per_panel_step=1e3 * float(np.median(host)) / geo.nblk,
               program_rows=len(run.programs["factor"]), rows_per_panel_step=len(run.programs["factor"]) / geo.nblk,
               issue_mode="hipGraph (one launch per step)" if run.use_graph else "rows (one HIP call per row)",
               graph_nodes=run.programs["factor"].graph_nodes, bytes_received_per_fit=comm.bytes_in,
               variance_check_rel=check, single_gpu_potrf_ms=single_potrf_ms)
    if cyc is not None:
        res["distributed_factor"] = cyc
    if want_latencies:
        res["own_latency_ms"] = lat
        res["own_latency0_ms"] = getattr(own_step_latencies, "lat0", None) if lat is not None else None
        res["own_dfact_ms"] = getattr(own_step_latencies, "dfact", None) if lat is not None else None
        res["own_early_ms"] = getattr(own_step_latencies, "early", None) if lat is not None else None
        res["last_arrived_ms"] = arr_all[-1] if arr_all else None      # (of the LAST timed step) the rest of the step is the rank's tail
        if arr_all and pace_us is not None:      # what a FOREIGN step costs this rank beyond its pacing (the stand-in copies, waits)
            if isinstance(pace_us, dict):
                pace_us = pace_us["panel"]
            gaps = [arr_all[k] - (arr_all[k - 1] if k else 0.0) for k in range(len(arr_all))]
            fk = [k for k in range(len(arr_all)) if k % geo.Pc != geo.pc]
            res["foreign_excess_ms"] = float(sum(gaps[k] - pace_us[k] * 1e-3 for k in fk))
            res["foreign_excess_worst"] = sorted(((round(gaps[k] - pace_us[k] * 1e-3, 3), k) for k in fk), reverse=True)[:6]
        res["last_step_ms"] = ts[-1]
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
    ap.add_argument("--nb", type=int, default=0, help="block size (default: what the product's runner chooses, dist.default_nb)")
    ap.add_argument("--agg", type=int, default=None, help="panels per trailing update (default: 4 with the streamed evaluation, 2 without)")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--no-stream", action="store_true", help="factorisation alone (no evaluation streamed underneath, as the C5 fit)")
    ap.add_argument("--out", default="")
    ap.add_argument("--paced-grid", action="store_true", help="whole-grid estimate: every process column replayed, paced by the "
                                                             "holder latencies of the step's own column, iterated (paced_grid)")
    ap.add_argument("--iters", type=int, default=5, help="sweeps of --paced-grid (the first one is unpaced)")
    ap.add_argument("--rows", default="", help="process rows replayed by --paced-grid (default: all)")
    ap.add_argument("--paced", action="store_true", help="paced replay: foreign panels arrive at the latency this rank shows in its "
                                                        "own holder steps (estimate of the whole grid's factorisation time)")
    args = ap.parse_args()
    if args.paced or args.paced_grid:
        os.environ.setdefault("GPX_EVENT_TIMING", "1")

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
        if args.paced_grid:
            res = paced_grid(ctx, spec, Xh, yh, Zh, noise, Lref, X, (Pr, Pc), nb=(args.nb or None), agg=args.agg,
                             streamed=(world >= 4 and not args.no_stream),
                             rows=[int(v) for v in args.rows.split(",")] if args.rows else None, iters=args.iters)
            res["single_gpu_potrf_ms"] = single_potrf_ms
            print(json.dumps(res), flush=True)
            continue
        ranks = sorted({(world - 1 if r == "last" else int(r)) for r in args.ranks.split(",") if r == "last" or int(r) < world})
        for rank in ranks:
            if args.paced:
                res = paced_replay(ctx, spec, Xh, yh, Zh, noise, Lref, X, (Pr, Pc), rank, nb=(args.nb or None), agg=args.agg,
                                   streamed=(world >= 4 and not args.no_stream))
                res["single_gpu_potrf_ms"] = single_potrf_ms
                print(json.dumps(res), flush=True)
                continue
            res = replay_rank(ctx, spec, Xh, yh, Zh, noise, Lref, X, (Pr, Pc), rank, nb=(args.nb or None), agg=args.agg,
                              streamed=(world >= 4 and not args.no_stream), steps=args.steps, single_potrf_ms=single_potrf_ms)
            results.append(res)
            print(json.dumps(res), flush=True)
    if args.paced or args.paced_grid:
        ctx.close()
        return
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
