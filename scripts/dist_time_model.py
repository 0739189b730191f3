"""Time model of the 2-D block-cyclic fit (gpexp_amd/dist.py), round 3: the per-rank GPU time and the host issue
time are MEASURED (single-rank replay of rank (pr, pc)'s exact kernel sequence on one MI355X, scripts/dist_replay.py ->
profiles/r03_dist_replay_fit_only.json / _fit_ivar.json); only the xGMI terms are still assumptions, because RCCL with more
than one rank has never run on the build's hardware (one GPU per lease).

    python scripts/dist_time_model.py [profiles/r03_dist_replay_fit_only.json profiles/r03_dist_replay_fit_ivar.json]

Per grid: the slowest replayed rank's GPU time per step (every receive already costs its device copy there, so staging is in),
its host issue time, the bytes that rank receives per fit, and two bracketing projections of the step on a real node:
  overlapped  max(GPU time, communication time): transfers hidden behind compute (what the stream plumbing is built for)
  exposed     GPU time + communication time: nothing hidden
with communication time = bytes received / (links used x 153 GB/s x 0.8) + 2 latency hops x 25 us x panel steps.  Links used:
the all-link panel broadcast (gpx_comm_panel_bcast) delivers over all W-1 links of the receiver; at 2 ranks there is one link."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
f_fit = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r03_dist_replay_fit_only.json")
f_all = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles", "r03_dist_replay_fit_ivar.json")
LINK, EFF, LAT = 153e9, 0.8, 25e-6
# The GLOBAL diagonal chain, which no single-rank replay contains (a rank executes only the diagonal blocks it owns; the others'
# arrive as copies): per panel step the potrf of one nb = 512 block (4 leaves + strip multiplies + rank-128 updates: 0.30 ms
# measured), the early solve of block row k+1 (0.08 ms) and two small broadcasts -- a lower bound of the factorisation at any grid.
CHAIN_STEP = 0.30e-3 + 0.08e-3 + 2 * LAT


def worst(path):
    out = {}
    for r in json.load(open(path)):
        g = r["grid"]
        if g not in out or r["ms_per_step"] > out[g]["ms_per_step"]:
            out[g] = r
    return out


fit, full = worst(f_fit), worst(f_all)
single_potrf = next(iter(fit.values()))["single_gpu_potrf_ms"]
print("inputs: %s, %s" % (os.path.basename(f_fit), os.path.basename(f_all)))
print("single-GPU potrf in the same run: %.1f ms (this box; boxes of the pool differ by up to 10 %%)\n" % single_potrf)
print("FACTORISATION alone (kfill + dist2_potrf; evaluation set of 1024 points)")
print("%-5s %6s | %12s %10s %9s %8s | %11s %11s | %18s" % ("grid", "ranks", "GPU ms/step", "host ms", "GB recv", "comm ms", "overlapped",
                                                            "exposed", "speed-up vs 1 GPU"))
for g in ("1x1", "1x2", "2x2", "2x4"):
    if g not in fit:
        continue
    r = fit[g]
    W = int(g[0]) * int(g[2])
    links = max(W - 1, 1)
    comm = (r["bytes_received_per_fit"] / (links * LINK * EFF) + 2 * LAT * r["steps_k"]) if W > 1 else 0.0
    t = r["ms_per_step"] * 1e-3
    chain = CHAIN_STEP * r["steps_k"] if W > 1 else 0.0
    lo, hi = max(t, comm, chain), max(t, chain) + comm
    print("%-5s %6d | %12.1f %10.1f %9.2f %8.1f | %8.1f ms %8.1f ms | %6.1fx .. %5.1fx   (global chain floor %.1f ms)" %
          (g, W, 1e3 * t, r["host_issue_ms_per_fit"], r["bytes_received_per_fit"] / 1e9, 1e3 * comm, 1e3 * lo, 1e3 * hi,
           single_potrf / (1e3 * hi), single_potrf / (1e3 * lo), 1e3 * chain))
print("\nFIT + IVAR over M = 32768 (the bench step without alpha / logdet; IVAR streamed from 4 ranks)")
print("%-5s %6s | %12s %9s %8s | %11s %11s" % ("grid", "ranks", "GPU ms/step", "GB recv", "comm ms", "overlapped", "exposed"))
for g in ("1x1", "1x2", "2x2", "2x4"):
    if g not in full:
        continue
    r = full[g]
    W = int(g[0]) * int(g[2])
    links = max(W - 1, 1)
    comm = (r["bytes_received_per_fit"] / (links * LINK * EFF) + 2 * LAT * r["steps_k"]) if W > 1 else 0.0
    t = r["ms_per_step"] * 1e-3
    extra = "" if r["streamed_ivar"] else "  (+ IVAR after the fit: ~487 ms / %d ranks, not in this replay)" % W
    print("%-5s %6d | %12.1f %9.2f %8.1f | %8.1f ms %8.1f ms%s" % (g, W, 1e3 * t, r["bytes_received_per_fit"] / 1e9, 1e3 * comm,
                                                                 1e3 * max(t, comm), 1e3 * (t + comm), extra))
print("\nalpha / log det behind the step: 2.8 ms of launches (streamed grids: backward sweep only, 128 small collectives) or 3.6 ms "
      "(grids with a replica: local sweeps, no exchange) -- scripts/probe_dist_solve.py")
print("NOT measured: xGMI transfer time, RCCL launch latency, waiting for peers.  Measured: everything a rank's GPU and host do.")
