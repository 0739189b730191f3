"""Node PROJECTION of the 2-D block-cyclic fit from the paced replay, round 4.  MEASURED (on one MI355X): the chain
across ranks and every rank's kernels, with each receive standing in as a device copy of the same bytes
(bench.py `multi_gpu_replay`, or a `scripts/dist_replay.py --paced-grid` line).  MODELLED (RCCL with more than one rank has never
run on the build's hardware -- one GPU per lease): what xGMI adds.  The two are kept apart in the output.

    python scripts/dist_time_model.py [profiles/r04_bench_n1.json]

Model.  The panel of step k is on the chain: panel k+1's solve needs it.  In the replay its pieces arrive as device copies on the
communication stream, in order, behind the pacing spin -- `foreign_copy_ms`, what those copies cost a rank over a factorisation, is
measured (the replay's `foreign_excess_ms`).  On a node the same bytes come over the receiver's links instead:
    transfer(k) = bytes_in(k) / (links x 153 GB/s x eff) + 2 phases x hop latency       (gpx_comm_panel_bcast: scatter + all-gather
                                                                                         over all W - 1 links of the receiver)
and the two small broadcasts of the diagonal chain (L_kk down the process column, L[k+1, k] along the process row; 2.6 and 2.1 MB
at nb = 512) cost one hop each: latency + bytes / (one link x eff).  The projection replaces the stand-in copies by the modelled
transfers and adds the small hops to every step (pessimistic: the diagonal chain is not the longer one in every step):
    projected = paced step - foreign_copy_ms + sum_k transfer(k) + steps x (hop(L_kk) + hop(L[k+1, k]))
for eff in {0.8, 0.5} and hop latency in {10, 25} us.  None of it is a measurement of xGMI."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r04_bench_n1.json")
LINK = 153e9

line = None
for ln in open(path):
    if ln.startswith("{"):
        line = json.loads(ln)
rep = line.get("multi_gpu_replay", line)          # a bench line, or a paced_grid line itself
fit = rep.get("fit_only", rep)
Pr, Pc = (int(v) for v in rep.get("grid", "2x4").split("x"))
W, nb = Pr * Pc, int(rep.get("nb", 512))
N = int(line.get("config", {}).get("N", 32768)) if "config" in line else 32768
steps = (N + nb - 1) // nb
paced = float(fit["paced_step_ms_max"])
single = float(rep.get("single_gpu_fit_ms", rep.get("single_gpu_potrf_ms", 0.0)))
excess = fit.get("foreign_excess_ms") or {}
copy_ms = fit.get("standin_copy_ms_max") or (max(excess.values()) if excess else 7.0)  # measured where the line carries it
bytes_in = float(fit.get("bytes_received_per_step", fit.get("bytes_received_per_fit", 0.0)))

print("input: %s   grid %dx%d, N = %d, nb = %d, %d panel steps" % (os.path.basename(path), Pr, Pc, N, nb, steps))
print("MEASURED  paced step of the slowest rank %.1f ms (chain %.1f ms), stand-in copies of the receives in it ~%.1f ms, %.2f GB received "
      "per rank and fit, single-GPU fit %.1f ms -> %.2fx" % (paced, float(fit["chain_ms"]), copy_ms, bytes_in / 1e9, single, single / paced))
print("MODELLED  (assumptions in the header; not a measurement)")
print("%-28s %12s %12s %12s %10s" % ("links eff / hop latency", "panels ms", "small hops ms", "projected ms", "vs 1 GPU"))
dsz = (nb * nb + nb * 128) * 8.0           # L_kk + its leaf inverses
early = nb * nb * 8.0                       # L[k+1, k]
for eff in (0.8, 0.5):
    for lat in (10e-6, 25e-6):
        # bytes a rank receives in step k: the panel's (N - (k+1) nb) x nb doubles minus its own piece (1 / W of it on average)
        panels = sum(max(N - (k + 1) * nb, 0) * nb * 8.0 * (W - 1) / W / ((W - 1) * LINK * eff) + 2 * lat for k in range(steps))
        hops = steps * ((lat + dsz / (LINK * eff)) * (1 if Pr > 1 else 0) + (lat + early / (LINK * eff)) * (1 if Pc > 1 else 0))
        proj = paced - copy_ms + 1e3 * panels + 1e3 * hops
        print("%-28s %12.1f %12.1f %12.1f %9.2fx" % ("%.1f / %2.0f us" % (eff, lat * 1e6), 1e3 * panels, 1e3 * hops, proj,
                                                   single / proj if single else float("nan")))
ivar = rep.get("ivar_slice_after_fit_ms")
if ivar:
    step1 = float(rep.get("single_gpu_step_ms", 0.0))
    print("\nfit + IVAR (evaluation after the fit against the rank's replica: no exchange but the final all-gather of M values):")
    print("MEASURED  %.1f + %.1f ms = %.1f ms -> %.2fx of the single-GPU step (%.1f ms); the projection adds the same xGMI terms as above"
          % (paced, float(ivar), float(rep["fit_then_ivar_ms"]), step1 / float(rep["fit_then_ivar_ms"]), step1))
