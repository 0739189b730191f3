"""Turn gpurun_out/final/ (scripts/collect_profiles.sh) into the judged artifacts under profiles/.
usage: summarise_profiles.py [gpurun_out/final] [r02]"""
import csv, glob, json, os, shutil, sys, collections
src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/final"
tag = sys.argv[2] if len(sys.argv) > 2 else "r01"
out = "profiles"


def cls(name):
    if "gemm_f64" in name or "leaf_mul" in name: return "gemm (gemm_f64_kernel + leaf_mul kernels)"
    for k in ("kfill", "leaf_kernel", "colreduce", "trsv", "gemv", "vec_sub", "logdet", "sum_kernel", "transpose"):
        if k in name: return k
    return name.split("(")[0][-40:]


def pmc(path, counter):
    agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
    f = glob.glob(os.path.join(src, path, "*counter_collection.csv"))[0]
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter: continue
        a = agg[cls(r["Kernel_Name"])]
        a[0] += 1; a[1] += float(r["Counter_Value"]); a[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    return agg


shutil.copy(os.path.join(src, "bench.json"), os.path.join(out, tag + "_bench_n1.json"))
for nm, dst in (("run_kernel_stats.csv", "_kernel_stats.csv"), ("run_domain_stats.csv", "_domain_stats.csv")):
    f = glob.glob(os.path.join(src, "stats", "*" + nm)) or glob.glob(os.path.join(src, "stats", "*", "*" + nm))
    if f: shutil.copy(f[0], os.path.join(out, tag + dst))
lines = ["# " + tag + " -- HBM-side traffic of one bench step (C4, N=32768, d=8, M=32768), rocprofv3 --pmc, separate passes",
         "# FETCH_SIZE is doubled per MI355X_MICROARCH.md (gfx950 reports 1/2 of wide coalesced reads); WRITE_SIZE as read; KB -> bytes x1024"]
res = {}
for path, counter, mult in (("pmc_fetch", "FETCH_SIZE", 2.0), ("pmc_write", "WRITE_SIZE", 1.0)):
    agg = pmc(path, counter)
    lines.append("%s (command: rocprofv3 --kernel-trace --pmc %s -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline)" % (counter, counter))
    for k, (n, v, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        b = v * 1024.0 * mult
        lines.append("  %-44s launches=%5d  bytes=%9.3f GB  kernel_ms=%8.2f  -> %.2f TB/s" % (k, n, b / 1e9, ms, b / ms / 1e9 if ms else 0))
        if k.startswith("gemm"): res[counter] = (n, b)
agg = {c: pmc("pmc_mfma", c) for c in ("SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_INSTS_VALU_MFMA_MOPS_F64", "SQ_BUSY_CU_CYCLES")}
gk = [k for k in agg["SQ_VALU_MFMA_BUSY_CYCLES"] if k.startswith("gemm")][0]
busy = agg["SQ_VALU_MFMA_BUSY_CYCLES"][gk][1]; gui = agg["GRBM_GUI_ACTIVE"][gk][1]; ms = agg["GRBM_GUI_ACTIVE"][gk][2]
lines.append("MFMA utilisation of the GEMM class over one step (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE):")
lines.append("  SQ_VALU_MFMA_BUSY_CYCLES=%.4e (64 per v_mfma_f64_16x16x4_f64, summed over 1024 SIMDs)  GRBM_GUI_ACTIVE=%.4e (summed over 8 XCDs)  kernel_ms=%.2f" % (busy, gui, ms))
lines.append("  -> MFMA pipe busy %.1f %% of SIMD cycles (busy / (GUI_ACTIVE/8 * 1024)); clock %.3f GHz (GUI_ACTIVE/8 / kernel time)" % (100.0 * busy / (gui / 8 * 1024), gui / 8 / ms / 1e6))
open(os.path.join(out, tag + "_pmc_traffic.txt"), "w").write("\n".join(lines) + "\n")
json.dump({"config": "C4 N=32768 d=8 M=32768: the profiled command = bench.py --steps 1 --warmup 0 = one timed step + the fit-only "
                     "and IVAR-only passes (3 step-equivalents); the *_per_step fields are totals over that command, "
                     "bench.py divides bytes by launches (per-launch traffic, like roofline.achieved)",
           "kernel": "gemm class (gemm_f64_kernel variants + leaf_mul kernels)",
           "launches_per_step": res["FETCH_SIZE"][0], "fetch_bytes_per_step_corrected": res["FETCH_SIZE"][1],
           "write_bytes_per_step": res["WRITE_SIZE"][1],
           "correction": "FETCH_SIZE x2 (gfx950 wide coalesced reads), WRITE_SIZE x1, KB x1024; separate --pmc passes",
           "source": "profiles/%s_pmc_traffic.txt" % tag}, open(os.path.join(out, tag + "_pmc_traffic.json"), "w"), indent=1)
print("\n".join(lines))
print(open(os.path.join(out, tag + "_bench_n1.json")).read())
