"""Pair the GEMM launches of the IVAR triangular solve (model of chol_trsm_left) with a rocprofv3 kernel trace.
usage: ivar_breakdown.py kernel_trace.csv N M   (trace of scripts/probe_ivar.py: one potrf, then two ivar calls)"""
import csv, sys, collections
NB = 128
def split(n): return (n // NB // 2) * NB
calls = []
def trsm_left(n, m):
    if n == NB:
        calls.append(("leaf", NB, m, NB, 2.0 * NB * m * NB)); return
    n1 = split(n); n2 = n - n1
    trsm_left(n1, m)
    calls.append(("upd", n2, m, n1, 2.0 * n2 * m * n1))
    trsm_left(n2, m)
N, M = int(sys.argv[2]), int(sys.argv[3])
trsm_left(N, M)
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        if "gemm_f64" in r["Kernel_Name"] or "leaf_mul" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
rows.sort()
rows = rows[-len(calls):]   # the last ivar call
agg = collections.OrderedDict()
for (tag, m, n, k, fl), (s, e) in zip(calls, rows):
    a = agg.setdefault((tag, m, n, k), [0, 0.0, 0.0]); a[0] += 1; a[1] += (e - s) / 1e6; a[2] += fl
wall = (rows[-1][1] - rows[0][0]) / 1e6; busy = sum(a[1] for a in agg.values())
print("trsm_left N=%d M=%d: %d launches, wall %.1f ms, gemm-busy %.1f ms" % (N, M, len(rows), wall, busy))
for (tag, m, n, k), (c, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-5s m=%6d n=%6d k=%6d calls=%4d total=%8.2f ms avg=%9.1f us  %.1f TF/s" % (tag, m, n, k, c, ms, 1e3 * ms / c, fl / ms / 1e9))
