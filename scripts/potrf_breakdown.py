"""Pair the GEMM/leaf launches of one recursive potrf (model of chol.hip's recursion) with a rocprofv3 kernel trace and
report time / TFLOP/s per call class.  usage: potrf_breakdown.py kernel_trace.csv N [which_potrf=last]"""
import csv, sys, collections
NB = 128


def split(n):
    return (n // NB // 2) * NB


calls = []  # (kind, m, n, k, flops)


def gemm(m, n, k, lower=False, tag=""):
    if m == 0 or n == 0:
        return
    t = 0.5 * (m // 128) * (m // 128 + 1) if lower else (m // 128) * (n // 128)
    calls.append((tag, m, n, k, 2.0 * t * 128 * 128 * k))


def trsm_right(m, n):
    if n == NB:
        gemm(m, NB, NB, tag="trsm-leaf")
        return
    n1 = split(n); n2 = n - n1
    trsm_right(m, n1)
    gemm(m, n2, n1, tag="trsm-upd")
    trsm_right(m, n2)


RL_MAX = 4096  # diagonal blocks up to this order are factored right-looking (chol.hip: potrf_right_looking)


def potrf(n):
    if n == NB:
        calls.append(("leaf", NB, NB, NB, 2.0 * NB ** 3 / 3))
        return
    if n <= RL_MAX:
        for j in range(0, n, NB):
            calls.append(("leaf", NB, NB, NB, 2.0 * NB ** 3 / 3))
            m = n - j - NB
            if m > 0:
                gemm(m, NB, NB, tag="trsm-leaf")
                gemm(m, m, NB, lower=True, tag="syrk-rl")
        return
    n1 = split(n); n2 = n - n1
    potrf(n1)
    trsm_right(n2, n1)
    gemm(n2, n2, n1, lower=True, tag="syrk")
    potrf(n2)


N = int(sys.argv[2])
potrf(N)
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        nm = r["Kernel_Name"]
        if "gemm_f64" in nm or "leaf_kernel" in nm or "leaf_mul" in nm:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "leaf" if "leaf_kernel" in nm else "gemm"))
rows.sort()
reps = len(rows) // len(calls)
assert reps * len(calls) == len(rows), (len(rows), len(calls))
rows = rows[(reps - 1) * len(calls):]
agg = collections.OrderedDict()
for (tag, m, n, k, fl), (s, e, kind) in zip(calls, rows):
    assert (tag == "leaf") == (kind == "leaf"), (tag, kind)
    key = (tag, m, n, k)
    a = agg.setdefault(key, [0, 0.0, 0.0])
    a[0] += 1; a[1] += (e - s) / 1e6; a[2] += fl
wall = (rows[-1][1] - rows[0][0]) / 1e6
busy = sum(a[1] for a in agg.values())
print("potrf N=%d: %d launches, wall %.1f ms, kernel-busy %.1f ms (gaps %.1f ms)" % (N, len(rows), wall, busy, wall - busy))
print("%-10s %6s %6s %6s %6s %9s %8s %7s" % ("class", "m", "n", "k", "calls", "total ms", "avg us", "TF/s"))
for (tag, m, n, k), (c, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print("%-10s %6d %6d %6d %6d %9.2f %8.1f %7.1f" % (tag, m, n, k, c, ms, 1e3 * ms / c, fl / ms / 1e9))
bytag = collections.defaultdict(lambda: [0, 0.0, 0.0])
for (tag, m, n, k), (c, ms, fl) in agg.items():
    b = bytag[tag]; b[0] += c; b[1] += ms; b[2] += fl
for tag, (c, ms, fl) in bytag.items():
    print("== %-10s calls=%5d total=%8.2f ms  %.1f TF/s" % (tag, c, ms, fl / ms / 1e9))
