#!/usr/bin/env python3
"""Per-kernel and per-stream summary of a rocprofv3 --kernel-trace database (rocpd .db).
usage: trace_summary.py results.db [t0_ms t1_ms]   -- window in ms relative to the first kernel (default: everything)"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
c = db.cursor()
rows = list(c.execute("select name, start, end, stream_id, queue_id, grid_x, workgroup_x from kernels order by start"))
t00 = rows[0][1]
lo = float(sys.argv[2]) if len(sys.argv) > 2 else -1e30
hi = float(sys.argv[3]) if len(sys.argv) > 3 else 1e30
rows = [r for r in rows if lo <= (r[1] - t00) / 1e6 <= hi]


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return n.split("(")[0][:60]


agg = {}
for n, s, e, st, q, gx, wx in rows:
    a = agg.setdefault(short(n), [0, 0.0, 0.0])
    a[0] += 1
    a[1] += (e - s) / 1e6
    a[2] = max(a[2], (e - s) / 1e3)
print("window %.1f .. %.1f ms, %d kernels" % ((rows[0][1] - t00) / 1e6, (rows[-1][2] - t00) / 1e6, len(rows)))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
    print("%-62s n=%6d tot=%9.2f ms avg=%8.1f us max=%9.1f us" % (k, v[0], v[1], 1e3 * v[1] / v[0], v[2]))
# per stream: busy time (union of intervals) and span
by = {}
for n, s, e, st, q, gx, wx in rows:
    by.setdefault(st, []).append((s, e))
print("\nper stream: kernels, busy (union) ms, first..last ms")
for st, iv in sorted(by.items()):
    iv.sort()
    busy, cs, ce = 0, iv[0][0], iv[0][1]
    for s, e in iv[1:]:
        if s > ce:
            busy += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    busy += ce - cs
    print("stream %3s  n=%6d busy=%9.2f  %9.2f .. %9.2f" % (st, len(iv), busy / 1e6, (iv[0][0] - t00) / 1e6, (iv[-1][1] - t00) / 1e6))
allv = sorted((s, e) for _, s, e, *_ in rows)
busy, cs, ce = 0, allv[0][0], allv[0][1]
for s, e in allv[1:]:
    if s > ce:
        busy += ce - cs
        cs, ce = s, e
    else:
        ce = max(ce, e)
busy += ce - cs
print("device busy (union over streams) %.2f ms of %.2f ms" % (busy / 1e6, (allv[-1][1] - allv[0][0]) / 1e6))
