"""Break a rocprofv3 kernel trace of scripts/probe_potrf.py (last factorisation) down by hardware queue: the main stream's
kernel classes and idle gaps, the look-ahead chain on the side stream, the chunk on the CU-masked stream.
usage: potrf_lookahead_trace.py <kernel_trace.csv>"""
import csv, collections, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r['Queue_Id'],
                     int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1)))
rows.sort()
idx = [i for i, r in enumerate(rows) if 'kfill_kernel' in r[2]]
seg = rows[idx[-1] + 1:]
end = [i for i, r in enumerate(seg) if 'logdet' in r[2]]
seg = seg[:end[0]] if end else seg
t0 = seg[0][0]
print("last factorisation: %d kernels, wall %.2f ms" % (len(seg), (seg[-1][1] - t0) / 1e6))
byq = collections.defaultdict(list)
for r in seg:
    byq[r[3]].append(r)
order = sorted(byq, key=lambda q: -sum(e - s for s, e, *_ in byq[q]))
for q in order:
    l = byq[q]
    print("queue %s: %4d kernels, busy %7.2f ms, span %.2f-%.2f ms" % (q, len(l), sum(e - s for s, e, *_ in l) / 1e6,
                                                                      (l[0][0] - t0) / 1e6, (l[-1][1] - t0) / 1e6))
main = byq[order[0]]
agg = collections.defaultdict(lambda: [0, 0.0])
for s, e, n, q, g in main:
    k = ('copy2d' if 'copy2d' in n else 'leaf_mul' if 'leaf_mul' in n else 'leaf' if 'leaf_kernel' in n else
         'gemm_batched' if 'batched' in n else 'gemm' if 'gemm' in n else n.split('(')[0][-28:])
    agg[(k, g)][0] += 1
    agg[(k, g)][1] += (e - s) / 1e6
print("main queue by kernel / workgroup count:")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:16]:
    print("  %-14s wgs=%6d calls=%4d total=%7.2f ms avg=%9.1f us" % (k[0], k[1], v[0], v[1], 1e3 * v[1] / v[0]))
gaps = [(round((a[1] - t0) / 1e6, 1), round((b[0] - a[1]) / 1e6, 2)) for a, b in zip(main, main[1:]) if b[0] - a[1] > 200_000]
print("main queue idle gaps > 0.2 ms (at ms, length ms):", gaps, "sum %.1f ms" % sum(g for _, g in gaps))
if len(order) > 1:
    side = byq[order[1]]
    chains = [[side[0]]]
    for a, b in zip(side, side[1:]):
        if b[0] - a[1] > 1_000_000:
            chains.append([])
        chains[-1].append(b)
    for c in chains:
        a2 = collections.defaultdict(lambda: [0, 0.0])
        for s, e, n, q, g in c:
            k = 'leaf' if 'leaf_kernel' in n else 'mul' if 'leaf_mul' in n else 'binv' if ('batched' in n or 'binv' in n) else 'upd' if 'gemm' in n else 'other'
            a2[k][0] += 1
            a2[k][1] += (e - s) / 1e3
        print("chain at %6.1f ms: span %.2f ms, %3d kernels: %s" % ((c[0][0] - t0) / 1e6, (c[-1][1] - c[0][0]) / 1e6, len(c),
              ", ".join("%s %d x = %.0f us" % (k, v[0], v[1]) for k, v in sorted(a2.items()))))
if len(order) > 2:
    print("masked-stream chunks (at ms, length ms):", [(round((s - t0) / 1e6, 1), round((e - s) / 1e6, 2)) for s, e, *_ in byq[order[2]]])
