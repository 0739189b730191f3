"""Summarise a rocprofv3 kernel_trace.csv: per kernel name x grid size -> calls, total ms, avg us."""
import csv, sys, collections
rows = collections.defaultdict(lambda: [0, 0.0])
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = r["Kernel_Name"].split("(")[0].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
        if "<" in r["Kernel_Name"]:
            name = r["Kernel_Name"].split("(anonymous namespace)::")[-1].split("(")[0]
        key = (name, int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1), int(r["Grid_Size_Y"]))
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        rows[key][0] += 1
        rows[key][1] += d
tot = sum(v[1] for v in rows.values())
lim = int(sys.argv[2]) if len(sys.argv) > 2 else 40
print("total kernel ms %.2f" % tot)
for k, v in sorted(rows.items(), key=lambda kv: -kv[1][1])[:lim]:
    print("%-40s grid=(%5d,%5d) calls=%5d total=%9.3f ms avg=%9.1f us  %.1f%%" % (k[0][:40], k[1], k[2], v[0], v[1], 1e3 * v[1] / v[0], 100 * v[1] / tot))
