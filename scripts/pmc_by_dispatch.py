"""Per-dispatch counter table of a rocprofv3 --pmc run (csv): python3 scripts/pmc_by_dispatch.py <dir> [name filter]
Prints every counter the run collected (summed over the XCDs' instances), one line per dispatch."""
import csv, glob, sys, collections
d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else "gemm"
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
rows = collections.OrderedDict()
names = []
for r in csv.DictReader(open(f)):
    key = int(r["Dispatch_Id"])
    e = rows.setdefault(key, {"name": r["Kernel_Name"]})
    e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    if r["Counter_Name"] not in names:
        names.append(r["Counter_Name"])
for key, e in rows.items():
    if flt not in e["name"]:
        continue
    name = e["name"].replace("(anonymous namespace)::", "").replace("void ", "")[:44]
    extra = ""
    if "TCC_HIT_sum" in e and "TCC_MISS_sum" in e:
        extra = "  hit rate %.3f" % (e["TCC_HIT_sum"] / max(e["TCC_HIT_sum"] + e["TCC_MISS_sum"], 1))
    print("%5d %-44s " % (key, name) + "  ".join("%s %.4e" % (n.replace("_sum", ""), e.get(n, 0.0)) for n in names) + extra)
