"""Per-dispatch counter table of a rocprofv3 --pmc run (csv): python3 scripts/pmc_by_dispatch.py <dir> [name filter]"""
import csv, glob, sys, collections
d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else "gemm"
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
rows = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    key = int(r["Dispatch_Id"])
    e = rows.setdefault(key, {"name": r["Kernel_Name"]})
    e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
for key, e in rows.items():
    if flt not in e["name"]:
        continue
    hit, miss, ea = e.get("TCC_HIT_sum", 0), e.get("TCC_MISS_sum", 0), e.get("TCC_EA0_RDREQ_sum", 0)
    name = e["name"].replace("(anonymous namespace)::", "").replace("void ", "")[:48]
    print("%5d %-48s hit %.3e miss %.3e rate %.3f  EA reads %.3e (%.1f GB)" % (key, name, hit, miss, hit / max(hit + miss, 1), ea, ea * 128 / 1e9))
