"""Mean counter values per kernel template instance from a rocprofv3 --pmc run.  usage: pmc_by_kernel.py <dir> [name filter]"""
import csv, glob, re, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
flt = sys.argv[2] if len(sys.argv) > 2 else "kfill_kernel"
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for r in csv.DictReader(open(f)):
    if flt not in r["Kernel_Name"]: continue
    m = re.search(r"(\w+)<([^>]*)>", r["Kernel_Name"])
    key = (m.group(1) + "<" + m.group(2) + ">") if m else r["Kernel_Name"][:60]
    a = agg[key][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for k in sorted(agg):
    print(k)
    for c in sorted(agg[k]):
        n, v = agg[k][c]
        print("    %-28s %.4g per launch (%d launches)" % (c, v / n, n))
