"""Durations of the kfill launches of a rocprofv3 --kernel-trace run, in launch order, with the idle time before each.
usage: fill_sequences_report.py <output dir of rocprofv3>"""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
prev_end = None
for r in rows:
    m = re.search(r"(kfill_\w+)<([^>]*)>", r["Kernel_Name"])
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if m:
        print("%s<%s> %.3f ms   idle before: %.3f ms" % (m.group(1), m.group(2), (e - s) / 1e6, (s - prev_end) / 1e6 if prev_end else -1))
    prev_end = e
