R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for sk in 0 16 32 48 80 144 272 528; do
  rm -rf /tmp/skp; GPX_LD_SKEW=$sk timeout -k 10 200 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d /tmp/skp -o run -- python3 $R/scripts/probe_gemm.py 16384,16384,4096,1,1,0 16384,16384,4096,0,1,0 > /tmp/skp.log 2>&1
  python3 - $sk <<'PY'
import csv, sys, re, collections, glob
f=glob.glob('/tmp/skp/**/*counter_collection.csv', recursive=True)[0]
rows=collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if "gemm_f64" not in r["Kernel_Name"]: continue
    key=(int(r["Dispatch_Id"]), re.search(r"<([^>]*)>", r["Kernel_Name"]).group(1))
    rows.setdefault(key,{})[r["Counter_Name"]]=float(r["Counter_Value"])
last={}
for k,v in rows.items(): last[k[1]]=v
print("skew", sys.argv[1], {k: round(v["TCC_HIT_sum"]/(v["TCC_HIT_sum"]+v["TCC_MISS_sum"]),3) for k,v in last.items()})
PY
  grep "gemm m" /tmp/skp.log | awk 'NR%3==0'
done
