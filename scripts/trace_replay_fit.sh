#!/bin/bash
# Runs ON THE GPU BOX: kernel trace of the replayed rank 0 of a 2x4 grid, factorisation only
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/replay_trace
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format rocpd -d $O/tr -o run -- python3 $R/scripts/dist_replay.py --grids 2x4 --ranks 0 --steps 2 --no-stream --m 1024 > $O/replay.log 2>&1 || exit 3
DB=$(find $O/tr -name "*.db" | head -1)
python3 $R/scripts/trace_summary.py $DB > $O/summary_all.txt 2>&1
python3 - "$DB" > $O/kernels.tsv <<'PY'
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1]); c = db.cursor()
rows = list(c.execute("select name, start, end, stream_id from kernels order by start"))
t0 = rows[0][1]
for n, s, e, st in rows:
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n).split("(")[0][:50]
    print("%.4f\t%.4f\t%s\t%s" % ((s - t0) / 1e6, (e - t0) / 1e6, st, n))
PY
tail -3 $O/replay.log
echo collected
