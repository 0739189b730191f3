#!/bin/bash
# Runs ON THE GPU BOX: paced single-rank replays (scripts/dist_replay.py --paced) over schedule knobs; one line per variant
R=${GRAFT_REPO_ROOT:-$PWD}
run() { tag=$1; shift; out=$(env "$@" timeout -k 10 200 python $R/scripts/dist_replay.py --grids ${GRID:-2x4} --ranks ${RANKS:-0} --no-stream --m 1024 --paced 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l)
        print('rank %d busy %.1f paced %.1f chain_est %.1f own_first_last %s' % (j['rank'], j['unpaced_rank_busy_ms'], j['paced_step_ms'], j['iterations'][-1]['chain_estimate_ms'], j['own_latency_ms_first_last']))
"); echo "== $tag: $out"; }
"$@"
