#!/bin/bash
# Runs ON THE GPU BOX: the all-ranks paced replay of the 2x4 grid (scripts/dist_replay.py --paced-grid, factorisation only, C4)
# under schedule knobs, one line per variant.  usage: paced_grid_sweep.sh tag1 "ENV=.. ENV=.." "extra args" tag2 ...
R=${GRAFT_REPO_ROOT:-$PWD}
while [ $# -ge 3 ]; do
  tag=$1; envs=$2; extra=$3; shift 3
  out=$(env $envs timeout -k 10 240 python3 $R/scripts/dist_replay.py --grids 2x4 --no-stream --m 1024 --paced-grid --iters ${ITERS:-6} $extra 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l)
        print('nb %d agg %d paced max %.2f (%s) chain %.2f by column %s single potrf %.1f iters %s' % (j['nb'], j['agg'], j['paced_step_ms_max'], ' '.join('%.1f' % v for v in j['paced_step_ms'].values()), j['chain_ms'], [round(v, 1) for v in j['chain_ms_by_process_column']], j['single_gpu_potrf_ms'], [round(i['chain_ms'], 1) for i in j['iterations']]))
")
  echo "== $tag: $out"
done
