R=${GRAFT_REPO_ROOT:-$PWD}
run() { tag=$1; shift; out=$(env "$@" timeout -k 10 600 python $R/scripts/dist_replay.py --grids ${GRID:-2x4} --paced-grid --rows 0 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l)
        print('chain %.1f by_col %s paced_max %.1f paced %s unpaced %s' % (j['chain_ms'], [round(x,1) for x in j['chain_ms_by_process_column']], j['paced_step_ms_max'], j['paced_step_ms'], j['iterations'][0]['rank_step_ms']))
"); echo "== $tag: $out"; }
run stream_agg4 GPX_DIST_AGG=4
run stream_agg2 GPX_DIST_AGG=2
run stream_agg4_nohoist GPX_DIST_AGG=4 GPX_DIST2_HOIST_INV=0 GPX_DIST_GATE_BULK=0
GRID=2x2 run stream_2x2_agg4 GPX_DIST_AGG=4
