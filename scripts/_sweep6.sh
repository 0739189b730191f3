R=${GRAFT_REPO_ROOT:-$PWD}
run() { tag=$1; shift; out=$(env "$@" timeout -k 10 400 python $R/scripts/dist_replay.py --grids ${GRID:-2x4} --no-stream --m 1024 --paced-grid --rows 0 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l)
        print('chain %.1f by_col %s paced_max %.1f paced %s unpaced %s' % (j['chain_ms'], [round(x,1) for x in j['chain_ms_by_process_column']], j['paced_step_ms_max'], j['paced_step_ms'], j['iterations'][0]['rank_step_ms']))
"); echo "== $tag: $out"; }
run agg2_gate GPX_DIST_AGG=2
run agg2_nogate GPX_DIST_AGG=2 GPX_DIST_GATE_BULK=0
run agg4_gate GPX_DIST_AGG=4
run agg1_gate GPX_DIST_AGG=1
