R=${GRAFT_REPO_ROOT:-$PWD}
run() { tag=$1; shift; out=$(env "$@" timeout -k 10 400 python $R/scripts/dist_replay.py --grids ${GRID:-2x4} --no-stream --m 1024 --paced-grid 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l)
        print('chain %.1f by_col %s paced_max %.1f paced %s it %s' % (j['chain_ms'], [round(x,1) for x in j['chain_ms_by_process_column']], j['paced_step_ms_max'], j['paced_step_ms'], [round(h['chain_ms'],1) for h in j['iterations']]))
"); echo "== $tag: $out"; }
run default X=1
run agg1 GPX_DIST_AGG=1
run agg2 GPX_DIST_AGG=2
run chunks GPX_DIST_BULK=chunks
