R=${GRAFT_REPO_ROOT:-$PWD}
run() { tag=$1; shift; out=$(env "$@" timeout -k 10 400 python $R/scripts/dist_replay.py --grids ${GRID:-2x4} --no-stream --m 1024 --paced-grid --rows 0 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l)
        print('chain %.1f by_col %s paced_max %.1f paced %s unpaced %s' % (j['chain_ms'], [round(x,1) for x in j['chain_ms_by_process_column']], j['paced_step_ms_max'], j['paced_step_ms'], j['iterations'][0]['rank_step_ms']))
"); echo "== $tag: $out"; }
run default X=1
run agg1 GPX_DIST_AGG=1
run bulk_r4 GPX_DIST_BULK=bulk
run bulk_r8 GPX_DIST_BULK=bulk GPX_CUMASK_RESERVE=0,1,2,3,4,5,6,7
run bulk_r16 GPX_DIST_BULK=bulk GPX_CUMASK_RESERVE=0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15
run bulk_r8_agg2 GPX_DIST_BULK=bulk GPX_DIST_AGG=2 GPX_CUMASK_RESERVE=0,1,2,3,4,5,6,7
run bulk_r16_agg2 GPX_DIST_BULK=bulk GPX_DIST_AGG=2 GPX_CUMASK_RESERVE=0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15
run bulk_r16_agg1 GPX_DIST_BULK=bulk GPX_DIST_AGG=1 GPX_CUMASK_RESERVE=0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15
run bulk_r8_agg1 GPX_DIST_BULK=bulk GPX_DIST_AGG=1 GPX_CUMASK_RESERVE=0,1,2,3,4,5,6,7
