R=${GRAFT_REPO_ROOT:-$PWD}
run() { tag=$1; nb=$2; shift; shift; out=$(env "$@" timeout -k 10 400 python $R/scripts/dist_replay.py --grids ${GRID:-2x4} --no-stream --m 1024 --paced-grid --rows 0 --nb $nb 2>&1 | python3 -c "
import sys, json
ok=False
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); ok=True
        print('chain %.1f by_col %s paced_max %.1f paced %s unpaced %s' % (j['chain_ms'], [round(x,1) for x in j['chain_ms_by_process_column']], j['paced_step_ms_max'], j['paced_step_ms'], j['iterations'][0]['rank_step_ms']))
    last=l
if not ok: print('FAILED', last[:300])
"); echo "== $tag: $out"; }
run nb512_agg2 512 GPX_DIST_AGG=2
run nb1024_agg1 1024 GPX_DIST_AGG=1
run nb1024_agg2 1024 GPX_DIST_AGG=2
run nb768_agg2 768 GPX_DIST_AGG=2
run nb256_agg4 256 GPX_DIST_AGG=4
