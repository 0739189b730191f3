R=${GRAFT_REPO_ROOT:-$PWD}
run() { tag=$1; shift; out=$(env "$@" timeout -k 10 400 python $R/scripts/dist_replay.py --grids ${GRID:-2x4} --no-stream --m 1024 --paced-grid 2>&1 | python3 -c "
import sys, json
ok=False
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); ok=True
        print('chain %.1f paced_max %.1f paced %s last_arrived %s' % (j['chain_ms'], j['paced_step_ms_max'], j['paced_step_ms'], j['last_panel_arrived_ms']))
    last=l
if not ok: print('FAILED', last[:300])
"); echo "== $tag: $out"; }
run ring4_agg2 GPX_DIST_RING=4 GPX_DIST_AGG=2
run ring8_agg2 GPX_DIST_RING=8 GPX_DIST_AGG=2
run ring16_agg2 GPX_DIST_RING=16 GPX_DIST_AGG=2
run ring16_agg4 GPX_DIST_RING=16 GPX_DIST_AGG=4
run ring8_agg1 GPX_DIST_RING=8 GPX_DIST_AGG=1
