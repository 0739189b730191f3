#!/bin/bash
# usage (on the GPU box): bash scripts/alias_sweep.sh  -- how the set of HIP streams a context owns changes the replayed 2-D step
run() { echo "== GPX_STREAM_ALIAS='$1'"; GPX_STREAM_ALIAS="$1" python scripts/dist_replay.py --grids 1x1,2x4 --ranks 0 --steps 3 --m 1024 --no-stream | tail -2; GPX_STREAM_ALIAS="$1" python scripts/dist_replay.py --grids 2x4 --ranks 0 --steps 3 | tail -1; }
run ""
run "3=4"
run "5=4"
run "3=4,5=4"
run "2=1"
run "2=1,3=4,5=4"
echo "== single GPU"
python scripts/probe_potrf.py 32768
GPX_STREAM_ALIAS="2=1,3=4,5=4" python scripts/probe_potrf.py 32768
GPX_STREAM_ALIAS="2=1,3=1,5=1,4=1" python scripts/probe_potrf.py 32768
