#!/bin/bash
# usage (on the GPU box): bash scripts/replay_sweep.sh OUTDIR   -- replay variants of the 2-D loop, one line each
out=$1; mkdir -p $out
run() { tag=$1; shift; env "$@" timeout -k 10 300 python scripts/dist_replay.py --grids ${GRIDS:-2x2,2x4} --ranks 0 --steps 2 $ARGS > $out/$tag.log 2>&1; echo "== $tag rc=$?"; tail -2 $out/$tag.log; }
ARGS="--nb 512 --agg 4" run nb512_agg4 X=1
ARGS="--nb 1024 --agg 2" run nb1024_agg2 X=1
ARGS="--nb 1024 --agg 4" run nb1024_agg4 X=1
ARGS="--nb 2048 --agg 2" run nb2048_agg2 X=1
ARGS="--nb 512 --agg 4 --m 1024" run fit_nb512_agg4 X=1
ARGS="--nb 1024 --agg 2 --m 1024" run fit_nb1024_agg2 X=1
ARGS="--nb 1024 --agg 4 --m 1024" run fit_nb1024_agg4 X=1
ARGS="--nb 2048 --agg 2 --m 1024" run fit_nb2048_agg2 X=1
