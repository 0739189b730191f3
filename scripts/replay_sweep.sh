#!/bin/bash
# usage (on the GPU box): bash scripts/replay_sweep.sh OUTDIR   -- replay variants of the 2-D loop, one line each
out=$1; mkdir -p $out
run() { tag=$1; shift; env "$@" timeout -k 10 300 python scripts/dist_replay.py --grids ${GRIDS:-2x2,2x4} --ranks 0 --steps 2 $ARGS > $out/$tag.log 2>&1; echo "== $tag rc=$?"; tail -3 $out/$tag.log; }
ARGS="--agg 4" run small512 X=1
ARGS="--agg 4" run small1024 GPX_DIST2_SMALL_MAX=1024
ARGS="--agg 4" run small2048 GPX_DIST2_SMALL_MAX=2048
ARGS="--agg 4" run small256 GPX_DIST2_SMALL_MAX=256
ARGS="--agg 4" run invmin0 GPX_DIST2_INV_MIN=0
ARGS="--agg 4" run ivarback GPX_DIST_IVAR_STREAM=back
