#!/bin/bash
# usage (on the GPU box): bash scripts/replay_sweep.sh OUTDIR   -- replay variants of the 2-D loop, one line each
out=$1; mkdir -p $out
run() { tag=$1; shift; env "$@" timeout -k 10 300 python scripts/dist_replay.py --grids ${GRIDS:-1x1,2x4} --ranks 0 --steps 2 $ARGS > $out/$tag.log 2>&1; echo "== $tag rc=$?"; tail -3 $out/$tag.log; }
export GPX_DIST_BULK_STREAM=eval GPX_DIST_IVAR_STREAM=eval
ARGS="--agg 4" run base X=1
ARGS="--agg 4" run hwq8 GPU_MAX_HW_QUEUES=8
ARGS="--agg 4" run hwq2 GPU_MAX_HW_QUEUES=2
ARGS="--agg 4" run hwq8_main GPU_MAX_HW_QUEUES=8 GPX_DIST_BULK_STREAM=main
