#!/bin/bash
# A/B of the 128-leaf with the round-5 diagonal step (GPX_LEAF_DIAG=1, default) and rounds 1-4's (0): kernel durations from
# rocprofv3's trace of scripts/probe_leaf.py, and potrf(256..2048) on the host clock.
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
for m in 1 0; do
  export GPX_LEAF_DIAG=$m
  rocprofv3 --kernel-trace --stats -d $OUT/leaf_ab_$m -o leaf -- python3 $GRAFT_REPO_ROOT/scripts/probe_leaf.py > $OUT/leaf_ab_$m.txt 2>&1 || exit 1
  python3 $GRAFT_REPO_ROOT/scripts/probe_small_potrf.py > $OUT/small_potrf_ab_$m.txt 2>&1 || exit 1
done
grep -h "leaf_kernel" $OUT/leaf_ab_1/*/*kernel_stats.csv $OUT/leaf_ab_0/*/*kernel_stats.csv 2>/dev/null || grep -rh "leaf_kernel" $OUT/leaf_ab_1 $OUT/leaf_ab_0 | head
