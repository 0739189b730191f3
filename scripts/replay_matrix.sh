#!/bin/bash
# usage (GPU box): bash scripts/replay_matrix.sh -- correctness sweep of the replayed 2-D step (window of the factor, fused forward
# substitution, group inverses) over block sizes, group sizes and grids at a small size; every run asserts its rank's streamed
# variance sum against the single-GPU path (1e-10)
fail=0
for cfg in "128 1" "128 3" "128 8" "256 2" "256 5" "512 4" "512 3" "1024 2"; do
  set -- $cfg
  for grids in "1x2,2x2" "2x4,4x2" "1x4,2x3"; do
    out=$(timeout -k 10 300 python scripts/dist_replay.py --n 6000 --m 3000 --d 5 --nb $1 --agg $2 --grids $grids --ranks 0,last --steps 1 2>&1 | tail -5)
    if echo "$out" | grep -q "GB received"; then echo "ok   nb=$1 agg=$2 grids=$grids"; else echo "FAIL nb=$1 agg=$2 grids=$grids"; echo "$out"; fail=1; fi
  done
done
exit $fail
