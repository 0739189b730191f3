#!/bin/bash
# Runs ON THE GPU BOX (gpurun): L2 hit/miss and fabric request counters of one bench step.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/l2
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
P=${1:-hit}
if [ "$P" = hit ]; then
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d $O/pmc_l2 -o run -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/pmc_l2.log 2>&1 || exit 3
else
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d $O/pmc_sizes -o run -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/pmc_sizes.log 2>&1 || exit 4
fi
echo collected
