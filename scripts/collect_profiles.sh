#!/bin/bash
# Runs ON THE GPU BOX (gpurun): bench line, rocprofv3 kernel stats and the separate PMC passes of the same workload.
# Outputs under gpurun_out/final/; summarise afterwards with scripts/summarise_profiles.py and copy into profiles/.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/final
rm -rf $O && mkdir -p $O
cd $R && timeout -k 10 900 python3 bench.py > $O/bench.json 2> $O/bench.err || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-replay > $O/stats.log 2>&1 || exit 2
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o run -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-replay > $O/pmc_fetch.log 2>&1 || exit 3
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o run -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-replay > $O/pmc_write.log 2>&1 || exit 4
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma -o run -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-replay > $O/pmc_mfma.log 2>&1 || exit 5
echo collected
