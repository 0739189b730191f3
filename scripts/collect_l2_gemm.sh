#!/bin/bash
# Runs ON THE GPU BOX: L2 hit rate of isolated GEMM launches (NN / NT / SYRK-lower) -- VERDICT r3 item 7
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/l2
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d $O/pmc_gemm -o run -- python3 $R/scripts/probe_gemm.py "$@" > $O/pmc_gemm.log 2>&1 || exit 3
echo collected
