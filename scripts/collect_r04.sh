#!/bin/bash
# Runs ON THE GPU BOX (gpurun): the round-4 artefacts beyond scripts/collect_profiles.sh -- L2 hit rates, potrf phases, the paced
# whole-grid replays, a step timeline, the world-1 RCCL bench.  Outputs under gpurun_out/r04/.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04
rm -rf $O && mkdir -p $O
cd $R
GPX_POTRF_TIMING=1 timeout -k 10 200 python3 scripts/probe_potrf.py 32768 > $O/potrf_phases.txt 2>&1
timeout -k 10 300 python3 scripts/dist_replay.py --grids 1x2,2x2,2x4 --no-stream --m 1024 --paced-grid --rows 0 > $O/paced_fit.jsonl 2> $O/paced_fit.err
timeout -k 10 400 python3 scripts/dist_replay.py --grids 2x2,2x4 --paced-grid --rows 0 > $O/paced_fit_ivar.jsonl 2> $O/paced_fit_ivar.err
GPX_DIST2_HOIST_INV=0 GPX_DIST_GATE_BULK=0 GPX_DIST_AGG=4 timeout -k 10 300 python3 scripts/dist_replay.py --grids 2x4 --no-stream --m 1024 --paced-grid --rows 0 > $O/paced_fit_round3_schedule.jsonl 2> /dev/null
timeout -k 10 200 python3 scripts/dist_replay.py --grids 1x1,1x2,2x2,2x4 --ranks 0,last --steps 3 --no-stream --m 1024 > $O/replay_unpaced_fit.txt 2>&1
timeout -k 10 200 python3 scripts/dist_timeline.py > $O/timeline_2x4_unpaced.txt 2>&1
GPX_FORCE_DIST=1 timeout -k 10 400 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_dist_world1_rccl.json 2> $O/bench_dist_world1_rccl.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d $O/pmc_l2 -o run -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-replay > $O/pmc_l2.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d $O/pmc_gemm -o run -- python3 $R/scripts/probe_gemm.py 24576,24576,4096,0,1,0 24576,24576,4096,1,1,0 24576,24576,4096,1,1,1 24576,24576,4096,0,1,1 16384,8192,16384,0,1,0 > $O/pmc_gemm.log 2>&1
echo collected
