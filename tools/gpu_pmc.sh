#!/bin/bash
# rocprofv3 PMC pass (own run, kernel-trace only) over a short bench; usage: gpu_pmc.sh <tag> "<counters>"
cd "$(dirname "$0")/.."
R=$(pwd); TAG=${1:-pmc}; shift
CTRS=${*:-SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY}
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$TAG -o pmc -- python $R/bench.py --steps ${BENCH_STEPS:-60} --warmup 10 --cpu-steps 0 --no-roofline --no-scale-workload > $R/gpurun_out/pmc_$TAG.log 2>&1
echo "rocprof exit $?"; ls $R/gpurun_out/pmc_$TAG | head
