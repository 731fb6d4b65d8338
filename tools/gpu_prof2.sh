#!/bin/bash
# rocprofv3 kernel trace of a short bench run with extra bench args; summaries land in gpurun_out/prof_<tag>/
cd "$(dirname "$0")/.."
R=$(pwd); TAG=${1:-run}; shift
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$TAG -o trace -- python $R/bench.py --steps 500 --warmup 50 --cpu-steps 0 "$@" > $R/gpurun_out/prof_$TAG.log 2>&1
echo "rocprof exit $?"
