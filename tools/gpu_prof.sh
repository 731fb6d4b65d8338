#!/bin/bash
# rocprofv3 kernel trace of a short bench run; summaries land in gpurun_out/prof_<tag>/
cd "$(dirname "$0")/.."
R=$(pwd)
TAG=${1:-run}
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$TAG -o trace -- python $R/bench.py --steps ${BENCH_STEPS:-500} --warmup 50 --cpu-steps 0 > $R/gpurun_out/prof_$TAG.log 2>&1
echo "rocprof exit $?"
cd $R
find gpurun_out/prof_$TAG -name "*stats*" | head
f=$(find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -40 "$f"
tail -2 gpurun_out/prof_$TAG.log
