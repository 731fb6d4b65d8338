#!/bin/bash
# One GPU-box visit: pytest -m gpu, a bench run, and a rocprofv3 kernel trace of the same bench command.
cd "$(dirname "$0")/.."
R=$(pwd)
TAG=${1:-run}
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
if [ -z "$SKIP_TESTS" ]; then
  timeout 1500 python -m pytest tests -m gpu -q -x --timeout 600 ${PYTEST_ARGS} > gpurun_out/pytest_$TAG.log 2>&1; echo "pytest exit $?"; tail -5 gpurun_out/pytest_$TAG.log
fi
if [ -z "$SKIP_TESTS" ]; then timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -2; fi
timeout 900 python bench.py --steps ${BENCH_STEPS:-3000} --warmup 300 ${BENCH_ARGS} > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; echo "bench exit $?"; tail -1 gpurun_out/bench_$TAG.json
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$TAG -o trace -- python $R/bench.py --steps ${BENCH_STEPS:-3000} --warmup 300 --cpu-steps 0 ${BENCH_ARGS} > $R/gpurun_out/prof_$TAG.log 2>&1
echo "rocprof exit $?"
cd $R
f=$(find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -30 "$f"
tail -1 gpurun_out/prof_$TAG.log
