#!/bin/bash
# round 2, session 2, visit g: reciprocal space on its own stream is now the default above 60 000 atoms -- whole GPU suite again, then
# the bench lines of the three sizes it changes and the default / driver lines (unchanged path) on the same box
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
T=r3g
timeout 1200 python -m pytest tests -m gpu -q -x --timeout 900 > gpurun_out/pytest_$T.log 2>&1; echo "pytest exit $?"; grep -n "passed\|failed" gpurun_out/pytest_$T.log | tail -2
for wl in apoa1 water98k water1m; do
  steps=1000; [ $wl = water1m ] && steps=300
  python bench.py --steps $steps --warmup 100 --workload $wl --cpu-steps 0 --no-scale-workload 2>/dev/null | tail -1 > gpurun_out/bench_${T}_$wl.json; cut -c1-200 gpurun_out/bench_${T}_$wl.json
done
( time timeout 900 python bench.py > gpurun_out/bench_${T}_default.json 2> gpurun_out/bench_${T}_default.err ) 2>&1 | grep real; tail -1 gpurun_out/bench_${T}_default.json | cut -c1-250
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_${T}_driver.json 2> gpurun_out/bench_${T}_driver.err ) 2>&1 | grep real; tail -1 gpurun_out/bench_${T}_driver.json | cut -c1-250
