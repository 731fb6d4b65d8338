#!/bin/bash
# round 2, visit a: GPU tests, DHFR bench, water-1M and apoa1-size single-GPU timings
cd "$(dirname "$0")/.."
R=$(pwd); mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 > gpurun_out/pytest_r2a.log 2>&1; echo "pytest exit $?"; tail -8 gpurun_out/pytest_r2a.log
grep -h "force max-rel-err" gpurun_out/pytest_r2a.log | head
timeout 300 python -m pytest tests/test_gpu_platform.py -m gpu -q -s -k "1e4" --timeout 600 2>&1 | grep -i "max-rel-err\|passed\|failed" | head
timeout 600 python bench.py --steps 3000 --warmup 300 > gpurun_out/bench_r2a.json 2> gpurun_out/bench_r2a.err; echo "bench exit $?"; tail -1 gpurun_out/bench_r2a.json
timeout 600 python bench.py --steps 300 --warmup 50 --workload water1m --cpu-steps 0 > gpurun_out/bench_r2a_w1m.json 2> gpurun_out/bench_r2a_w1m.err; echo "bench1m exit $?"; tail -1 gpurun_out/bench_r2a_w1m.json
