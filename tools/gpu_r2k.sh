#!/bin/bash
# round 2, visit k: decomposed path after the compact wire format (one rank over RCCL, two processes, four ranks against the 1M golden),
# one-rank RCCL timing at 1M atoms
cd "$(dirname "$0")/.."
R=$(pwd); mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests/test_gpu_multirank.py -m gpu -q --timeout 1200 -s > gpurun_out/pytest_r2k.log 2>&1; echo "pytest exit $?"; grep -h "RCCL vs\|forces\|water-1M\|passed\|failed\|rror" gpurun_out/pytest_r2k.log | cut -c1-300 | head
timeout 400 python bench.py --steps 300 --warmup 20 --workload water1m --cpu-steps 0 --no-scale-workload --decompose > gpurun_out/bench_r2k_w1m_dd1.json 2> gpurun_out/bench_r2k_w1m_dd1.err; echo "w1m dd1 exit $?"; grep "^{" gpurun_out/bench_r2k_w1m_dd1.json | cut -c100-200
