#!/bin/bash
# round 2, visit e: real DHFR (tests, 2 fs and 4 fs bench), decomposed one-rank run with the side stream, re-sort timing
cd "$(dirname "$0")/.."
R=$(pwd); mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_platform.py tests/test_gpu_multirank.py -m gpu -q --timeout 600 -s -k "real_dhfr or rccl or sharing" > gpurun_out/pytest_r2e.log 2>&1; echo "pytest exit $?"; grep -h "max-rel-err\|RCCL vs\|DHFR at\|passed\|failed\|Error" gpurun_out/pytest_r2e.log | cut -c1-300 | head
timeout 600 python bench.py --steps 3000 --warmup 300 --no-scale-workload > gpurun_out/bench_r2e.json 2> gpurun_out/bench_r2e.err; echo "bench exit $?"; tail -1 gpurun_out/bench_r2e.json | cut -c1-420
timeout 600 python bench.py --steps 3000 --warmup 300 --no-scale-workload --dt-fs 4 --cpu-steps 0 > gpurun_out/bench_r2e_4fs.json 2> gpurun_out/bench_r2e_4fs.err; echo "bench 4fs exit $?"; tail -1 gpurun_out/bench_r2e_4fs.json | cut -c1-300
timeout 600 python bench.py --steps 3000 --warmup 300 --no-scale-workload --workload dhfr_like --cpu-steps 0 > gpurun_out/bench_r2e_like.json 2> gpurun_out/bench_r2e_like.err; echo "bench like exit $?"; tail -1 gpurun_out/bench_r2e_like.json | cut -c1-300
OPENMM_HIP_TIMING=1 timeout 400 python bench.py --steps 300 --warmup 20 --workload water1m --cpu-steps 0 --no-scale-workload --decompose > gpurun_out/bench_r2e_w1m_dd1.json 2> gpurun_out/bench_r2e_w1m_dd1.err; echo "w1m dd1 exit $?"; grep "^{" gpurun_out/bench_r2e_w1m_dd1.json | cut -c1-330; grep "re-sort\|rror" gpurun_out/bench_r2e_w1m_dd1.err | head -5
OPENMM_HIP_TIMING=1 timeout 400 python bench.py --steps 300 --warmup 20 --workload water1m --cpu-steps 0 --no-scale-workload > gpurun_out/bench_r2e_w1m.json 2> gpurun_out/bench_r2e_w1m.err; echo "w1m exit $?"; grep "^{" gpurun_out/bench_r2e_w1m.json | cut -c1-330; grep "re-sort" gpurun_out/bench_r2e_w1m.err | head -3
