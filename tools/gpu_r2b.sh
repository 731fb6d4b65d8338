#!/bin/bash
# round 2, visit b: decomposed path on one GPU (RCCL one rank, two ranks over gloo), 1M parity, water-1M timings + kernel trace
cd "$(dirname "$0")/.."
R=$(pwd); mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_multirank.py tests/test_gpu_platform.py -m gpu -q -x --timeout 600 -s -k "rccl or sharing or water1m" > gpurun_out/pytest_r2b.log 2>&1; echo "pytest exit $?"; grep -h "max-rel-err\|RCCL vs\|forces\|passed\|failed\|Error\|error" gpurun_out/pytest_r2b.log | head -20
timeout 400 python bench.py --steps 200 --warmup 20 --workload water1m --cpu-steps 0 --no-scale-workload > gpurun_out/bench_r2b_w1m.json 2> gpurun_out/bench_r2b_w1m.err; echo "w1m exit $?"; tail -1 gpurun_out/bench_r2b_w1m.json | cut -c1-1200
timeout 400 python bench.py --steps 200 --warmup 20 --workload water1m --cpu-steps 0 --no-scale-workload --decompose > gpurun_out/bench_r2b_w1m_dd1.json 2> gpurun_out/bench_r2b_w1m_dd1.err; echo "w1m dd1 exit $?"; tail -1 gpurun_out/bench_r2b_w1m_dd1.json | cut -c1-1200; tail -3 gpurun_out/bench_r2b_w1m_dd1.err
BENCH_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 2 --steps 50 --warmup 10 --workload water98k --transport gloo --cpu-steps 0 > gpurun_out/bench_r2b_2rank_gloo.json 2> gpurun_out/bench_r2b_2rank_gloo.err; echo "2rank exit $?"; tail -1 gpurun_out/bench_r2b_2rank_gloo.json | cut -c1-600; tail -3 gpurun_out/bench_r2b_2rank_gloo.err
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r2b_w1m -o trace -- python $R/bench.py --steps 100 --warmup 20 --workload water1m --cpu-steps 0 --no-roofline --no-scale-workload > $R/gpurun_out/prof_r2b_w1m.log 2>&1; echo "rocprof exit $?"
timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r2b_w1m_dd1 -o trace -- python $R/bench.py --steps 100 --warmup 20 --workload water1m --cpu-steps 0 --no-roofline --no-scale-workload --decompose > $R/gpurun_out/prof_r2b_w1m_dd1.log 2>&1; echo "rocprof dd1 exit $?"
cd $R
for t in w1m w1m_dd1; do f=$(find gpurun_out/prof_r2b_$t -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && { echo "== $t"; head -24 "$f" | cut -c1-160; }; done
