#!/bin/bash
# round 2, visit m: the launcher/child flow of bench.py at N > 1 on a one-GPU box: (1) N = 2 asking for RCCL -- both RCCL
# configurations must fail (rank 1 has no GPU of its own) and the launchers fall back to host-staged gloo together;
# (2) N = 8 rehearsal over gloo; (3) default N = 1 line
cd "$(dirname "$0")/.."
R=$(pwd); mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
t0=$(date +%s)
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 2 --steps 20 --warmup 5 --attempt-timeout 150 > gpurun_out/bench_r2m_n2.json 2> gpurun_out/bench_r2m_n2.err; echo "N=2 exit $? after $(( $(date +%s) - t0 )) s"
grep "launcher\|rror" gpurun_out/bench_r2m_n2.err | head -8; tail -1 gpurun_out/bench_r2m_n2.json | cut -c1-900
t0=$(date +%s)
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29581 bench.py --gpus 8 --steps 20 --warmup 5 --transport gloo --attempt-timeout 200 > gpurun_out/bench_r2m_n8.json 2> gpurun_out/bench_r2m_n8.err; echo "N=8 exit $? after $(( $(date +%s) - t0 )) s"
grep "launcher\|rror" gpurun_out/bench_r2m_n8.err | head -8; tail -1 gpurun_out/bench_r2m_n8.json | cut -c1-300; tail -1 gpurun_out/bench_r2m_n8.json | grep -o '"single_gpu_same_box.*' | cut -c1-300
t0=$(date +%s)
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_r2m_n1.json 2> gpurun_out/bench_r2m_n1.err; echo "N=1 exit $? after $(( $(date +%s) - t0 )) s"
tail -1 gpurun_out/bench_r2m_n1.json | cut -c1-250
