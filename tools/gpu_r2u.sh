#!/bin/bash
# round 2, visit u: each rank's step without communication at 2 / 4 / 8 ranks, ranks serialized on one GPU (bench.py --serialize-ranks)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
for n in 8 4 2; do
  t0=$(date +%s)
  timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2960$n bench.py --gpus $n --steps 60 --warmup 5 --transport gloo --serialize-ranks --no-scale-workload --no-roofline --attempt-timeout 400 > gpurun_out/bench_r2u_n$n.json 2> gpurun_out/bench_r2u_n$n.err; echo "N=$n exit $? after $(( $(date +%s) - t0 )) s"
  tail -1 gpurun_out/bench_r2u_n$n.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['per_rank_compute_ms_per_step']['ranks'], d['per_rank_compute_ms_per_step']['collectives_per_step'], d['ms_per_step'])"
done
