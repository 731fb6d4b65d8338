#!/bin/bash
# round 2, visit j: rehearsal of the driver's N = 4 and N = 8 command lines on ONE GPU (8 processes sharing the device, collectives
# host-staged over gloo because RCCL refuses several ranks on one device): the 8-rank code paths on real kernels
cd "$(dirname "$0")/.."
R=$(pwd); mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
for n in 4 8; do
  BENCH_DIST_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2958$n bench.py --gpus $n --steps 20 --warmup 5 --transport gloo > gpurun_out/bench_r2j_n$n.json 2> gpurun_out/bench_r2j_n$n.err; echo "N=$n exit $?"
  grep "^{" gpurun_out/bench_r2j_n$n.json | cut -c1-700; grep -i "error\|Traceback\|exception" gpurun_out/bench_r2j_n$n.err | head -5
done
