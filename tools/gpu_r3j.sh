#!/bin/bash
# round 2, session 2, visit j: decomposed run, 8 ranks serialised on one GPU: persistent pair grid (new default for long lists) against
# one wavefront per chunk (OPENMM_HIP_DIRECT_GRID=-1), interleaved on the same box -- the cost side of the setting (the benefit, the side
# stream's launches not queueing behind the pair kernel's dispatch backlog, cannot show with the ranks' streams serialised)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
port=29610
for rep in 1 2; do
  for g in auto -1; do
    ( [ $g != auto ] && export OPENMM_HIP_DIRECT_GRID=$g
      port=$((port + 1))
      timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port $((29610 + rep * 10 + ${#g})) bench.py --gpus 8 --steps 60 --warmup 5 --transport gloo --serialize-ranks --no-scale-workload --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('grid $g', d['per_rank_compute_ms_per_step']['ranks'])" )
  done
done 2>&1 | tee gpurun_out/r3j_ab_serialized_n8_grid.txt
