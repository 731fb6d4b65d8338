#!/bin/bash
# round 2, session 2, visit i: persistent pair-kernel grid beside the side stream is now the default for long lists -- GPU suite, the
# three sizes it touches, 8 ranks serialised (the decomposed path takes the same setting), default and driver lines
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
T=r3i
timeout 1200 python -m pytest tests -m gpu -q -x --timeout 900 > gpurun_out/pytest_$T.log 2>&1; echo "pytest exit $?"; grep -n "passed\|failed" gpurun_out/pytest_$T.log | tail -2
for wl in apoa1 water98k water1m; do
  steps=1000; [ $wl = water1m ] && steps=300
  python bench.py --steps $steps --warmup 100 --workload $wl --cpu-steps 0 --no-scale-workload 2>/dev/null | tail -1 > gpurun_out/bench_${T}_$wl.json; cut -c1-200 gpurun_out/bench_${T}_$wl.json
done
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29608 bench.py --gpus 8 --steps 60 --warmup 5 --transport gloo --serialize-ranks --no-scale-workload --no-roofline > gpurun_out/bench_${T}_serialized_n8.json 2> gpurun_out/bench_${T}_serialized_n8.err; echo "serialized N=8 exit $?"
tail -1 gpurun_out/bench_${T}_serialized_n8.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['per_rank_compute_ms_per_step']['ranks'], d['per_rank_compute_ms_per_step']['collectives_per_step'])"
( time timeout 900 python bench.py > gpurun_out/bench_${T}_default.json 2> gpurun_out/bench_${T}_default.err ) 2>&1 | grep real; tail -1 gpurun_out/bench_${T}_default.json | cut -c1-250
