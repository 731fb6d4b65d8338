#!/bin/bash
# round 2, session 2, visit l (last): whole GPU suite on the final tree; 8 ranks serialised after the decomposed path stopped refreshing
# foreign double-precision positions nobody reads; default bench line
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
T=r3l
timeout 1200 python -m pytest tests -m gpu -q -x --timeout 900 > gpurun_out/pytest_$T.log 2>&1; echo "pytest exit $?"; grep -n "passed\|failed" gpurun_out/pytest_$T.log | tail -2
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
for rep in 1 2; do
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 2962$rep bench.py --gpus 8 --steps 60 --warmup 5 --transport gloo --serialize-ranks --no-scale-workload --no-roofline 2>/dev/null | tail -1 > gpurun_out/bench_${T}_serialized_n8_$rep.json
python -c "import sys,json; d=json.loads(open('gpurun_out/bench_${T}_serialized_n8_$rep.json').read()); print(d['per_rank_compute_ms_per_step']['ranks'])"
done
( time timeout 900 python bench.py > gpurun_out/bench_${T}_default.json 2> gpurun_out/bench_${T}_default.err ) 2>&1 | grep real; tail -1 gpurun_out/bench_${T}_default.json | cut -c1-250
