#!/bin/bash
# round 5, visit av: 1M atoms, two streams -- the large fused plane transform (one workgroup takes a whole CU's LDS: it starves beside the pair kernel,
# 522 us per launch against 48 alone) against the line passes (OPENMM_HIP_NO_BIG_PLANE=1), which fit beside it
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
show() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d.get('roofline',{}); print('%-34s' % '$1', d['value'], d['ms_per_step'], {k:(round(v['avg_us'],1) if v['avg_us'] else None) for k,v in r.get('kernel_timers_us',{}).items() if not k.startswith('pairs')})"; }
for rep in 1 2; do for v in 0 1; do
  if [ $v = 1 ]; then export OPENMM_HIP_NO_BIG_PLANE=1; else unset OPENMM_HIP_NO_BIG_PLANE; fi
  timeout 600 python bench.py --workload water1m --steps 400 --warmup 100 --cpu-steps 0 --no-extra-workloads --no-scale-workload --no-pmc 2>&1 | tail -1 | show "water1m no_big_plane=$v"
done; done 2>&1 | tee gpurun_out/r11av_ab_big_plane_two_streams.txt
