#!/bin/bash
# round 2, session 2, visit f: reciprocal space on its own stream (DisablePmeStream=false) against the single-stream default at the
# sizes where the pair kernel is a launch of its own (apoa1, 98k, 1M), interleaved on one box
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
show() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-44s' % '$1', d['value'], d['ms_per_step'], {k:(round(v['avg_us'],1) if v['avg_us'] else None) for k,v in r['kernel_timers_us'].items()})"; }
for rep in 1 2; do
  for wl in apoa1 water98k water1m; do
    steps=1000; [ $wl = water1m ] && steps=300
    for props in "" "DisablePmeStream=false"; do
      timeout 300 python bench.py --steps $steps --warmup 200 --cpu-steps 0 --no-scale-workload --workload $wl ${props:+--props $props} 2>/dev/null | show "$wl ${props:-single-stream}"
    done
  done
done 2>&1 | tee gpurun_out/r3f_ab_pme_stream.txt
