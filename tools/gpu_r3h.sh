#!/bin/bash
# round 2, session 2, visit h: two-stream mode (default above 60 000 atoms) with a persistent pair-kernel grid that leaves wave slots
# free for the reciprocal-space launches of the side stream (OPENMM_HIP_DIRECT_GRID = number of pair wavefronts; 2048 = 2 per SIMD)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
show() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-30s' % '$1', d['value'], d['ms_per_step'], {k:(round(v['avg_us'],1) if v['avg_us'] else None) for k,v in r['kernel_timers_us'].items()})"; }
for rep in 1 2; do
  for wl in apoa1 water1m; do
    steps=1000; [ $wl = water1m ] && steps=300
    for g in 0 2048 1792 1536 1280; do
      ( [ $g != 0 ] && export OPENMM_HIP_DIRECT_GRID=$g
        timeout 300 python bench.py --steps $steps --warmup 200 --cpu-steps 0 --no-scale-workload --workload $wl 2>/dev/null | show "$wl grid=$g" )
    done
  done
done 2>&1 | tee gpurun_out/r3h_ab_persistent_grid_two_streams.txt
