#!/bin/bash
# round 5, visit m: resident list builder at 3 (as it was: 143 VGPRs), 4 (128) and 5 (94, 4 spilled) wavefronts per SIMD, same box
cd "$(dirname "$0")/.."
cp openmm_amd/lib/libopenmm_hip_kernels.so /tmp/keep.so
show() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d.get('roofline',{}); print('%-6s' % '$1', d['value'], d['ms_per_step'], 'rebuilds', r.get('rebuilds'), {k:(round(v['avg_us'],1) if v['avg_us'] else None) for k,v in r.get('kernel_timers_us',{}).items() if not k.startswith('pairs')})"; }
for wl in water1m apoa1; do for rep in 1 2; do for v in w3 w4 w5; do
  cp build/ab/$v.so openmm_amd/lib/libopenmm_hip_kernels.so
  steps=400; [ $wl = apoa1 ] && steps=1500
  timeout 600 python bench.py --workload $wl --steps $steps --warmup 100 --cpu-steps 0 --no-extra-workloads --no-scale-workload --props DisablePmeStream=true 2>/dev/null | tail -1 | show "$wl $v"
done; done; done 2>&1 | tee gpurun_out/r11m_ab_builder_occupancy.txt
cp build/ab/w5.so openmm_amd/lib/libopenmm_hip_kernels.so
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "cell_binned or direct_space_kernel" 2>&1 | tail -2
cp /tmp/keep.so openmm_amd/lib/libopenmm_hip_kernels.so
