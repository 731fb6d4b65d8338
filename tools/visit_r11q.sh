#!/bin/bash
# round 5, visit q: pair kernel with two half passes over the i atoms (48 accumulators: 189 VGPRs; asked for three wavefronts per SIMD: 168 with 61 spilled)
# against the kernel as it is (254 VGPRs), same box, one stream; parity of the half-pass kernel through the kernel tests
cd "$(dirname "$0")/.."
cp openmm_amd/lib/libopenmm_hip_kernels.so /tmp/keep.so
show() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d.get('roofline',{}); print('%-22s' % '$1', d['value'], d['ms_per_step'], {k:(round(v['avg_us'],1) if v['avg_us'] else None) for k,v in r.get('kernel_timers_us',{}).items() if k in ('nb_direct',)})"; }
for wl in water1m apoa1; do for rep in 1 2; do for v in "h2 0" "h2 1" "h3 1"; do
  set -- $v
  cp build/ab/$1.so openmm_amd/lib/libopenmm_hip_kernels.so
  steps=400; [ $wl = apoa1 ] && steps=1500
  OPENMM_HIP_PAIR_HALVES=$2 timeout 600 python bench.py --workload $wl --steps $steps --warmup 100 --cpu-steps 0 --no-extra-workloads --no-scale-workload --no-pmc --props DisablePmeStream=true 2>/dev/null | tail -1 | show "$wl $1 halves=$2"
done; done; done 2>&1 | tee gpurun_out/r11q_ab_pair_half_passes.txt
cp /tmp/keep.so openmm_amd/lib/libopenmm_hip_kernels.so
OPENMM_HIP_PAIR_HALVES=1 timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "direct_space or cutoff_edge" 2>&1 | tail -2
