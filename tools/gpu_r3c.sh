#!/bin/bash
# round 2, session 2, visit c: i atoms of the pair loops from lanes (v_readlane) against scalar loads -- two builds of the kernel
# library, interleaved on the same box -- then, on the new build, the environment knobs (polynomial Ewald force, LJ-free tails,
# LDS-staged interpolation), DHFR and the 1M-atom box; kernel tests first
cd "$(dirname "$0")/.."
R=$(pwd); mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x --timeout 300 2>&1 | tail -2
show() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-44s' % '$1', d['value'], d['ms_per_step'], 'rows', r['rows'], {k:(round(v['avg_us'],1) if v['avg_us'] else None) for k,v in r['kernel_timers_us'].items()})"; }
cp openmm_amd/lib/libopenmm_hip_kernels.so /tmp/keep.so
for wl in dhfr water1m; do
  steps=3000; [ $wl = water1m ] && steps=300
  for rep in 1 2; do
    for v in old new; do
      cp build/ab/$v.so openmm_amd/lib/libopenmm_hip_kernels.so
      timeout 300 python bench.py --steps $steps --warmup 300 --cpu-steps 0 --no-scale-workload --workload $wl 2>/dev/null | show "$wl lib=$v"
    done
  done
done 2>&1 | tee gpurun_out/ab_r3c_lib.txt
cp build/ab/new.so openmm_amd/lib/libopenmm_hip_kernels.so
for wl in dhfr water1m; do
  steps=3000; [ $wl = water1m ] && steps=300
  for rep in 1 2; do
    for cfg in "-" "OPENMM_HIP_NO_EWALD_POLY=1" "OPENMM_HIP_NO_LJ_SPLIT=1" "OPENMM_HIP_INTERP_STAGED=1"; do
      ( [ "$cfg" != "-" ] && export "$cfg"
        timeout 300 python bench.py --steps $steps --warmup 300 --cpu-steps 0 --no-scale-workload --workload $wl 2>/dev/null | show "$wl $cfg" )
    done
  done
done 2>&1 | tee gpurun_out/ab_r3c_env.txt
cp /tmp/keep.so openmm_amd/lib/libopenmm_hip_kernels.so
