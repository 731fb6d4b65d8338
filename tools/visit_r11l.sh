#!/bin/bash
# round 5, visit l: k_mp_dipole_field with four list entries in flight per lane (new) against one (old), same box; parity of the new build
cd "$(dirname "$0")/.."
cp openmm_amd/lib/libopenmm_hip_kernels.so /tmp/keep.so
for rep in 1 2 3; do for v in old new; do
  cp build/ab/$v.so openmm_amd/lib/libopenmm_hip_kernels.so
  echo "$v dhfr  $(timeout 300 python tools/bench_amoeba.py --dhfr --steps 30 2>/dev/null | tail -1 | cut -c90-260)"
  echo "$v water $(timeout 300 python tools/bench_amoeba.py --steps 30 2>/dev/null | tail -1 | cut -c80-250)"
done; done 2>&1 | tee gpurun_out/r11l_ab_dipole_field_unroll.txt
cp /tmp/keep.so openmm_amd/lib/libopenmm_hip_kernels.so
python tools/diag_amoeba_run_epsilon.py 2>&1 | grep max_rel | tee -a gpurun_out/r11l_ab_dipole_field_unroll.txt
