#!/bin/bash
# round 4, visit 9i: MP_SPLIT 4 (default) against 8 lanes per atom in the AMOEBA multipole list kernels, on DHFR (23 558 atoms) and the water tile
cd /root/repo
mkdir -p gpurun_out/r09i
cp openmm_amd/lib/libopenmm_hip_kernels.so /tmp/keep.so
run() { timeout 300 python tools/bench_amoeba.py $* --steps 40 --warm 10 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ns_per_day'], d['solver_iterations_per_solve'], d['E1'])"; }
{
for rep in 1 2; do for v in keep mp8; do
  if [ $v = keep ]; then cp /tmp/keep.so openmm_amd/lib/libopenmm_hip_kernels.so; else cp build/ab/$v.so openmm_amd/lib/libopenmm_hip_kernels.so; fi
  echo "== $v dhfr"; run --dhfr; echo "== $v water"; run
done; done
} 2>&1 | tee gpurun_out/r09i/amoeba_mp_split.txt
cp /tmp/keep.so openmm_amd/lib/libopenmm_hip_kernels.so
