#!/bin/bash
# round 5, visit ad: AMOEBA -- the hook (vdW launch) right behind the multipole builder's kernels (default) or behind the list-free work as well
# (OPENMM_HIP_AMOEBA_HOOK_LAST=1), against the order before round 5 (OPENMM_HIP_AMOEBA_EARLY_FIRST=1)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
run() { echo -n "$1  "; env $2 timeout 300 python tools/bench_amoeba.py $3 --steps 40 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ns_per_day'])"; }
for rep in 1 2 3; do
  run "dhfr  early_first" OPENMM_HIP_AMOEBA_EARLY_FIRST=1 --dhfr
  run "dhfr  hook_last  " OPENMM_HIP_AMOEBA_HOOK_LAST=1 --dhfr
  run "dhfr  hook_first " X=1 --dhfr
done | tee gpurun_out/r11ad_amoeba_hook.txt
for rep in 1 2 3; do
  run "water early_first" OPENMM_HIP_AMOEBA_EARLY_FIRST=1 ""
  run "water hook_last  " OPENMM_HIP_AMOEBA_HOOK_LAST=1 ""
  run "water hook_first " X=1 ""
done | tee -a gpurun_out/r11ad_amoeba_hook.txt
