#!/bin/bash
# round 5, visit u: the AMOEBA list builder with block bounding boxes + prefetched staging (DHFR step time, kernel stats), the AMOEBA GPU
# tests, and the 1M-atom Reference golden on 4 and 8 ranks sharing the GPU
cd "$(dirname "$0")/.."
R=$(pwd)
mkdir -p gpurun_out
for rep in 1 2 3; do timeout 300 python tools/bench_amoeba.py --dhfr --steps 40 2>&1 | tail -1 | cut -c1-420; done | tee gpurun_out/r11u_amoeba_dhfr.txt
OPENMM_HIP_PL_DEBUG=4 timeout 300 python tools/bench_amoeba.py --dhfr --steps 6 2>&1 | grep "pl_build trace" | tail -4 | cut -c1-300 | tee -a gpurun_out/r11u_amoeba_dhfr.txt
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_am -o trace -- python $R/tools/bench_amoeba.py --dhfr --steps 40 > $R/gpurun_out/r11u_amoeba_traced.log 2>&1 )
f=$(find gpurun_out/prof_am -name "*.db" | head -1); [ -n "$f" ] && python tools/rocpd_kernel_stats.py "$f" > gpurun_out/r11u_amoeba_dhfr_kernel_stats.txt 2>&1; rm -rf gpurun_out/prof_am
head -16 gpurun_out/r11u_amoeba_dhfr_kernel_stats.txt | cut -c1-180
timeout 1500 python -m pytest tests/test_gpu_platform.py -q -x -k "amoeba" 2>&1 | tail -3 | tee gpurun_out/r11u_pytest.txt
timeout 1500 python -m pytest tests/test_gpu_multirank.py -q -x -k "golden" -s 2>&1 | grep -a "water-1M\|domain\|passed\|failed" | tee -a gpurun_out/r11u_pytest.txt
