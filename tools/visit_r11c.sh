#!/bin/bash
# round 5, visit c: kernel trace of the energy query (DHFR) and of the 1M-atom step, as baselines for this round's kernel work
cd "$(dirname "$0")/.."
R=$(pwd)
mkdir -p gpurun_out
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_eq -o trace -- python $R/tools/diag_energy_query_trace.py 60 > $R/gpurun_out/r11c_energy_query.log 2>&1 )
tail -2 gpurun_out/r11c_energy_query.log
f=$(find gpurun_out/prof_eq -name "*.db" | head -1); [ -n "$f" ] && python tools/rocpd_kernel_stats.py "$f" > gpurun_out/r11c_energy_query_kernel_stats.txt 2>&1; rm -rf gpurun_out/prof_eq
head -25 gpurun_out/r11c_energy_query_kernel_stats.txt | cut -c1-200
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_1m -o trace -- python $R/bench.py --workload water1m --steps 300 --warmup 50 --cpu-steps 0 --no-roofline --no-scale-workload --no-extra-workloads --no-pmc > $R/gpurun_out/r11c_water1m.log 2>&1 )
tail -1 gpurun_out/r11c_water1m.log | cut -c1-300
f=$(find gpurun_out/prof_1m -name "*.db" | head -1); [ -n "$f" ] && python tools/rocpd_kernel_stats.py "$f" > gpurun_out/r11c_water1m_kernel_stats.txt 2>&1; rm -rf gpurun_out/prof_1m
head -25 gpurun_out/r11c_water1m_kernel_stats.txt | cut -c1-200
