#!/bin/bash
# round 4, visit 9g: interleaved pair-cache layout -- amoeba_dhfr and amoeba_water timings + kernel stats of the solver kernels
cd /root/repo
mkdir -p gpurun_out/r09g
run() { echo "== $*"; timeout 300 python tools/bench_amoeba.py $* --steps 40 --warm 10 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ns_per_day'], d['list_builds_per_step'], d['solver_iterations_per_solve'], d['E1'])"; }
{ run --dhfr; run --dhfr; run; run; } 2>&1 | tee gpurun_out/r09g/amoeba_cache_layout.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r09g/prof -o amoeba_dhfr -- python /root/repo/tools/bench_amoeba.py --dhfr --steps 30 > /dev/null 2>&1
cd /root/repo; python tools/rocpd_kernel_stats.py gpurun_out/r09g/prof/amoeba_dhfr_results.db > gpurun_out/r09g/amoeba_dhfr_kernel_stats.txt 2>&1; head -8 gpurun_out/r09g/amoeba_dhfr_kernel_stats.txt | cut -c1-160
rm -rf gpurun_out/r09g/prof
