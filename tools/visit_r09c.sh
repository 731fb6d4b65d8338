#!/bin/bash
# round 4, visit 9c: per-call durations of the list-builder launches at 1M atoms (are the no-op calls cheap?)
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_r09c -o trace -- python /root/repo/bench.py --cpu-steps 0 --no-extra-workloads --workload water1m --steps 300 --warmup 50 > /root/repo/gpurun_out/r09c.log 2>&1
cd /root/repo
python tools/rocpd_kernel_stats.py gpurun_out/prof_r09c/trace_results.db nl_ > gpurun_out/r09c_water1m_kernel_stats.txt 2>&1
grep -E "deciles|nl_|ctor_type" gpurun_out/r09c_water1m_kernel_stats.txt | cut -c1-250
rm -rf gpurun_out/prof_r09c
