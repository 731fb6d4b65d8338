#!/bin/bash
# round 2, visit l: kernel traces of the 1M-atom box, single and decomposed over one rank (RCCL), to see what a rank's step is made of
cd "$(dirname "$0")/.."
R=$(pwd); mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r2l_w1m -o trace -- python $R/bench.py --steps 300 --warmup 20 --workload water1m --cpu-steps 0 --no-roofline --no-scale-workload > $R/gpurun_out/prof_r2l_w1m.log 2>&1; echo "rocprof single exit $?"
timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r2l_w1m_dd -o trace -- python $R/bench.py --steps 300 --warmup 20 --workload water1m --cpu-steps 0 --no-roofline --no-scale-workload --decompose > $R/gpurun_out/prof_r2l_w1m_dd.log 2>&1; echo "rocprof dd exit $?"
cd $R
python tools/rocpd_kernel_stats.py gpurun_out/prof_r2l_w1m/trace_results.db > gpurun_out/r02l_water1m_single_kernel_stats.txt 2>&1
python tools/rocpd_kernel_stats.py gpurun_out/prof_r2l_w1m_dd/trace_results.db > gpurun_out/r02l_water1m_dd1rank_kernel_stats.txt 2>&1
head -22 gpurun_out/r02l_water1m_single_kernel_stats.txt | cut -c40-150
head -26 gpurun_out/r02l_water1m_dd1rank_kernel_stats.txt | cut -c40-150
rm -rf gpurun_out/prof_r2l_w1m gpurun_out/prof_r2l_w1m_dd
