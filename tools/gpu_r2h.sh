#!/bin/bash
# round 2, visit h: what makes the front launch of the real DHFR System slower than the stand-in's (separate launches, kernel trace)
cd "$(dirname "$0")/.."
R=$(pwd); mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd /tmp && export TMPDIR=/tmp
for wl in dhfr dhfr_like; do
OPENMM_HIP_NO_FUSED_FRONT=1 timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r2h_$wl -o trace -- python $R/bench.py --steps 1500 --warmup 100 --cpu-steps 0 --no-roofline --no-scale-workload --workload $wl > $R/gpurun_out/prof_r2h_$wl.log 2>&1; echo "rocprof $wl exit $?"
python $R/tools/rocpd_kernel_stats.py $R/gpurun_out/prof_r2h_$wl/trace_results.db 2>&1 | head -12 | cut -c1-150
done
