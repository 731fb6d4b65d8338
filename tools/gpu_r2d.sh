#!/bin/bash
# round 2, visit d: whole GPU suite; water-1M (re-sort cost, kernel trace); apoa1 front-launch A/B; DHFR regression
cd "$(dirname "$0")/.."
R=$(pwd); mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > gpurun_out/pytest_r2d.log 2>&1; echo "pytest exit $?"; tail -6 gpurun_out/pytest_r2d.log
timeout 600 python bench.py --steps 3000 --warmup 300 --no-scale-workload > gpurun_out/bench_r2d.json 2> gpurun_out/bench_r2d.err; echo "bench exit $?"; tail -1 gpurun_out/bench_r2d.json | cut -c1-330
OPENMM_HIP_TIMING=1 timeout 600 python bench.py --steps 1100 --warmup 20 --workload water1m --cpu-steps 0 --no-scale-workload > gpurun_out/bench_r2d_w1m.json 2> gpurun_out/bench_r2d_w1m.err; echo "w1m exit $?"; tail -1 gpurun_out/bench_r2d_w1m.json | cut -c1-330; grep "re-sort" gpurun_out/bench_r2d_w1m.err
for fm in 60000 200000; do
  OPENMM_HIP_FUSED_FRONT_MAX_ATOMS=$fm timeout 300 python bench.py --steps 1000 --warmup 100 --workload apoa1 --cpu-steps 0 --no-scale-workload > gpurun_out/bench_r2d_apoa1_$fm.json 2> gpurun_out/bench_r2d_apoa1_$fm.err; echo "apoa1 front<=$fm exit $?"; tail -1 gpurun_out/bench_r2d_apoa1_$fm.json | cut -c1-260
done
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r2d_w1m -o trace -- python $R/bench.py --steps 200 --warmup 20 --workload water1m --cpu-steps 0 --no-roofline --no-scale-workload > $R/gpurun_out/prof_r2d_w1m.log 2>&1; echo "rocprof exit $?"
cd $R
python tools/rocpd_kernel_stats.py gpurun_out/prof_r2d_w1m/trace_results.db 2>&1 | head -16 | cut -c1-150
