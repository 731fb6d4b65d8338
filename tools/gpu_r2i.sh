#!/bin/bash
# round 2, visit i: bonded-term changes on the real DHFR System (unfused kernel trace + default bench + torsion/parity tests)
cd "$(dirname "$0")/.."
R=$(pwd); mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_gpu_platform.py -m gpu -q --timeout 600 -k "real_dhfr" > gpurun_out/pytest_r2i.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/pytest_r2i.log
for i in 1 2; do OPENMM_HIP_TRANSPOSE_TERMS=$((i-1)) timeout 600 python bench.py --steps 3000 --warmup 300 --cpu-steps 0 --no-scale-workload > gpurun_out/bench_r2i_$i.json 2> gpurun_out/bench_r2i_$i.err; grep "^{" gpurun_out/bench_r2i_$i.json | cut -c100-190; done
cd /tmp && export TMPDIR=/tmp
OPENMM_HIP_NO_FUSED_FRONT=1 timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r2i -o trace -- python $R/bench.py --steps 1500 --warmup 100 --cpu-steps 0 --no-roofline --no-scale-workload > $R/gpurun_out/prof_r2i.log 2>&1; echo "rocprof exit $?"
python $R/tools/rocpd_kernel_stats.py $R/gpurun_out/prof_r2i/trace_results.db 2>&1 | grep "k_terms\|k_step\|interpolate\|spread"
