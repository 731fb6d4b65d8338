#!/bin/bash
# round 4, visit 8d: interpreter with one thread per DOF (kernel stats), preconditioner A/B on amoeba_dhfr and amoeba_water, the reference's CustomIntegrator body
cd /root/repo
mkdir -p gpurun_out/r08d
run() { echo "== $*"; env "$@" timeout 300 python tools/bench_amoeba.py $ARGS --steps 40 --warm 10 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ns_per_day'], d['list_builds_per_step'], d['solver_iterations_per_solve'], d['E1'])"; }
{
ARGS=--dhfr
run A=0
run OPENMM_HIP_AMOEBA_PRECOND=1
run A=0
run OPENMM_HIP_AMOEBA_PRECOND=1
ARGS=
run A=0
run OPENMM_HIP_AMOEBA_PRECOND=1
} 2>&1 | tee gpurun_out/r08d/amoeba_precond_ab.txt
timeout 900 build/tests/TestHipCustomIntegrator > gpurun_out/r08d/TestHipCustomIntegrator.txt 2>&1; echo "exit $?" >> gpurun_out/r08d/TestHipCustomIntegrator.txt; tail -2 gpurun_out/r08d/TestHipCustomIntegrator.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r08d/prof -o amoeba_dhfr -- python /root/repo/tools/bench_amoeba.py --dhfr --steps 30 > /root/repo/gpurun_out/r08d/bench_amoeba_dhfr.txt 2>&1
cd /root/repo; python tools/rocpd_kernel_stats.py gpurun_out/r08d/prof/amoeba_dhfr_results.db > gpurun_out/r08d/amoeba_dhfr_kernel_stats.txt 2>&1; grep -E "k_vm|k_valence|sum of kernel" gpurun_out/r08d/amoeba_dhfr_kernel_stats.txt | cut -c1-160
rm -rf gpurun_out/r08d/prof
