#!/bin/bash
# round 4, visit 8a: full GPU suite on the current tree + the driver's bench with the new amoeba_dhfr workload (host mode: CustomIntegrator + fallback valence forces)
cd /root/repo
mkdir -p gpurun_out/r08a
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r08a/pytest_gpu.txt 2>&1
echo "pytest exit $?" >> gpurun_out/r08a/pytest_gpu.txt
tail -5 gpurun_out/r08a/pytest_gpu.txt
timeout 900 python bench.py > gpurun_out/r08a/bench_driver.json 2> gpurun_out/r08a/bench_driver.err
tail -c 6000 gpurun_out/r08a/bench_driver.json
