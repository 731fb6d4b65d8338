#!/bin/bash
# round 4, visit 8b: native AMOEBA valence kernels + CustomIntegrator on the device: new GPU tests, the amoeba_dhfr bench entry, kernel stats of an amoeba_dhfr run
cd /root/repo
mkdir -p gpurun_out/r08b
timeout 1500 python -m pytest tests/test_gpu_platform.py -m gpu -x -q -k "amoeba2009 or CustomIntegrator or CustomAngle or CustomCompound or AmoebaTorsionTorsion or Amoeba" > gpurun_out/r08b/pytest_gpu_subset.txt 2>&1
echo "pytest exit $?" >> gpurun_out/r08b/pytest_gpu_subset.txt
tail -15 gpurun_out/r08b/pytest_gpu_subset.txt
timeout 900 python bench.py > gpurun_out/r08b/bench_driver.json 2> gpurun_out/r08b/bench_driver.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r08b/bench_driver.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'])
for k, v in d['extra_workloads'].items(): print(k, json.dumps(v)[:1200])
PY
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r08b/prof -o amoeba_dhfr -- python /root/repo/tools/bench_amoeba.py --dhfr --steps 30 > /root/repo/gpurun_out/r08b/bench_amoeba_dhfr.txt 2>&1
tail -5 /root/repo/gpurun_out/r08b/bench_amoeba_dhfr.txt
find /root/repo/gpurun_out/r08b/prof -name "*kernel_stats.csv" | head -2 | while read f; do head -30 "$f" | cut -c1-200; done
