#!/bin/bash
# round 5, visit b: AMOEBA solver target / polish A/B (parity at the run epsilon + ms per step), the new bench line with in-run PMC
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
O=gpurun_out/r11b_amoeba_epsilon.txt
: > $O
for cfg in "OPENMM_HIP_AMOEBA_NO_POLISH=1" "X=1" "OPENMM_HIP_AMOEBA_EPSILON_SCALE=0.5" "OPENMM_HIP_AMOEBA_EPSILON_SCALE=0.3"; do
  echo "== $cfg" >> $O
  env $cfg timeout 300 python tools/diag_amoeba_run_epsilon.py 2>&1 | grep "max_rel" >> $O
  env $cfg timeout 300 python tools/bench_amoeba.py --dhfr --steps 20 2>/dev/null | tail -1 | cut -c1-400 >> $O
  env $cfg timeout 300 python tools/bench_amoeba.py --steps 20 2>/dev/null | tail -1 | cut -c1-400 >> $O
done
cat $O
( time timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r11b_bench_driver.json 2> gpurun_out/r11b_bench_driver.err ) 2>&1 | grep real
tail -1 gpurun_out/r11b_bench_driver.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(d['value'], d['ms_per_step'], {k:r.get(k) for k in ('frac','frac_counting_list_words','traffic','traffic_source','traffic_detail','algorithmic_bytes_per_launch')})
a=d['extra_workloads']['amoeba_dhfr']; print({k:a.get(k) for k in ('value','ms_per_step','force_parity','roofline')})
print(d['extra_workloads']['amoeba_water'].get('force_parity'))
"
