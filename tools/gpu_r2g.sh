#!/bin/bash
# round 2, visit g: FFT after the split-radix butterflies (56^3, 90^3, 98x98x70, 192^3), DHFR after the trig-free torsions
cd "$(dirname "$0")/.."
R=$(pwd); mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_platform.py -m gpu -q --timeout 600 -k "fft or real_dhfr or pme" > gpurun_out/pytest_r2g.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/pytest_r2g.log
for wl in dhfr water98k apoa1 water1m; do
  st=2000; [ $wl = water1m ] && st=300
  timeout 600 python bench.py --steps $st --warmup 100 --workload $wl --cpu-steps 0 --no-scale-workload > gpurun_out/bench_r2g_$wl.json 2> gpurun_out/bench_r2g_$wl.err; echo "$wl exit $?"
  python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/bench_r2g_$wl.json").read().splitlines() if l.startswith("{")][-1])
r=d["roofline"]; t=r["kernel_timers_us"]; f=d.get("roofline_fft",{})
print("$wl: %.1f ns/day %.4f ms/step | nb %.1f nl %.1f fft %s interp %.1f | fft roofline: %s us, %s GB/s, frac %s" % (d["value"], d["ms_per_step"], t["nb_direct"]["avg_us"] or 0, t["nl_update"]["avg_us"] or 0, t["pme_fft"]["avg_us"], t["pme_interpolate"]["avg_us"] or 0, f.get("avg_us"), f.get("achieved"), f.get("frac")))
PY
done
