#!/bin/bash
# round 4, visit 10a: the final tree -- GPU suite (with the triclinic decomposed case), the driver's bench line
cd /root/repo; mkdir -p gpurun_out/r10a
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 | tee gpurun_out/r10a/pytest_gpu.txt
timeout 600 python bench.py 2> gpurun_out/r10a/bench.err | tail -1 > gpurun_out/r10a/bench.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r10a/bench.json"))
print("value", d["value"], d["unit"], "ms/step", d["ms_per_step"], "roofline", d["roofline"].get("frac"))
for k, v in d.get("extra_workloads", {}).items():
    print(k, {q: v[q] for q in v if q in ("ms_per_step", "ns_per_day", "force_parity", "integration_mode")})
PY
