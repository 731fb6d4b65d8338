#!/bin/bash
# A/B two complete library sets (build/ab/setA vs build/ab/setB: kernels + plugin + harness) on the same box, interleaved
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p /tmp/keep && cp openmm_amd/lib/*.so /tmp/keep/
for rep in 1 2 3; do
  for v in setA setB; do
    cp build/ab/$v/*.so openmm_amd/lib/
    echo "$v $(python bench.py --steps 3000 --warmup 300 --cpu-steps 0 $BENCH_ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], round(r['kernel_timers_us']['nb_direct']['avg_us'],1), 'rows', r['rows'], 'chunks', r['chunks'])")"
  done
done
cp /tmp/keep/*.so openmm_amd/lib/
