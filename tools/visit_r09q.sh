#!/bin/bash
# round 4, visit 9q: grid clears of the AMOEBA solver inside stage 3 of k_mp_cg (default) against their own launch, same box, alternating
cd /root/repo; mkdir -p gpurun_out/r09q
for rep in 1 2 3; do
  for v in "A=0" "OPENMM_HIP_AMOEBA_CLEAR_LAUNCH=1"; do
    echo "== $v dhfr"; env $v python tools/bench_amoeba.py --dhfr 2>/dev/null | tail -1
    echo "== $v water"; env $v python tools/bench_amoeba.py 2>/dev/null | tail -1
  done
done 2>&1 | tee gpurun_out/r09q/clear_in_stage3.txt
python -m pytest tests/test_gpu_platform.py -q -x -k "amoeba or Amoeba" 2>&1 | tail -3 | tee -a gpurun_out/r09q/clear_in_stage3.txt
