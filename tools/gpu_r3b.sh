#!/bin/bash
# round 2, session 2, visit b: what bounds the pair kernel?  SQ counter passes on the 1M-atom box (pair kernel as a launch of its own),
# new loops (polynomial Ewald force + LJ-free tails) against the old ones
cd "$(dirname "$0")/.."
R=$(pwd); mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
run() {  # tag, counters
  tag=$1; shift
  ( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$tag -o pmc -- python $R/bench.py --steps 40 --warmup 10 --workload water1m --prepare-steps 50 --cpu-steps 0 --no-roofline --no-scale-workload > $R/gpurun_out/pmc_$tag.log 2>&1; echo "rocprof $tag exit $?" )
  f=$(find gpurun_out/pmc_$tag -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python tools/pmc_summary.py "$f" nl_find=40 > gpurun_out/r3b_pmc_${tag}.txt 2>&1 && grep -v "^Scratch\|^LDS_Block\|^Accum" gpurun_out/r3b_pmc_${tag}.txt | cut -c1-200
  rm -rf gpurun_out/pmc_$tag
}
run new1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY
( export OPENMM_HIP_NO_EWALD_POLY=1 OPENMM_HIP_NO_LJ_SPLIT=1; run old1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY )
run new2 SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_IFETCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS
