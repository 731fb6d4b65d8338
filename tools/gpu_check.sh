#!/bin/bash
# First-contact script for the GPU box: smoke, the reference's own test bodies on the HIP platform, a short bench.
# Everything is logged under gpurun_out/ (merged back by gpurun).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out
(rocminfo | grep -E "Marketing Name|gfx9|Compute Unit" | head -8; nproc; lscpu | grep "Model name") > $OUT/device.txt 2>&1
echo "== smoke" ; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?" | tee -a $OUT/summary.txt
for t in ${TESTS:-HarmonicBondForce HarmonicAngleForce PeriodicTorsionForce CMMotionRemover Checkpoints NonbondedForce Ewald Settle VerletIntegrator LangevinIntegrator LangevinMiddleIntegrator CustomBondForce VirtualSites}; do
  s=$(date +%s.%N)
  timeout ${TEST_TIMEOUT:-400} build/tests/TestHip$t > $OUT/test_$t.log 2>&1
  rc=$?
  e=$(date +%s.%N)
  echo "TestHip$t exit $rc  $(tail -1 $OUT/test_$t.log)  $(echo "$e - $s" | bc 2>/dev/null || python3 -c "print($e-$s)") s" | tee -a $OUT/summary.txt
done
echo "== bench"; timeout 900 python bench.py --steps ${BENCH_STEPS:-1000} --warmup 100 > $OUT/bench.log 2> $OUT/bench.err; echo "bench exit $?" | tee -a $OUT/summary.txt
tail -1 $OUT/bench.log
