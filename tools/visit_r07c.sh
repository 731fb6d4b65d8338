cd /root/repo
export TAG=r07c
bash tools/gpu_visit.sh tests:"tests/test_gpu_multirank.py" 2>&1 | tail -6
grep -E "passed|failed|domain:|water-1M on" gpurun_out/r07c_pytest.log | head
bash tools/gpu_visit.sh serial:8 2>&1 | tail -4
OPENMM_HIP_DD_BOTH_SIDES=1 OPENMM_HIP_DD_DRIFT=0.75 OPENMM_HIP_REORDER_LAG=128 TAG=r07c_round3 bash tools/gpu_visit.sh serial:8 2>&1 | tail -4
for cfg in "-" "OPENMM_HIP_AMOEBA_NO_GATHER_COPY=1"; do
  ( if [ "$cfg" != "-" ]; then for kv in ${cfg//,/ }; do export "$kv"; done; fi
    echo "== $cfg"; timeout 200 python tools/bench_amoeba.py --steps 40 2>&1 | tail -1 | cut -c1-420 ) 2>&1 | tee -a gpurun_out/${TAG}_amoeba_ab.txt
done
