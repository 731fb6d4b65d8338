cd /root/repo
OPENMM_HIP_DD_DRIFT=0.75 OPENMM_HIP_REORDER_LAG=128 TAG=r07e_hs075 bash tools/gpu_visit.sh serial:8 2>&1 | tail -1
OPENMM_HIP_DD_BOTH_SIDES=1 OPENMM_HIP_DD_DRIFT=0.75 OPENMM_HIP_REORDER_LAG=128 TAG=r07e_bs075 bash tools/gpu_visit.sh serial:8 2>&1 | tail -1
OPENMM_HIP_DD_DRIFT=0.75 OPENMM_HIP_REORDER_LAG=128 TAG=r07e_hs075 bash tools/gpu_visit.sh serialtrace:8 2>&1 | tail -24
OPENMM_HIP_DD_BOTH_SIDES=1 OPENMM_HIP_DD_DRIFT=0.75 OPENMM_HIP_REORDER_LAG=128 TAG=r07e_bs075 bash tools/gpu_visit.sh serialtrace:8 2>&1 | tail -24
