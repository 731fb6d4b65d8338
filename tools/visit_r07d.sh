cd /root/repo
export OPENMM_HIP_TIMING=1
OPENMM_HIP_DD_DRIFT=0.75 TAG=r07d_hs075 bash tools/gpu_visit.sh serial:8 2>&1 | tail -2
grep -h "re-sort" gpurun_out/r07d_hs075_serialized_n8.err | sort | uniq -c | sort -rn | head -8
OPENMM_HIP_DD_DRIFT=0.4 TAG=r07d_hs04 bash tools/gpu_visit.sh serial:8 2>&1 | tail -2
grep -h "re-sort" gpurun_out/r07d_hs04_serialized_n8.err | sort | uniq -c | sort -rn | head -12
OPENMM_HIP_DD_BOTH_SIDES=1 OPENMM_HIP_DD_DRIFT=0.4 TAG=r07d_bs04 bash tools/gpu_visit.sh serial:8 2>&1 | tail -2
