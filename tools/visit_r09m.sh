#!/bin/bash
# round 4, visit 9m: what an energy query (getState(getEnergy)) of the DHFR Context consists of -- wall time and the kernels / copies of 20 of them
cd /root/repo; mkdir -p gpurun_out/r09m
python tools/diag_energy_query_cost.py 2>/dev/null | tee gpurun_out/r09m/energy_query.txt
cat > /tmp/eq.py <<'PY'
import sys
sys.path.insert(0, '/root/repo')
from openmm_amd import harness as H, testsystems as T
H.load_hip_platform()
w = T.dhfr(); s, nb = w.build()
integ = H.Integrator(H.LANGEVIN_MIDDLE, 0.002, 300.0, 1.0, seed=1)
ctx = H.Context(s, integ, "HIP"); ctx.setPositions(w.positions); ctx.setVelocitiesToTemperature(300.0, 1)
integ.step(50); ctx.getState(getEnergy=True)
for k in range(200): ctx.getState(getEnergy=True)
PY
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --stats -d /root/repo/gpurun_out/r09m/prof -o eq -- python /tmp/eq.py > /dev/null 2>&1
cd /root/repo; python tools/rocpd_kernel_stats.py gpurun_out/r09m/prof/eq_results.db 2>&1 | head -24 | cut -c1-150 | tee -a gpurun_out/r09m/energy_query.txt
rm -rf gpurun_out/r09m/prof
