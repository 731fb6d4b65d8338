"""CM-motion removal on a box large enough for the last block of the fused integration kernel to walk through several batches of
block partials (273 k atoms = 712 blocks): after a few steps the total momentum must be that of the remover (zero up to rounding),
and a wrong or stale partial would show as a momentum of the size of a block's.  usage: check_cm_large.py [n_side]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openmm_amd import harness as H, testsystems as T

H.load_hip_platform(emulated=os.environ.get("BENCH_EMULATED") == "1")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 45
w = T.water_box(n, seed=2)
w.cm_remover = True
system, nb = w.build()
integ = H.Integrator(H.VERLET, 0.002)          # no thermostat: what the remover leaves is what the forces (net force ~ 0) add in one step
ctx = H.Context(system, integ, "HIP")
ctx.setPositions(w.positions)
ctx.applyConstraints(1e-6)
ctx.setVelocitiesToTemperature(300.0, 5)
integ.step(12)
st = ctx.getState(getVelocities=True, getEnergy=True)
m = np.tile(np.array([15.99943, 1.007947, 1.007947]), w.num_atoms // 3)
p = (m[:, None] * st.velocities).sum(0)
scale = np.sqrt((m[:, None] ** 2 * st.velocities ** 2).sum())       # size of a random sum of the same terms
print("atoms", w.num_atoms, "blocks", (w.num_atoms // 3 + 127) // 128, "total momentum", p, "relative to a random sum", np.abs(p).max() / scale)
assert np.isfinite(st.potentialEnergy) and np.abs(p).max() / scale < 1e-3
print("OK")
