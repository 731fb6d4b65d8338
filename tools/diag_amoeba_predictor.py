"""Iterations of the AMOEBA dipole solver per time step, with and without the first guess from earlier steps
(OPENMM_HIP_AMOEBA_NO_PREDICTOR=1), on a small AMOEBA water box; BENCH_EMULATED=1 runs it on the CPU emulator.

    [BENCH_EMULATED=1] python tools/diag_amoeba_predictor.py [n_side=6] [steps=12] [epsilon=1e-5]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["OPENMM_HIP_AMOEBA_DEBUG"] = "1"          # the solver reports its iterations on stderr

from openmm_amd import harness as H, testsystems as T

n_side = int(sys.argv[1]) if len(sys.argv) > 1 else 6
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
eps = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-5
emulated = os.environ.get("BENCH_EMULATED") == "1"
H.load_amoeba_plugins(emulated=emulated)
w = T.amoeba_water_box(n_side, seed=3, polarization=H.Mutual, cutoff=0.7 if n_side >= 8 else 0.6, vdw_cutoff=0.9 if n_side >= 8 else 0.6, grid=(max(16, 4 * n_side),) * 3, a_ewald=5.4459052, epsilon=eps)
s, mp, vdw = w.build()
integ = H.Integrator(H.VERLET, 0.001)
ctx = H.Context(s, integ, "HIP")
ctx.setPositions(w.positions)
ctx.minimizeEnergy(50.0, 40)           # the generated box is a jittered lattice: relax it, or the first femtoseconds are an explosion
ctx.setVelocitiesToTemperature(300.0, 5)
integ.step(20)
e0 = ctx.getState(getEnergy=True)
print("start: E_pot %.4f E_kin %.4f" % (e0.potentialEnergy, e0.kineticEnergy), flush=True)
for k in range(steps):
    integ.step(1)
st = ctx.getState(getEnergy=True)
print("end:   E_pot %.4f E_kin %.4f  total drift %.5f" % (st.potentialEnergy, st.kineticEnergy, st.potentialEnergy + st.kineticEnergy - e0.potentialEnergy - e0.kineticEnergy), flush=True)
ctx.close()
