#!/usr/bin/env python
"""Step the DHFR-like workload on HIP one step at a time and check the forces of every configuration against the
reference's CPU platform (oracle/_ref; diagnostics only).  Stops at the first configuration whose force error or
temperature looks wrong and saves it to gpurun_out/."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openmm_amd import harness as H, testsystems as T

H.load_hip_platform()
H.load_cpu_platform()
seed = int(os.environ.get("SEED", "1"))
total = int(os.environ.get("TOTAL", "6000"))
stride = int(os.environ.get("STRIDE", "1"))
w = T.dhfr_like(seed=1)
s, nb = w.build()
integ = H.Integrator(H.LANGEVIN_MIDDLE, 0.002, 300.0, 1.0, seed=seed, constraintTolerance=1e-5)
c = H.Context(s, integ, "HIP")
c.setPositions(w.positions)
c.applyConstraints(1e-5)
c.setVelocities(w.velocities)
s2, nb2 = w.build()
integ2 = H.Integrator(H.VERLET, 0.001)
cpu = H.Context(s2, integ2, "CPU")
pairs, dist = w.constraints
ndof = 3 * w.num_atoms - len(dist) - 3
bad = 0
for k in range(0, total, stride):
    integ.step(stride)
    st = c.getState(getPositions=True, getVelocities=True, getForces=True, getEnergy=True)
    cpu.setPositions(st.positions)
    sr = cpu.getState(getForces=True, getEnergy=True)
    rms = np.sqrt((sr.forces ** 2).sum(1).mean())
    diff = np.linalg.norm(st.forces - sr.forces, axis=1)
    err = diff.max() / rms
    temp = 2 * st.kineticEnergy / (ndof * 0.00831446261815324)
    vmax = np.abs(st.velocities).max()
    d = np.linalg.norm(st.positions[pairs[:, 0]] - st.positions[pairs[:, 1]], axis=1)
    cerr = np.abs(d - dist).max()
    flag = (not np.isfinite(err)) or err > 2e-3 or temp > 330 or cerr > 1e-4 or vmax > 12
    if flag or (k + stride) % 250 == 0:
        print("step %5d  T %.1f  PE %.1f (cpu %.1f)  force err %.3g  vmax %.2f  constraint err %.2e %s" % (
            k + stride, temp, st.potentialEnergy, sr.potentialEnergy, err, vmax, cerr, "  <-- FLAG" if flag else ""), flush=True)
    if flag:
        worst = np.argsort(-np.nan_to_num(diff, nan=1e30))[:10]
        for i in worst:
            print("     atom %6d q %+.3f |Fcpu| %.1f |dF| %.3f Fhip %s Fcpu %s |v| %.2f" % (i, w.charge[i], np.linalg.norm(sr.forces[i]), diff[i],
                  np.round(st.forces[i], 2), np.round(sr.forces[i], 2), np.linalg.norm(st.velocities[i])))
        np.savez_compressed(os.path.join(ROOT, "gpurun_out", "flag_seed%d_step%d.npz" % (seed, k + stride)), positions=st.positions,
                            velocities=st.velocities, forces_hip=st.forces, forces_cpu=sr.forces)
        bad += 1
        if bad >= 4:
            break
print("finished", "flags", bad)
