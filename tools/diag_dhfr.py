#!/usr/bin/env python
"""Diagnostics on the DHFR-like workload (run on the GPU box): (1) where the HIP-vs-Reference force error sits,
per force term and per atom; (2) a long LangevinMiddle run printed in chunks (temperature, energies)."""
import copy
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openmm_amd import harness as H, testsystems as T

H.load_hip_platform()
what = sys.argv[1] if len(sys.argv) > 1 else "forces"
w = T.dhfr_like(seed=1)


def forces(w, plat, props=None):
    s, nb = w.build()
    integ = H.Integrator(H.VERLET, 0.001)
    c = H.Context(s, integ, plat, props)
    c.setPositions(w.positions)
    st = c.getState(getForces=True, getEnergy=True)
    c.close()
    return st


def report(tag, fr, fh):
    rms = np.sqrt((fr ** 2).sum(1).mean())
    diff = np.linalg.norm(fh - fr, axis=1)
    worst = np.argsort(-diff)[:8]
    print("%-28s rms %.1f  max|dF|/rms %.3g  p99.9 %.3g  p99 %.3g  median %.3g" % (
        tag, rms, diff.max() / rms, np.percentile(diff, 99.9) / rms, np.percentile(diff, 99) / rms, np.median(diff) / rms))
    for i in worst:
        print("     atom %6d  q %+.3f  |Fref| %9.2f  |dF| %.4f  dF %s" % (i, w.charge[i], np.linalg.norm(fr[i]), diff[i], np.round(fh[i] - fr[i], 4)))


if what == "forces":
    variants = {}
    full = w
    variants["full"] = full
    nb_only = copy.copy(w); nb_only.bonds = nb_only.angles = nb_only.torsions = None
    variants["nonbonded only (PME)"] = nb_only
    rf = copy.copy(nb_only); rf.method = H.CutoffPeriodic
    variants["nonbonded only (RF cutoff)"] = rf
    noq = copy.copy(nb_only); noq.charge = np.zeros_like(w.charge); noq.exceptions = None
    variants["LJ only (PME method)"] = noq
    nolj = copy.copy(nb_only); nolj.epsilon = np.zeros_like(w.epsilon)
    variants["Coulomb only (PME)"] = nolj
    bonded = copy.copy(w); bonded.charge = np.zeros_like(w.charge); bonded.epsilon = np.zeros_like(w.epsilon); bonded.method = H.CutoffPeriodic
    variants["bonded only"] = bonded
    for tag, v in variants.items():
        try:
            r, h = forces(v, "Reference"), forces(v, "HIP")
            report(tag, r.forces, h.forces)
            print("     E ref %.4f hip %.4f" % (r.potentialEnergy, h.potentialEnergy))
        except Exception as e:
            print(tag, "failed:", e)
else:
    s, nb = w.build()
    integ = H.Integrator(H.LANGEVIN_MIDDLE, 0.002, 300.0, 1.0, seed=int(os.environ.get("SEED", "1")), constraintTolerance=1e-5)
    c = H.Context(s, integ, "HIP")
    c.setPositions(w.positions)
    c.applyConstraints(1e-5)
    c.setVelocities(w.velocities) if w.velocities is not None else c.setVelocitiesToTemperature(300.0, 1)
    pairs, dist = w.constraints
    ndof = 3 * w.num_atoms - len(dist) - 3
    chunk = int(os.environ.get("CHUNK", "100"))
    total = int(os.environ.get("TOTAL", "5000"))
    last_good = None
    for k in range(total // chunk):
        integ.step(chunk)
        st = c.getState(getEnergy=True, getPositions=True, getVelocities=True)
        temp = 2 * st.kineticEnergy / (ndof * 0.00831446261815324)
        d = np.linalg.norm(st.positions[pairs[:, 0]] - st.positions[pairs[:, 1]], axis=1)
        if (k + 1) % max(1, 1000 // chunk) == 0 or not np.isfinite(st.potentialEnergy):
            print("step %5d  T %.1f  PE %.1f  KE %.1f  max constraint err %.2e" % ((k + 1) * chunk, temp, st.potentialEnergy, st.kineticEnergy, np.abs(d - dist).max()), flush=True)
        if not np.isfinite(st.potentialEnergy) or temp > 330:
            print("NAN at", (k + 1) * chunk)
            if last_good is not None:
                np.savez_compressed(os.path.join(ROOT, "gpurun_out", "lastgood_%s_seed%s_step%d.npz" % (os.environ.get("TAG", "run"), os.environ.get("SEED", "1"), k * chunk)),
                                    positions=last_good.positions, velocities=last_good.velocities)
            break
        last_good = st
