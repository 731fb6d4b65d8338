"""Force error of the AMOEBA path at the benchmark's mutualInducedTargetEpsilon (1e-5 D) against the Reference-platform goldens solved to
1e-6 D: amoeba2009 DHFR (all 23 558 atoms, multipoles + vdW) and the 36 501-atom water tile (12 000 sampled atoms).  One line per system;
environment knobs (OPENMM_HIP_AMOEBA_EPSILON_SCALE, OPENMM_HIP_AMOEBA_NO_POLISH) are read by the plugin.  GPU box.

    python tools/diag_amoeba_run_epsilon.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openmm_amd import harness as H, testsystems as T

H.load_amoeba_plugins()
tag = "scale=%s polish=%s" % (os.environ.get("OPENMM_HIP_AMOEBA_EPSILON_SCALE", "default"), "off" if os.environ.get("OPENMM_HIP_AMOEBA_NO_POLISH") else "on")
g = np.load(os.path.join(ROOT, "tests", "golden", "reference_forces_amoeba_dhfr.npz"))
ref = g["forces_vdw"].astype(np.float64) + g["forces_multipole"].astype(np.float64)
w = T.amoeba_dhfr(epsilon=1e-5, pin_grid=True)
s, mp, vdw = w.build()
ctx = H.Context(s, H.MTSLangevinIntegrator(300.0, 1.0, 0.002, [(0, 2), (1, 1)], seed=7), "HIP")
ctx.setPositions(w.positions)
s0 = H.amoeba_solver_iterations()
f = ctx.getState(getForces=True, groups=2).forces
s1 = H.amoeba_solver_iterations()
rms = float(np.sqrt((ref ** 2).sum(1).mean()))
rel = np.linalg.norm(f - ref, axis=1) / np.maximum(np.linalg.norm(ref, axis=1), rms)
print("%s  amoeba_dhfr   max_rel_err_at_the_run_epsilon %.3e  99.9%% %.3e  above 5e-5: %d  iterations %d" % (tag, rel.max(), np.percentile(rel, 99.9), int((rel > 5e-5).sum()), s1[1] - s0[1]), flush=True)
ctx.close()
g = np.load(os.path.join(ROOT, "tests", "golden", "reference_forces_amoeba_water_tile_36501_mutual_sample.npz"))
aw = T.amoeba_water_tile(cutoff=0.7, vdw_cutoff=0.9, polarization=H.Mutual, epsilon=1e-5, ewald_tol=7.5e-4, grid=(80, 80, 80), a_ewald=float(np.sqrt(-np.log(2 * 7.5e-4)) / 0.7))
s, mp, vdw = aw.build()
ctx = H.Context(s, H.Integrator(H.VERLET, 0.001), "HIP")
ctx.setPositions(aw.positions)
s0 = H.amoeba_solver_iterations()
f = ctx.getState(getForces=True).forces
s1 = H.amoeba_solver_iterations()
rel = np.linalg.norm(f[g["indices"]] - g["forces"], axis=1) / np.maximum(np.linalg.norm(g["forces"], axis=1), float(g["rms_force"]))
print("%s  amoeba_water  max_rel_err_at_the_run_epsilon %.3e  99.9%% %.3e  above 5e-5: %d  iterations %d" % (tag, rel.max(), np.percentile(rel, 99.9), int((rel > 5e-5).sum()), s1[1] - s0[1]), flush=True)
ctx.close()
