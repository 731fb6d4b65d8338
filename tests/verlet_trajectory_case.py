"""Ten deterministic VerletIntegrator steps with every constraint algorithm in play, HIP platform against the Reference platform
(ReferenceVerletDynamics.cpp:76-119 around ReferenceConstraints.cpp:194-206).  Two Systems:
 * `zoo`   -- rigid waters (SETTLE) + X-H clusters (SHAKE) + an all-bonds stretch of the chain (CCMA): the staged kernels with the
              device-resident CCMA loop;
 * `chain` -- waters + X-H clusters only: the one-launch fused step (k_step_units) with SETTLE and SHAKE in registers.
Forces: PME + bonds + angles + torsions + 1-4s.  Shared by the CPU-emulator test and the GPU test."""
import re
import subprocess
import sys

from conftest import ROOT

CHILD = r'''
import os, sys, numpy as np
sys.path.insert(0, %r)
from openmm_amd import harness as H, testsystems as T
H.load_hip_platform(emulated=%r)
steps, dt, tol = %d, 0.001, 1e-8
for name in ("zoo", "chain"):
    w = T.constraint_zoo() if name == "zoo" else T.small_solvated_chain(seed=5)
    w.cutoff = 0.9
    w.pme_params = (float(np.sqrt(-np.log(2 * w.ewald_tol)) / w.cutoff), 28, 28, 28)
    w.cm_remover = False
    start = None
    res = {}
    for plat in ("Reference", "HIP"):
        s, nb = w.build()
        integ = H.Integrator(H.VERLET, dt, constraintTolerance=tol)
        c = H.Context(s, integ, plat)
        if start is None:
            c.setPositions(w.positions); c.applyConstraints(tol); c.setVelocitiesToTemperature(300.0, 4)
            st = c.getState(getPositions=True, getVelocities=True)
            start = (st.positions, st.velocities)
        c.setPositions(start[0]); c.setVelocities(start[1])
        integ.step(steps)
        res[plat] = c.getState(getPositions=True, getVelocities=True, getEnergy=True)
        if plat == "HIP":
            print(name, "MODE", c.getPlatformProperty("IntegrationMode"), "CONSTRAINTS", c.getPlatformProperty("ConstraintPartition"))
        c.close()
    r, h = res["Reference"], res["HIP"]
    moved = np.abs(r.positions - start[0]).max()
    p, d = w.constraints
    viol = np.abs(np.linalg.norm(h.positions[p[:, 0]] - h.positions[p[:, 1]], axis=1) / d - 1).max()
    print(name, "RESULT dpos %%.3e dvel %%.3e moved %%.3e ke_rel %%.3e constraints %%.3e time %%g %%g" %% (
        np.abs(r.positions - h.positions).max(), np.abs(r.velocities - h.velocities).max(), moved,
        abs(r.kineticEnergy - h.kineticEnergy) / r.kineticEnergy, viol, r.time, h.time))
'''


def run_verlet_trajectory_case(tmp_path, emulated, steps=10):
    """-> {system: dict(mode, partition, dpos, dvel, moved, ke_rel, constraints)}"""
    script = tmp_path / "verlet_trajectory_child.py"
    script.write_text(CHILD % (ROOT, emulated, steps))
    out = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=2400)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    result = {}
    for name in ("zoo", "chain"):
        m = re.search(name + r" RESULT dpos (\S+) dvel (\S+) moved (\S+) ke_rel (\S+) constraints (\S+) time (\S+) (\S+)", out.stdout)
        mode = re.search(name + r" MODE (.*) CONSTRAINTS (.*)", out.stdout)
        v = [float(x) for x in m.groups()]
        result[name] = {"mode": mode.group(1).strip(), "partition": mode.group(2).strip(), "dpos": v[0], "dvel": v[1], "moved": v[2], "ke_rel": v[3],
                        "constraints": v[4], "times": (v[5], v[6])}
    return result
