"""A velocity-Verlet CustomIntegrator with SETTLE constraints (the form of tests/TestCustomIntegrator.h testConstraints) on a rigid TIP3P box
with PME: the HIP platform's device interpreter against the Reference platform, and -- the neighbour list shrunk so
that a device-triggered rebuild overflows in the middle of the run -- against its own undisturbed run (the custom integrator looks for an
overflow synchronously after every force evaluation and evaluates again).  Shared by the CPU-emulator test and the GPU test."""
import subprocess
import sys

from conftest import ROOT

CHILD = r'''
import os, sys, numpy as np
sys.path.insert(0, %r)
from openmm_amd import harness as H, testsystems as T
H.load_hip_platform(emulated=%r)
steps = %d


def run(plat, shrink):
    if shrink: os.environ["OPENMM_HIP_DEBUG_SHRINK_LIST_AFTER"] = "3"
    else: os.environ.pop("OPENMM_HIP_DEBUG_SHRINK_LIST_AFTER", None)
    w = T.water_box(%d, seed=9, cutoff=%r)
    w.pme_params = (float(np.sqrt(-np.log(2 * w.ewald_tol)) / w.cutoff), %d, %d, %d)
    w.cm_remover = True
    s, nb = w.build()
    integ = H.CustomIntegrator(0.002, seed=3, constraintTolerance=1e-7)
    integ.addGlobalVariable("ke", 0.0)
    integ.addPerDofVariable("x1", 0)
    integ.addUpdateContextState()
    integ.addComputePerDof("v", "v+0.5*dt*f/m")
    integ.addComputePerDof("x", "x+dt*v")
    integ.addComputePerDof("x1", "x")
    integ.addConstrainPositions()
    integ.addComputePerDof("v", "v+0.5*dt*f/m+(x-x1)/dt")
    integ.addConstrainVelocities()
    integ.addComputeSum("ke", "m*v*v/2")
    c = H.Context(s, integ, plat)
    c.setPositions(w.positions); c.applyConstraints(1e-7); c.setVelocitiesToTemperature(300.0, 2)
    integ.step(steps)
    st = c.getState(getPositions=True, getVelocities=True, getEnergy=True)
    st.ke_global = integ.getGlobalVariable(0)
    st.mode = c.getPlatformProperty("IntegrationMode") if plat == "HIP" else ""
    c.close()
    return st


ref, hip, hip_overflow = run("Reference", False), run("HIP", False), run("HIP", True)
print("MODE", hip.mode)
print("REFERENCE dpos", np.abs(ref.positions - hip.positions).max(), "dvel", np.abs(ref.velocities - hip.velocities).max(), "ke", ref.ke_global, hip.ke_global, ref.kineticEnergy, hip.kineticEnergy)
print("OVERFLOW dpos", np.abs(hip.positions - hip_overflow.positions).max(), "dvel", np.abs(hip.velocities - hip_overflow.velocities).max(), "time", hip.time, hip_overflow.time)
o = w = None
pairs = T.water_box(%d, seed=9).constraints
d = np.linalg.norm(hip.positions[pairs[0][:, 0]] - hip.positions[pairs[0][:, 1]], axis=1)
print("CONSTRAINTS", np.abs(d - pairs[1]).max())
'''


def run_custom_integrator_case(tmp_path, emulated, n_side=6, grid=20, steps=8, cutoff=0.8):
    """-> dict(mode, dpos / dvel against Reference, relative difference of the kinetic-energy global, dpos / dvel of the run with an overflow,
    largest constraint violation, number of overflow messages)"""
    import re
    script = tmp_path / "custom_integrator_child.py"
    script.write_text(CHILD % (ROOT, emulated, steps, n_side, cutoff, grid, grid, grid, n_side))
    out = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=2400)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    f = lambda pattern: [float(v) for v in re.search(pattern, out.stdout).groups()]
    r = f(r"REFERENCE dpos (\S+) dvel (\S+) ke (\S+) (\S+) (\S+) (\S+)")
    o = f(r"OVERFLOW dpos (\S+) dvel (\S+) time (\S+) (\S+)")
    return {"mode": re.search(r"MODE (.*)", out.stdout).group(1).strip(), "dpos": r[0], "dvel": r[1], "ke_rel": abs(r[2] - r[3]) / r[2], "ke_state_rel": abs(r[4] - r[5]) / r[4],
            "overflow_dpos": o[0], "overflow_dvel": o[1], "times": (o[2], o[3]), "constraints": f(r"CONSTRAINTS (\S+)")[0],
            "overflows": out.stderr.count("neighbour list overflowed")}
