"""Restart of a stochastic CustomIntegrator (the benchmark's MTSLangevinIntegrator, device interpreter) and of the native LangevinMiddle
integrator from a checkpoint: 6 steps, checkpoint, 6 more steps -- against (a) the same Context rewound to the checkpoint and (b) a NEW
Context (integrator seed 0 = "pick one", as a restarted job would) that loads the blob.  All three must walk the same trajectory: the
per-DOF noise is keyed by (seed, draw counter, atom), both of which travel in the checkpoint.  A ComputeGlobal step that
draws `gaussian` on the host rides along (the host generator's state is part of the checkpoint, as on the Reference platform,
ReferenceKernels.cpp:282-294).  Shared by the CPU-emulator test and the GPU test."""
import re
import subprocess
import sys

from conftest import ROOT

CHILD = r'''
import sys, numpy as np
sys.path.insert(0, %r)
from openmm_amd import harness as H, testsystems as T
H.load_hip_platform(emulated=%r)
w = T.water_box(%d, seed=21, cutoff=%r)
w.pme_params = (float(np.sqrt(-np.log(2 * w.ewald_tol)) / w.cutoff), %d, %d, %d)


def integrator(kind, seed):
    if kind == "native":
        return H.Integrator(H.LANGEVIN_MIDDLE, 0.002, 300.0, 1.0, seed=seed, constraintTolerance=1e-7)
    integ = H.MTSLangevinIntegrator(300.0, 1.0, 0.002, [(0, 1)], seed=seed, constraintTolerance=1e-7)
    integ.addGlobalVariable("kick", 0.0)
    integ.addComputeGlobal("kick", "0.001*gaussian")          # a host-side draw per step
    integ.addComputePerDof("v", "v+kick")
    integ.addConstrainVelocities()
    return integ


def context(kind, seed):
    s, nb = w.build()
    integ = integrator(kind, seed)
    c = H.Context(s, integ, "HIP")
    return c, integ


for kind in ("custom", "native"):
    c, integ = context(kind, 5)
    c.setPositions(w.positions); c.applyConstraints(1e-7); c.setVelocitiesToTemperature(300.0, 2)
    integ.step(6)
    blob = c.createCheckpoint()
    integ.step(6)
    a = c.getState(getPositions=True, getVelocities=True)
    mode = c.getPlatformProperty("IntegrationMode")
    c.loadCheckpoint(blob)
    integ.step(6)
    b = c.getState(getPositions=True, getVelocities=True)
    c.close()
    c2, integ2 = context(kind, 0)
    c2.setPositions(w.positions)
    c2.loadCheckpoint(blob)
    integ2.step(6)
    d = c2.getState(getPositions=True, getVelocities=True)
    c2.close()
    # control: a new Context WITHOUT the checkpoint's noise state walks somewhere else
    print(kind, "MODE", mode)
    print(kind, "RESULT same %%.3e %%.3e new %%.3e %%.3e time %%g %%g %%g" %% (np.abs(a.positions - b.positions).max(), np.abs(a.velocities - b.velocities).max(),
          np.abs(a.positions - d.positions).max(), np.abs(a.velocities - d.velocities).max(), a.time, b.time, d.time))
'''


def run_checkpoint_case(tmp_path, emulated, n_side=5, grid=16, cutoff=0.7):
    script = tmp_path / "checkpoint_child.py"
    script.write_text(CHILD % (ROOT, emulated, n_side, cutoff, grid, grid, grid))
    out = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=2400)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    result = {}
    for kind in ("custom", "native"):
        v = [float(x) for x in re.search(kind + r" RESULT same (\S+) (\S+) new (\S+) (\S+) time (\S+) (\S+) (\S+)", out.stdout).groups()]
        result[kind] = {"mode": re.search(kind + r" MODE (.*)", out.stdout).group(1).strip(), "same": v[:2], "new": v[2:4], "times": v[4:]}
    return result
