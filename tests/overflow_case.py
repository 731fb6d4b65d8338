"""Neighbour-list overflow on a device-triggered rebuild: the device freezes the integration
while the list is incomplete, the host notices late (every 16th evaluation, or at the next download), grows the list and
redoes the skipped steps in order -- the trajectory must be the one of an undisturbed run.  Shared by the CPU-emulator test
and the GPU test; the reference's remedy is the immediate retry of ContextImpl.cpp:298-307 / CudaNonbondedUtilities.cpp:423-456."""
import os
import subprocess
import sys

from conftest import ROOT

CHILD = r'''
import os, sys, numpy as np
sys.path.insert(0, %r)
from openmm_amd import harness as H, testsystems as T
H.load_hip_platform(emulated=%r)


def run(shrink, steps):
    if shrink:
        os.environ["OPENMM_HIP_DEBUG_SHRINK_LIST_AFTER"] = "3"      # the 3rd evaluation rebuilds into an allocation 8 chunks too small
    else:
        os.environ.pop("OPENMM_HIP_DEBUG_SHRINK_LIST_AFTER", None)
    w = T.water_box(%d, seed=9)
    w.pme_params = (float(np.sqrt(-np.log(2 * w.ewald_tol)) / w.cutoff), %d, %d, %d)
    w.cm_remover = True
    s, nb = w.build()
    integ = H.Integrator(H.LANGEVIN_MIDDLE, 0.002, 300.0, 1.0, seed=5)
    c = H.Context(s, integ, "HIP", %r)
    c.setPositions(w.positions); c.applyConstraints(1e-6); c.setVelocitiesToTemperature(300.0, 2)
    integ.step(steps)
    # an energy on its own first (ADVICE r2): while the device is frozen this evaluation is the one that finds the overflow -- it must
    # come back with the energy of the complete list at the up-to-date positions, not with a sum over the rows that fitted
    e = c.getState(getEnergy=True).potentialEnergy
    st = c.getState(getPositions=True, getVelocities=True, getEnergy=True)
    # (two evaluations of one configuration differ by ~1e-9 of the energy on the GPU: the charge grid is summed with float atomics; an energy from
    # an incomplete list is off by 1e-3 and more)
    assert abs(e - st.potentialEnergy) < 1e-7 * max(abs(e), 1.0), (e, st.potentialEnergy)
    c.close()
    return st


for steps in %r:        # 5: found at the download (getState); 40: found by the lazy read-back during the run
    a, b = run(False, steps), run(True, steps)
    dpos, dvel = np.abs(a.positions - b.positions).max(), np.abs(a.velocities - b.velocities).max()
    print(steps, "steps: dpos", dpos, "dvel", dvel, "time", a.time, b.time, flush=True)
    assert a.time == b.time
    assert dpos < %g and dvel < %g, (dpos, dvel)
    assert abs(a.potentialEnergy - b.potentialEnergy) < 2e-6 * max(abs(a.potentialEnergy), 5.0 * len(a.positions)), (a.potentialEnergy, b.potentialEnergy)
print("OK")
'''


def run_overflow_case(tmp_path, emulated, n_side, grid, pos_tol, vel_tol, props=None, step_counts=(5, 20)):
    script = tmp_path / "overflow_child.py"
    script.write_text(CHILD % (ROOT, emulated, n_side, grid, grid, grid, props, tuple(step_counts), pos_tol, vel_tol))
    out = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=1200)
    assert "OK" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]
    assert out.stderr.count("neighbour list overflowed") == len(step_counts), out.stderr[-2000:]
    return out.stdout
