"""Relax and equilibrate the synthetic benchmark workloads on the reference's own CPU platform
(oracle/_ref/libOpenMMCPU.so) and store coordinates + velocities as fixtures under tests/golden/.

The generators in openmm_amd/testsystems.py place molecules on lattices with random orientations, which
is far from equilibrium; production-rate dynamics at 2 fs needs a relaxed start.  This script is run
once in the build container (python tools/make_workload_fixtures.py [dhfr] [water24k]); bench.py and the
tests only read the resulting .npz files.
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openmm_amd import harness as H, testsystems as T  # noqa: E402


def relax(w, out, minimize_iters=300, schedule=((0.0005, 400), (0.001, 400), (0.002, 1200))):
    H.load_cpu_platform()
    system, nb = w.build()
    integ = H.Integrator(H.LANGEVIN_MIDDLE, schedule[0][0], 300.0, 5.0, seed=3)
    ctx = H.Context(system, integ, "CPU")
    ctx.setPositions(w.positions)
    ctx.applyConstraints(1e-6)
    t0 = time.time()
    e0 = ctx.getState(getEnergy=True).potentialEnergy
    ctx.minimizeEnergy(10.0, minimize_iters)
    e1 = ctx.getState(getEnergy=True).potentialEnergy
    print("%s: minimised %.1f -> %.1f kJ/mol in %.0f s" % (w.name, e0, e1, time.time() - t0), flush=True)
    ctx.setVelocitiesToTemperature(300.0, 5)
    for dt, steps in schedule:
        H._check(H.lib().omm_integrator_set_step_size(integ.h, H.C.c_double(dt)))
        integ.step(steps)
        st = ctx.getState(getEnergy=True)
        print("  dt=%.4f ps x %d: PE %.1f KE %.1f (%.0f s)" % (dt, steps, st.potentialEnergy, st.kineticEnergy, time.time() - t0), flush=True)
    st = ctx.getState(getPositions=True, getVelocities=True, getEnergy=True)
    np.savez_compressed(out, positions=st.positions.astype(np.float32), velocities=st.velocities.astype(np.float32),
                        potential_energy=st.potentialEnergy, kinetic_energy=st.kineticEnergy)
    print("  wrote %s (%.0f kB)" % (out, os.path.getsize(out) / 1e3))


if __name__ == "__main__":
    which = sys.argv[1:] or ["dhfr"]
    if "dhfr" in which:
        relax(T.dhfr_like(seed=1, relaxed=False), os.path.join(ROOT, "tests", "golden", "dhfr_like_seed1_equilibrated.npz"))
    if "water24k" in which:
        relax(T.water_box(20, seed=1), os.path.join(ROOT, "tests", "golden", "water_24000_seed1_equilibrated.npz"), schedule=((0.0005, 300), (0.002, 700)))
