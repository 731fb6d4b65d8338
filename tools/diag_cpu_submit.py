#!/usr/bin/env python
"""How far ahead of the GPU does the host run?  Time for step(K) to return (launches queued) vs. time until the device is idle."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openmm_amd import harness as H, testsystems as T
H.load_hip_platform()
w = T.dhfr_like(seed=1)
s, nb = w.build()
integ = H.Integrator(H.LANGEVIN_MIDDLE, 0.002, 300.0, 1.0, seed=1, constraintTolerance=1e-5)
props = dict(kv.split("=") for kv in sys.argv[1:])
c = H.Context(s, integ, "HIP", props)
c.setPositions(w.positions); c.setVelocities(w.velocities)
integ.step(300); c.getState(getEnergy=True)
for K in (100, 400, 1600):
    t0 = time.perf_counter(); integ.step(K); t1 = time.perf_counter(); c.getState(getEnergy=True); t2 = time.perf_counter()
    print("K=%d: host returned after %.1f us/step, device idle after %.1f us/step" % (K, 1e6 * (t1 - t0) / K, 1e6 * (t2 - t0) / K))
