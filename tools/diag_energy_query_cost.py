import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from openmm_amd import harness as H, testsystems as T
H.load_hip_platform()
w = T.dhfr()
s, nb = w.build()
integ = H.Integrator(H.LANGEVIN_MIDDLE, 0.002, 300.0, 1.0, seed=1)
ctx = H.Context(s, integ, "HIP")
ctx.setPositions(w.positions)
if getattr(w, "velocities", None) is not None: ctx.setVelocities(w.velocities)
else: ctx.setVelocitiesToTemperature(300.0, 1)
integ.step(300); ctx.getState(getEnergy=True)
for rep in range(3):
    t0 = time.perf_counter(); integ.step(20); t1 = time.perf_counter(); st = ctx.getState(getEnergy=True); t2 = time.perf_counter()
    print("20 steps enqueue %.0f us, getState(energy) incl. drain %.0f us, total %.0f us" % ((t1 - t0) * 1e6, (t2 - t1) * 1e6, (t2 - t0) * 1e6))
integ.step(20); ctx.getState(getEnergy=True)
t0 = time.perf_counter()
for k in range(20): ctx.getState(getEnergy=True)
print("getState(getEnergy) alone: %.0f us each" % ((time.perf_counter() - t0) / 20 * 1e6))
t0 = time.perf_counter()
for k in range(20): ctx.getState(getPositions=True)
print("getState(getPositions) alone: %.0f us each" % ((time.perf_counter() - t0) / 20 * 1e6))
