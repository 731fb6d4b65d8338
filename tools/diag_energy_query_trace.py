"""What a closing Context.getState(getEnergy=True) is made of on the DHFR benchmark System: 300 steps, then 40 energy queries back to back (run
under rocprofv3 --kernel-trace --stats by tools/visit_r11c.sh; the per-query wall time is printed as well)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openmm_amd import harness as H, testsystems as T
H.load_hip_platform()
w = T.dhfr()
s, nb = w.build()
integ = H.Integrator(H.LANGEVIN_MIDDLE, 0.002, 300.0, 1.0, seed=1)
ctx = H.Context(s, integ, "HIP")
ctx.setPositions(w.positions)
ctx.setVelocities(w.velocities)
integ.step(300); ctx.getState(getEnergy=True)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
t0 = time.perf_counter()
for k in range(n):
    integ.step(1)
    ctx.getState(getEnergy=True)
t1 = time.perf_counter()
integ.step(n); ctx.getState(getEnergy=True)
t2 = time.perf_counter()
print("step + getState(getEnergy): %.1f us per pair; %d steps + one query: %.1f us per step" % ((t1 - t0) / n * 1e6, n, (t2 - t1) / n * 1e6))
