"""What the CM-momentum tail of the fused step kernel costs on the DHFR benchmark System: ns/day with and without the CMMotionRemover
(3 000 steps each, two repetitions).  Round 5 diagnostic."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openmm_amd import harness as H, testsystems as T
H.load_hip_platform()
for rep in range(2):
    for cm in (True, False):
        w = T.dhfr()
        w.cm_remover = cm
        s, nb = w.build()
        integ = H.Integrator(H.LANGEVIN_MIDDLE, 0.002, 300.0, 1.0, seed=1)
        ctx = H.Context(s, integ, "HIP")
        ctx.setPositions(w.positions)
        ctx.setVelocities(w.velocities)
        integ.step(300); ctx.getState(getEnergy=True)
        t0 = time.perf_counter()
        integ.step(3000); ctx.getState(getEnergy=True)
        dt = time.perf_counter() - t0
        print("cm_remover=%s  %.2f us per step  %.1f ns/day" % (cm, dt / 3000 * 1e6, 0.002 * 3000 / dt * 86400 * 1e-3), flush=True)
        ctx.close()
