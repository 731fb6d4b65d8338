"""Timing experiment: the three fused pair+FFT launches with one of the two halves switched off (results are wrong then,
so only a few steps are run and no energies are checked)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openmm_amd import capi, harness as H, testsystems as T
H.load_hip_platform()
k = capi.load()
w = T.dhfr_like(seed=1)
system, nb = w.build()
integ = H.Integrator(H.LANGEVIN_MIDDLE, 0.002, 300.0, 1.0, seed=1, constraintTolerance=1e-5)
ctx = H.Context(system, integ, "HIP", {})
ctx.setPositions(w.positions); ctx.applyConstraints(1e-5); ctx.setVelocities(w.velocities)
integ.step(20)
k.lib.ommhip_profile_reset(); k.lib.ommhip_profile_enable(1)
integ.step(int(os.environ.get("STEPS", "60")))
k.lib.ommhip_profile_enable(0)
calls, ms = C.c_longlong(), C.c_double()
k.lib.ommhip_profile_collect(0, C.byref(calls), C.byref(ms))
print(os.environ.get("TAG", ""), "pairs+fft launches: %.1f us over %d calls" % (1e3 * ms.value / max(calls.value, 1), calls.value))
