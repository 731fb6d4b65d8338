#!/usr/bin/env python
"""Where a line-pass FFT workgroup spends its time (OMMHIP_FFT_DIAG=1; workgroup 0, thread 0; clock ticks summed over launches):
staging the tile into LDS (includes the wait for its global reads), issuing the next tile's reads, the radix passes, the store."""
import ctypes as C, os, sys
os.environ["OMMHIP_FFT_DIAG"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openmm_amd import harness as H, testsystems as T, capi
H.load_hip_platform()
k = capi.load()
n_side = int(sys.argv[1]) if len(sys.argv) > 1 else 69
w = T.water_box(n_side, seed=1)
s, nb = w.build()
integ = H.Integrator(H.LANGEVIN_MIDDLE, 0.002, 300.0, 1.0, seed=1)
c = H.Context(s, integ, "HIP")
c.setPositions(w.positions)
c.setVelocitiesToTemperature(300.0, 1)
integ.step(50)
c.getState(getEnergy=True)
out = (C.c_ulonglong * 32)()
rc = k.lib.ommhip_fft_diag(out)
print("grid", nb.getPMEParametersInContext(c)[1:], "rc", rc)
for mode, name in ((1, "z r2c"), (0, "y c2c"), (3, "x conv"), (2, "z c2r")):
    d = [out[8 * mode + i] for i in range(6)]
    if d[0]:
        n = float(d[0])
        print("%-7s launches %4d | ticks per launch: total %8.0f = staging(+read wait) %7.0f + issue next %6.0f + passes %7.0f + store %6.0f" % (
            name, d[0], d[1] / n, d[2] / n, d[3] / n, d[4] / n, d[5] / n))
c.close()
