"""The dual pair list (include/openmm_hip_kernels.h, chunk_info_inner) through the platform: a water box stepped with the pruned list
switched on at test size (OPENMM_HIP_PRUNE=1; by default only systems above the fused size use it), a generous outer padding (rebuilds
are rare) and a tight inner one (the list is re-cut every few steps, on the device's own decision).  After the run the forces at the
final positions must be the Reference platform's: a pair lost by a cut -- or a cut that came a step late -- shows as an error of the
size of a pair force.  Shared by the CPU-emulator test and the GPU test."""
import os
import subprocess
import sys

from conftest import ROOT

CHILD = r'''
import os, sys, ctypes as C, numpy as np
os.environ["OPENMM_HIP_PRUNE"] = "1"
os.environ["OPENMM_HIP_NL_PADDING"] = "0.3"
os.environ["OPENMM_HIP_NL_INNER_PADDING"] = "0.04"
sys.path.insert(0, %r)
from openmm_amd import harness as H, testsystems as T
emulated = %r
H.load_hip_platform(emulated=emulated)
plugin = C.CDLL(os.path.join(H.EMU_DIR if emulated else H.LIB_DIR, "libOpenMMHIP.so"))
w = T.water_box(%d, seed=9)
w.pme_params = (float(np.sqrt(-np.log(2 * w.ewald_tol)) / w.cutoff), %d, %d, %d)
s, nb = w.build()
integ = H.Integrator(H.LANGEVIN_MIDDLE, 0.002, 300.0, 1.0, seed=5)
c = H.Context(s, integ, "HIP", %r)
c.setPositions(w.positions); c.applyConstraints(1e-6); c.setVelocitiesToTemperature(300.0, 2)
worst = 0.0
for leg in range(%d):
    integ.step(%d)
    st = c.getState(getPositions=True, getForces=True, getEnergy=True)
    stats = (C.c_longlong * 8)()
    plugin.ommhip_plugin_nl_stats(stats)
    s2, nb2 = w.build()
    r = H.Context(s2, H.Integrator(H.VERLET, 0.001), "Reference")
    r.setPositions(st.positions)
    ref = r.getState(getForces=True, getEnergy=True)
    r.close()
    rms = np.sqrt((ref.forces ** 2).sum(1).mean())
    err = np.linalg.norm(st.forces - ref.forces, axis=1).max() / rms
    worst = max(worst, err)
    print("leg", leg, "rows walked", stats[3], "rows as built", stats[7], "rebuilds", stats[5], "force error", err, "dE", st.potentialEnergy - ref.potentialEnergy, flush=True)
    assert 0 < stats[3] < stats[7], list(stats)
    assert err < 1e-4, err
    assert abs(st.potentialEnergy - ref.potentialEnergy) < 1e-5 * max(abs(ref.potentialEnergy), 5.0 * w.num_atoms)
c.close()
print("OK", worst)
'''


def run_pruned_list_case(tmp_path, emulated, n_side, grid, legs, steps, props=None):
    script = tmp_path / "pruned_child.py"
    script.write_text(CHILD % (ROOT, emulated, n_side, grid, grid, grid, props or {}, legs, steps))
    out = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=1500)
    assert "OK" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]
    return out.stdout
