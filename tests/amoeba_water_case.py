"""AMOEBA water box through the Python harness (openmm_amd.testsystems.AmoebaWaterWorkload: AmoebaMultipoleForce PME + AmoebaVdwForce with
the amoeba2009 water parameters): the native kernels of libOpenMMAmoebaHIP.so -- pair scan in the platform's slot order, tiles farther
apart than the cutoff skipped while the per-atom pair lists are built -- against (a) the AMOEBA plugin's own Reference kernel for the multipole force
(OPENMM_HIP_REFERENCE_AMOEBA_MULTIPOLE=1 hands the force to it) and (b) the same native kernels scanning every tile in atom order
(OPENMM_HIP_AMOEBA_NO_TILES=1, the first version, itself checked against the Reference by the reference's test bodies).  Shared by the
CPU-emulator test and the GPU test; every variant runs in a process of its own (the knobs are read once per process)."""
import os
import subprocess
import sys

from conftest import ROOT

CHILD = r'''
import sys, numpy as np
sys.path.insert(0, %r)
from openmm_amd import harness as H, testsystems as T
H.load_amoeba_plugins(emulated=%r)
w = T.amoeba_water_box(%d, seed=3, polarization=%s, cutoff=0.7, vdw_cutoff=%r, grid=(%d,) * 3, a_ewald=5.4459052, epsilon=1e-6)
s, mp, vdw = w.build()
ctx = H.Context(s, H.Integrator(H.VERLET, 0.001), "HIP")
ctx.setPositions(w.positions)
st = ctx.getState(getForces=True, getEnergy=True)
np.save(sys.argv[1], np.concatenate([st.forces.reshape(-1), [st.potentialEnergy], H.amoeba_native_evaluations()]))
'''


def run_amoeba_water_case(tmp_path, emulated, n_side, grid, mutual, vdw_cutoff=0.9, with_reference=True, tiles_env=None):
    """-> dict of the worst force difference relative to the RMS force and the relative energy difference, native (tiles) against the
    Reference multipole kernel and against the full scan"""
    import numpy as np
    script = tmp_path / "amoeba_water_child.py"
    script.write_text(CHILD % (ROOT, emulated, n_side, "H.Mutual" if mutual else "H.Direct", vdw_cutoff, grid))
    res = {}
    # full_scan: the list builder looks at every tile, and starts from lists of 8 entries per atom (two rounds of growing them)
    variants = (("tiles", dict(tiles_env or {})), ("reference", {"OPENMM_HIP_REFERENCE_AMOEBA_MULTIPOLE": "1"}), ("full_scan", {"OPENMM_HIP_AMOEBA_NO_TILES": "1", "OPENMM_HIP_AMOEBA_PAIR_CAP": "8"}))
    for name, env in variants:
        if name == "reference" and not with_reference:
            continue
        path = str(tmp_path / ("amoeba_%s.npy" % name))
        out = subprocess.run([sys.executable, str(script), path], capture_output=True, text=True, timeout=1500, env=dict(os.environ, **env))
        assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
        v = np.load(path)
        res[name] = (v[:-3].reshape(-1, 3), v[-3], int(v[-2]), int(v[-1]))
    f, e, n_vdw, n_mp = res["tiles"]
    assert n_vdw == 1 and n_mp == 1, "the native kernels did not run"
    assert res["full_scan"][3] == 1 and (not with_reference or res["reference"][3] == 0)
    rms = np.sqrt((res["full_scan"][0] ** 2).sum(1).mean())
    summary = {}
    for name in ("reference", "full_scan") if with_reference else ("full_scan",):
        summary[name] = (float(np.sqrt(((f - res[name][0]) ** 2).sum(1)).max() / rms), float(abs(e - res[name][1]) / max(abs(res[name][1]), 1.0)))
    return summary
