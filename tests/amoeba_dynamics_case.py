"""A few Verlet steps of a small AMOEBA water box (mutual polarization) on the native kernels, twice: as round 3 ran it -- pair lists rebuilt
at every evaluation, every dipole solve started from the direct dipoles with a host round trip per iteration -- and with this round's
machinery (lists with a Verlet skin rebuilt on displacement, first guess extrapolated from the previous steps, convergence decided on the
device with iterations enqueued ahead).  Both must walk the same trajectory: the lists only change which pairs are LOOKED at, the guess only
where the solver starts.  Shared by the CPU-emulator test and the GPU test; each variant runs in a process of its own (the knobs are read
once per process)."""
import os
import re
import subprocess
import sys

from conftest import ROOT

CHILD = r'''
import sys, numpy as np
sys.path.insert(0, %r)
from openmm_amd import harness as H, testsystems as T
H.load_amoeba_plugins(emulated=%r)
w = T.amoeba_water_box(%d, seed=3, polarization=H.Mutual, cutoff=%r, vdw_cutoff=%r, grid=(%d,) * 3, a_ewald=5.4459052, epsilon=1e-6)
s, mp, vdw = w.build()
integ = H.Integrator(H.VERLET, 0.001)
ctx = H.Context(s, integ, "HIP")
ctx.setPositions(w.positions)
ctx.minimizeEnergy(50.0, %d)
ctx.setVelocitiesToTemperature(300.0, 5)
ctx.getState(getEnergy=True)
before, builds0 = H.amoeba_native_evaluations(), H.amoeba_list_builds()
integ.step(%d)
st = ctx.getState(getPositions=True, getForces=True, getEnergy=True)
after = H.amoeba_native_evaluations()
np.save(sys.argv[1], np.concatenate([st.positions.reshape(-1), st.forces.reshape(-1), [st.potentialEnergy, after[0] - before[0], after[1] - before[1]], np.array(H.amoeba_list_builds()) - np.array(builds0)]))
'''


def run_amoeba_dynamics_case(tmp_path, emulated, n_side=5, steps=12, cutoff=0.6, grid=20, minimize=25):
    """-> dict: largest position difference (nm), worst force difference relative to the RMS force, relative energy difference between the
    two variants after `steps` steps; list builds and evaluations and the solver's iterations per evaluation of each variant"""
    import numpy as np
    script = tmp_path / "amoeba_dynamics_child.py"
    script.write_text(CHILD % (ROOT, emulated, n_side, cutoff, cutoff, grid, minimize, steps))
    res = {}
    old = {"OPENMM_HIP_AMOEBA_SKIN": "0", "OPENMM_HIP_AMOEBA_NO_PREDICTOR": "1", "OPENMM_HIP_AMOEBA_CHECK_EVERY_ITERATION": "1"}
    for name, env in (("round3", old), ("now", {})):
        path = str(tmp_path / ("amoeba_dyn_%s.npy" % name))
        out = subprocess.run([sys.executable, str(script), path], capture_output=True, text=True, timeout=3000, env=dict(os.environ, OPENMM_HIP_AMOEBA_DEBUG="1", **env))
        assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
        v = np.load(path)
        n3 = (len(v) - 5) // 2
        # the solver's own report, one line per solve: the last `steps` + 1 belong to the steps and the closing getState
        its = [int(m.group(1)) for m in re.finditer(r"amoeba solver: (\d+) iterations", out.stderr)]
        res[name] = {"pos": v[:n3].reshape(-1, 3), "forces": v[n3:2 * n3].reshape(-1, 3), "energy": v[2 * n3], "evaluations": (int(v[2 * n3 + 1]), int(v[2 * n3 + 2])),
                     "builds": (int(v[2 * n3 + 3]), int(v[2 * n3 + 4])), "iterations": its[-(steps + 1):]}
    a, b = res["round3"], res["now"]
    rms = np.sqrt((a["forces"] ** 2).sum(1).mean())
    return {"dpos": float(np.abs(a["pos"] - b["pos"]).max()), "dforce": float(np.sqrt(((a["forces"] - b["forces"]) ** 2).sum(1)).max() / rms),
            "denergy": float(abs(a["energy"] - b["energy"]) / max(abs(a["energy"]), 1.0)), "round3": {k: a[k] for k in ("evaluations", "builds", "iterations")},
            "now": {k: b[k] for k in ("evaluations", "builds", "iterations")}}
