"""The solute of the amoebapme benchmark System (DHFR, 2 489 atoms, every kind of AMOEBA term: tests/golden/amoeba_dhfr_5dfr_amoeba2009.npz cut
before the water) on the HIP platform against the Reference platform: forces by force group, and a few steps of the benchmark's
MTSLangevinIntegrator (examples/benchmark.py:74-78) with the same seed on both.  Shared by the CPU-emulator test and the GPU test; runs in
a child process (the emulated and the product plugin must not meet in one process)."""
import subprocess
import sys

from conftest import ROOT

CHILD = r'''
import sys, os, numpy as np
sys.path.insert(0, %r)
from openmm_amd import harness as H, testsystems as T, forcefield_amoeba as A
emulated, atoms, steps, thermostat_steps = %r, %d, %d, %d
H.load_amoeba_plugins(emulated=emulated)
d = A.subset(A.load_description(os.path.join(%r, "tests", "golden", "amoeba_dhfr_5dfr_amoeba2009.npz")), atoms)
out = {}
for plat in ("Reference", "HIP"):
    w = T.AmoebaWorkload(d, cutoff=0.7, vdw_cutoff=0.9, epsilon=1e-6, grid=(64, 64, 64), a_ewald=float(np.sqrt(-np.log(2 * 7.5e-4)) / 0.7))
    w.cm_remover = True
    s, mp, vdw = w.build()
    # friction 0: the two platforms draw different random numbers, without noise the integrator is the deterministic RESPA scheme
    integ = H.MTSLangevinIntegrator(300.0, 0.0, 0.002, [(0, 2), (1, 1)], seed=11)
    ctx = H.Context(s, integ, plat)
    ctx.setPositions(w.positions)
    for name, groups in (("valence", 1), ("nonbonded", 2)):
        st = ctx.getState(getForces=True, getEnergy=True, groups=groups)
        out[plat + "_f_" + name], out[plat + "_e_" + name] = st.forces, st.potentialEnergy
    ctx.setVelocitiesToTemperature(300.0, 3)
    before = np.array(H.amoeba_native_evaluations()) if plat == "HIP" else None
    integ.step(steps)
    st = ctx.getState(getPositions=True, getVelocities=True, getEnergy=True)
    out[plat + "_pos"], out[plat + "_vel"], out[plat + "_ke"] = st.positions, st.velocities, st.kineticEnergy
    if plat == "HIP":
        out["native"] = np.array(list(H.amoeba_native_evaluations()) + [H.valence_lists_launched()])
        out["mode"] = np.array(ctx.getPlatformProperty("IntegrationMode"))
        out["evaluations_per_step"] = (np.array(H.amoeba_native_evaluations()) - before) / float(steps)
    ctx.close()
# the thermostat: from rest, friction so strong that the noise decides the temperature within a few steps whatever the forces do
w = T.AmoebaWorkload(d, cutoff=0.7, vdw_cutoff=0.9, epsilon=1e-5, grid=(64, 64, 64), a_ewald=float(np.sqrt(-np.log(2 * 7.5e-4)) / 0.7))
s, mp, vdw = w.build()
integ = H.MTSLangevinIntegrator(300.0, 300.0, 0.001, [(0, 2), (1, 1)], seed=5)
ctx = H.Context(s, integ, "HIP")
ctx.setPositions(w.positions)
integ.step(thermostat_steps)
out["thermostat_ke"] = ctx.getState(getEnergy=True).kineticEnergy
ctx.close()
np.savez(sys.argv[1], **out)
'''


def run_amoeba_dhfr_case(tmp_path, emulated, atoms=2489, steps=3, thermostat_steps=12):
    """-> dict: worst force difference relative to the RMS force and relative energy difference per group, largest position / velocity
    difference after `steps` MTS steps without friction, the temperature a strongly damped MTS Langevin run reaches from rest, native evaluation counts (vdw, multipole, lists of AMOEBA valence terms)"""
    import numpy as np
    script = tmp_path / "amoeba_dhfr_child.py"
    script.write_text(CHILD % (ROOT, emulated, atoms, steps, thermostat_steps, ROOT))
    path = str(tmp_path / "amoeba_dhfr.npz")
    run = subprocess.run([sys.executable, str(script), path], capture_output=True, text=True, timeout=3000)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-3000:]
    z = np.load(path)
    res = {"native": tuple(int(v) for v in z["native"])}
    for name in ("valence", "nonbonded"):
        ref, hip = z["Reference_f_" + name], z["HIP_f_" + name]
        res["force_" + name] = float(np.sqrt(((ref - hip) ** 2).sum(1)).max() / np.sqrt((ref ** 2).sum(1).mean()))
        res["energy_" + name] = float(abs(z["Reference_e_" + name] - z["HIP_e_" + name]) / abs(z["Reference_e_" + name]))
    res["dpos"] = float(np.abs(z["Reference_pos"] - z["HIP_pos"]).max())
    res["dvel"] = float(np.abs(z["Reference_vel"] - z["HIP_vel"]).max())
    res["ke"] = (float(z["Reference_ke"]), float(z["HIP_ke"]))
    res["mode"] = str(z["mode"])
    res["evaluations_per_step"] = tuple(float(v) for v in z["evaluations_per_step"])
    res["thermostat_temperature"] = 2 * float(z["thermostat_ke"]) / (3 * atoms * 8.31446261815324e-3)
    return res
