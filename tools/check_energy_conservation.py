"""Energy conservation of the DHFR benchmark System (23 558 atoms, PME 0.9 nm, HBonds constraints + rigid water; the fixture of bench.py) under the
VerletIntegrator on the HIP platform -- no thermostat, no CMMotionRemover; the whole hot path of SURVEY 8(a): pair kernel on a list that is
rebuilt on the device's own displacement check, PME, bonded terms, the fused integration step with SETTLE / SHAKE in registers:
    python tools/check_energy_conservation.py [ps=20] [dt_fs=2] [constraint_tolerance=1e-6] [device list, e.g. "0,0"] [workload: dhfr | apoa1 | water1m] [skip_ps=0]
(skip_ps: the start of the run left out of the fit -- the tiled 1M-atom box relaxes the seams between its copies during the first picosecond,
bad contacts that cost ~200 kJ/mol of integration error on every path alike: profiles/r12/r12av_*)
(a device list: ONE Context over several ranks of the slab decomposition, DESIGN.md (e) -- a device named twice runs two ranks on one GPU over
the host-staged transport: slow, but every step goes through the halo exchange, the half-shell force return and the slab PME)
prints the total energy every 0.25 ps and one JSON line: the drift from a linear fit (kJ/mol per ps per degree of freedom; kT at 300 K per ns per
DOF -- the figure the MD literature quotes) and the RMS fluctuation around the fit."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openmm_amd import harness as H, testsystems as T

ps = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
dt_fs = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
tol = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-6
devices = sys.argv[4] if len(sys.argv) > 4 else ""
workload = sys.argv[5] if len(sys.argv) > 5 else "dhfr"
skip_ps = float(sys.argv[6]) if len(sys.argv) > 6 else 0.0
H.load_hip_platform()
if workload == "dhfr":
    w = T.dhfr()
else:
    import bench
    w = bench.make_workload(workload, seed=1)
w.cm_remover = False
positions, velocities = w.positions, getattr(w, "velocities", None)
if velocities is None:
    # a generated start (a jittered lattice): melt and thermalise it first, in a Context of its own (LangevinMiddle 300 K, 5 / ps, 6 ps)
    system, nb = w.build()
    relax = H.Integrator(H.LANGEVIN_MIDDLE, 0.002, 300.0, 5.0, seed=7, constraintTolerance=1e-6)
    c = H.Context(system, relax, "HIP")
    c.setPositions(positions)
    c.applyConstraints(1e-6)
    c.setVelocitiesToTemperature(300.0, 1)
    relax.step(3000)
    st = c.getState(getPositions=True, getVelocities=True)
    positions, velocities = st.positions, st.velocities
    c.close()
system, nb = w.build()
integ = H.Integrator(H.VERLET, dt_fs * 1e-3, constraintTolerance=tol)
c = H.Context(system, integ, "HIP", {"DeviceIndex": devices} if devices else None)
c.setPositions(positions)
c.applyConstraints(tol)
c.setVelocities(velocities)
c.applyVelocityConstraints(tol)
mode = c.getPlatformProperty("IntegrationMode")
every = max(1, int(round(0.25 / (dt_fs * 1e-3))))
blocks = int(round(ps / 0.25))
num_constraints = len(w.constraints[0]) if getattr(w, "constraints", None) is not None else system.getNumConstraints()
dof = 3 * w.num_atoms - num_constraints - 3
t, e = [], []
for k in range(blocks + 1):
    x = c.getState(getEnergy=True)
    t.append(k * every * dt_fs * 1e-3); e.append(x.potentialEnergy + x.kineticEnergy)
    if k % 8 == 0 or k == blocks:
        print("t = %6.2f ps  E = %.2f kJ/mol (potential %.1f kinetic %.1f, T = %.1f K)" % (t[-1], e[-1], x.potentialEnergy, x.kineticEnergy, 2 * x.kineticEnergy / (dof * 8.31446261815324e-3)), flush=True)
    if k < blocks:
        integ.step(every)
t, e = np.array(t), np.array(e)
keep = t >= skip_ps - 1e-9
t, e = t[keep], e[keep]
fit = np.polyfit(t, e, 1)
print(json.dumps({"workload": w.name, "devices": devices or "one", "integrator": "VerletIntegrator %.1f fs, constraint tolerance %g" % (dt_fs, tol), "integration_mode": mode, "ps": ps, "fit_from_ps": skip_ps, "degrees_of_freedom": dof,
                  "drift_kJ_per_mol_per_ps_per_dof": fit[0] / dof, "drift_kT_per_ns_per_dof": fit[0] * 1000 / dof / (8.31446261815324e-3 * 300),
                  "energy_fluctuation_rms_kJ_per_mol": float(np.std(e - np.polyval(fit, t))), "mean_energy_kJ_per_mol": float(e.mean()),
                  "kinetic_energy_kJ_per_mol": float(x.kineticEnergy)}))
