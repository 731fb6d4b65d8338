"""Energy conservation of the amoebapme System (DHFR, amoeba2009, all forces and the integrator native) under the reference's MTSIntegrator
(rRESPA, no thermostat; wrappers/python/openmm/mtsintegrator.py:30-110), 1 fs outer step with the valence terms twice per step, after a
short relaxation with the MTSLangevinIntegrator of the benchmark (the PDB coordinates are amber-equilibrated):
    python tools/check_amoeba_energy_conservation.py [steps=400] [epsilon=1e-5]
prints total energy every 20 steps and the drift (kJ/mol per ps per degree of freedom, and in units of kT at 300 K per ns per DOF)."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openmm_amd import harness as H, testsystems as T

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
eps = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-5
H.load_amoeba_plugins()
w = T.amoeba_dhfr(epsilon=eps)
# relax under the benchmark's thermostat first, in a Context of its own
s, mp, vdw = w.build()
relax = H.MTSLangevinIntegrator(300.0, 5.0, 0.001, [(0, 2), (1, 1)], seed=3)
c = H.Context(s, relax, "HIP")
c.setPositions(w.positions); c.setVelocitiesToTemperature(300.0, 5)
relax.step(600)
st = c.getState(getPositions=True, getVelocities=True)
c.close()
s, mp, vdw = w.build()
w.cm_remover = False
integ = H.MTSIntegrator(0.001, [(0, 2), (1, 1)])
c = H.Context(s, integ, "HIP")
c.setPositions(st.positions); c.setVelocities(st.velocities)
mode = c.getPlatformProperty("IntegrationMode")
t, e = [], []
for k in range(steps // 20 + 1):
    x = c.getState(getEnergy=True)
    t.append(0.001 * 20 * k); e.append(x.potentialEnergy + x.kineticEnergy)
    print("t = %.3f ps  E = %.3f kJ/mol (potential %.1f kinetic %.1f)" % (t[-1], e[-1], x.potentialEnergy, x.kineticEnergy), flush=True)
    if k < steps // 20: integ.step(20)
slope = np.polyfit(t, e, 1)[0]
dof = 3 * w.num_atoms
print(json.dumps({"workload": w.name, "integrator": "MTSIntegrator 1 fs [(0,2),(1,1)]", "integration_mode": mode, "mutual_epsilon": eps, "steps": steps,
                  "drift_kJ_per_mol_per_ps_per_dof": slope / dof, "drift_kT_per_ns_per_dof": slope * 1000 / dof / (8.31446261815324e-3 * 300),
                  "energy_fluctuation_rms_kJ_per_mol": float(np.std(np.array(e) - np.polyval(np.polyfit(t, e, 1), t))), "mean_energy": float(np.mean(e))}))
