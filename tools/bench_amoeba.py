"""Times the AMOEBA workloads of bench.py's `extra_workloads` (amoeba_water, or with --dhfr amoeba_dhfr: the amoebapme benchmark System with its
MTSLangevinIntegrator, 2 fs) on their own (GPU box):
    python tools/bench_amoeba.py [--n-side N | --tile | --dhfr] [--steps K] [--direct] [--grid G]
prints one JSON line (ms per step, ns/day at 1 fs).  OPENMM_HIP_AMOEBA_NO_TILES=1 gives the scan over all atoms for an A/B."""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openmm_amd import harness as H, testsystems as T

p = argparse.ArgumentParser()
p.add_argument("--n-side", type=int, default=0)
p.add_argument("--steps", type=int, default=10)
p.add_argument("--direct", action="store_true")
p.add_argument("--grid", type=int, default=80)
p.add_argument("--dhfr", action="store_true")
p.add_argument("--warm", type=int, default=6, help="untimed steps before the timed ones (the solver's first guess uses up to four earlier steps)")
a = p.parse_args()
H.load_amoeba_plugins()
kw = dict(cutoff=0.7, vdw_cutoff=0.9, polarization=H.Direct if a.direct else H.Mutual, epsilon=1e-5, ewald_tol=7.5e-4, grid=(a.grid,) * 3, a_ewald=float(np.sqrt(-np.log(2 * 7.5e-4)) / 0.7))
if a.dhfr:
    w = T.amoeba_dhfr(epsilon=1e-5)
    s, mp, vdw = w.build()
    integ = H.MTSLangevinIntegrator(300.0, 1.0, 0.002, [(0, 2), (1, 1)], seed=7)
else:
    w = T.amoeba_water_box(a.n_side, seed=3, **kw) if a.n_side else T.amoeba_water_tile(**kw)
    s, mp, vdw = w.build()
    integ = H.Integrator(H.VERLET, 0.001)
ctx = H.Context(s, integ, "HIP")
ctx.setPositions(w.positions)
ctx.setVelocitiesToTemperature(300.0, 5)
integ.step(a.warm)
e0 = ctx.getState(getEnergy=True).potentialEnergy
b0, s0 = H.amoeba_list_builds(), H.amoeba_solver_iterations()
t0 = time.perf_counter()
integ.step(a.steps)
e1 = ctx.getState(getEnergy=True).potentialEnergy
dt = time.perf_counter() - t0
b1, s1 = H.amoeba_list_builds(), H.amoeba_solver_iterations()
print(json.dumps({"workload": w.name, "atoms": w.num_atoms, "polarization": "direct" if a.direct else "mutual", "grid": a.grid, "steps": a.steps,
                  "ms_per_step": round(1e3 * dt / a.steps, 3), "ns_per_day": round((2e-6 if a.dhfr else 1e-6) * a.steps / dt * 86400, 4), "step_fs": 2 if a.dhfr else 1,
                  "integration_mode": ctx.getPlatformProperty("IntegrationMode"), "E0": e0, "E1": e1,
                  "native_evaluations": H.amoeba_native_evaluations(), "list_builds_per_step": [round((b1[k] - b0[k]) / a.steps, 3) for k in range(2)],
                  "solver_iterations_per_solve": round((s1[1] - s0[1]) / max(1, s1[0] - s0[0]), 2), "no_tiles": os.environ.get("OPENMM_HIP_AMOEBA_NO_TILES") is not None}))
