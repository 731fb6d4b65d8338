"""Reference-kernel forces of the AMOEBA water workload bench.py times (extra_workloads.amoeba_water: 12 167 AMOEBA waters on the
coordinates of the equilibrated tile, multipole PME 80^3 with cutoff 0.7 nm, buffered 14-7 vdW 0.9 nm, harmonic bonds / angles) at the
INITIAL configuration, from the AMOEBA plugin's own Reference kernels on the Reference platform (oracle: /root/reference compiled by
openmm_host/Makefile).  A sample of atoms is kept, as tools/make_golden_water1m.py does for the 1M-atom box.

    python tools/make_golden_amoeba_water_tile.py [direct|mutual] [sample=12000]

mutual: epsilon 1e-6 -- tighter than the 1e-5 of the timed run, so that the bar of the comparison is the kernel, not the solver.
Runs in the build container only (needs build/openmm/lib).
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from openmm_amd import harness as H, testsystems as T
    kind = sys.argv[1] if len(sys.argv) > 1 else "direct"
    sample = int(sys.argv[2]) if len(sys.argv) > 2 else 12000
    H.lib()
    H._check(H.lib().omm_load_plugin(os.path.join(H.HOST_LIB_DIR, "libOpenMMAmoebaReference.so").encode()))
    ewald_tol = 7.5e-4
    w = T.amoeba_water_tile(cutoff=0.7, vdw_cutoff=0.9, polarization=H.Mutual if kind == "mutual" else H.Direct, epsilon=1e-6, ewald_tol=ewald_tol,
                            grid=(80, 80, 80), a_ewald=float(np.sqrt(-np.log(2 * ewald_tol)) / 0.7))
    system, mp, vdw = w.build()
    ctx = H.Context(system, H.Integrator(H.VERLET, 0.001), "Reference")
    ctx.setPositions(w.positions)
    t0 = time.time()
    st = ctx.getState(getForces=True, getEnergy=True)
    print("Reference platform, %s polarization: %d atoms, %.1f s, E = %.6f" % (kind, w.num_atoms, time.time() - t0, st.potentialEnergy), flush=True)
    rng = np.random.default_rng(2026)
    idx = np.sort(rng.choice(w.num_atoms, size=min(sample, w.num_atoms), replace=False)).astype(np.int32)
    f = st.forces
    out = os.path.join(ROOT, "tests", "golden", "reference_forces_amoeba_water_tile_36501_%s_sample.npz" % kind)
    np.savez_compressed(out, indices=idx, forces=f[idx], energy=st.potentialEnergy, rms_force=float(np.sqrt((f ** 2).sum(1).mean())),
                        source="tools/make_golden_amoeba_water_tile.py %s: AMOEBA Reference kernels on the Reference platform, one evaluation, "
                               "mutual epsilon 1e-6, cutoff 0.7 / vdW 0.9 nm, PME 80^3 alpha from tolerance 7.5e-4" % kind)
    print("wrote", out, os.path.getsize(out), "bytes")
    ctx.close()


if __name__ == "__main__":
    main()
