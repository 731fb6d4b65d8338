"""The amoebapme test of examples/benchmark.py (BASELINE.json configs[4]) as fixtures:

  tests/golden/amoeba_dhfr_5dfr_amoeba2009.npz          the System description (openmm_amd/forcefield_amoeba.py reading amoeba2009.xml and
                                                        5dfr_solv-cube_equil.pdb of the reference tree)
  tests/golden/reference_forces_amoeba_dhfr.npz         forces and energies of the three parts at the PDB coordinates from the Reference
                                                        platform (the reference's own kernels: openmm_host/Makefile): all valence terms,
                                                        AmoebaVdwForce, AmoebaMultipoleForce (mutual, epsilon 1e-6: the bar of the comparison
                                                        is the kernel, not the solver), the PME grid pinned to the 64^3 both platforms choose

    python tools/make_amoeba_dhfr_fixture.py

Runs in the build container only (reads /root/reference, needs build/openmm/lib); about 3 minutes, nearly all of it the Reference multipole kernel.
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from openmm_amd import forcefield_amoeba as A, harness as H, testsystems as T
    golden = os.path.join(ROOT, "tests", "golden")
    d = A.dhfr()
    path = os.path.join(golden, "amoeba_dhfr_5dfr_amoeba2009.npz")
    A.save_description(d, path)
    print("wrote", path, os.path.getsize(path), "bytes")
    d = A.load_description(path)
    H.lib()
    H._check(H.lib().omm_load_plugin(os.path.join(H.HOST_LIB_DIR, "libOpenMMAmoebaReference.so").encode()))
    w = T.amoeba_dhfr(epsilon=1e-6, pin_grid=True)
    system, mp, vdw = w.build()
    H.lib().omm_force_set_group(vdw.h, 2)
    ctx = H.Context(system, H.Integrator(H.VERLET, 0.001), "Reference")
    ctx.setPositions(w.positions)
    out = {}
    for name, group in (("valence", 1), ("vdw", 4), ("multipole", 2)):
        t0 = time.time()
        st = ctx.getState(getForces=True, getEnergy=True, groups=group)
        print("%-10s E = %.6f kJ/mol, rms force %.3f, %.1f s" % (name, st.potentialEnergy, np.sqrt((st.forces ** 2).sum(1).mean()), time.time() - t0), flush=True)
        out["forces_" + name] = st.forces.astype(np.float32)
        out["energy_" + name] = st.potentialEnergy
        out["rms_force_" + name] = float(np.sqrt((st.forces ** 2).sum(1).mean()))
    path = os.path.join(golden, "reference_forces_amoeba_dhfr.npz")
    np.savez_compressed(path, source="tools/make_amoeba_dhfr_fixture.py: Reference platform (reference kernels), PDB coordinates, mutual epsilon 1e-6, cutoff 0.7 / vdW 0.9 nm, "
                                     "PME 64^3, alpha from tolerance 7.5e-4; forces stored as float32", **out)
    print("wrote", path, os.path.getsize(path), "bytes")
    ctx.close()


if __name__ == "__main__":
    main()
