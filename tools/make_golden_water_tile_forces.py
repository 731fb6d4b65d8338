"""Reference-platform forces of the equilibrated water tile (tests/golden/water_tile_36501_equilibrated.npz) as ONE periodic box,
PME grid 64^3 with the alpha of the 0.9 nm cutoff.  The bench's 1M-atom workload is 3 x 3 x 3 copies of this tile on a 192^3 grid:
the same charge density on the same mesh spacing, so every copy of an atom must feel the force computed here
(tests/test_gpu_platform.py::test_water1m_tiled_forces_match_reference_of_the_tile).  Runs in the build container (oracle/_ref).

    python tools/make_golden_water_tile_forces.py [sample=12000]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from openmm_amd import harness as H, testsystems as T
    sample = int(sys.argv[1]) if len(sys.argv) > 1 else 12000
    w = T.water_tiled(1)
    grid = 64
    w.pme_params = (float(np.sqrt(-np.log(2 * w.ewald_tol)) / w.cutoff), grid, grid, grid)
    system, nb = w.build()
    ctx = H.Context(system, H.Integrator(H.VERLET, 0.001), "Reference")
    ctx.setPositions(w.positions)
    t0 = time.time()
    st = ctx.getState(getForces=True, getEnergy=True)
    print("Reference platform: %d atoms, %.1f s, E = %.6f" % (w.num_atoms, time.time() - t0, st.potentialEnergy), flush=True)
    alpha, nx, ny, nz = nb.getPMEParametersInContext(ctx)
    rng = np.random.default_rng(2025)
    idx = np.sort(rng.choice(w.num_atoms, size=min(sample, w.num_atoms), replace=False)).astype(np.int32)
    f = st.forces
    out = os.path.join(ROOT, "tests", "golden", "reference_forces_water_tile_36501_sample.npz")
    np.savez_compressed(out, indices=idx, forces=f[idx], energy=st.potentialEnergy, rms_force=float(np.sqrt((f ** 2).sum(1).mean())),
                        pme=np.array([alpha, nx, ny, nz]), source="tools/make_golden_water_tile_forces.py: Reference platform of oracle/_ref, one evaluation")
    print("wrote", out, os.path.getsize(out), "bytes")
    ctx.close()


if __name__ == "__main__":
    main()
