"""Golden forces of BASELINE.json configs[3] (the 985 527-atom TIP3P box of bench.py --workload water1m) from the real
Reference platform (oracle/_ref): ONE force evaluation, run in the build container (minutes, several GB).  The full
force array is 24 MB, so the fixture keeps a seeded sample of atoms (indices + float64 forces), the potential energy and
checksums of the generated positions (the GPU test regenerates them from the same seed and verifies the checksums).

    python tools/make_golden_water1m.py [n_side=69] [sample=40000]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from openmm_amd import harness as H, testsystems as T
    n_side = int(sys.argv[1]) if len(sys.argv) > 1 else 69
    sample = int(sys.argv[2]) if len(sys.argv) > 2 else 40000
    platform = sys.argv[3] if len(sys.argv) > 3 else "Reference"
    w = T.water_box(n_side, seed=1)
    # The Reference platform takes the PME grid as NonbondedForceImpl::calcPMEParameters gives it (191^3 here); the HIP platform
    # rounds up to an FFT-friendly size (192^3).  Parity is about the arithmetic, so both sides get the same explicit grid.
    grid = int(sys.argv[4]) if len(sys.argv) > 4 else 192
    w.pme_params = (float(np.sqrt(-np.log(2 * w.ewald_tol)) / w.cutoff), grid, grid, grid)
    if platform == "CPU":
        H.load_cpu_platform()
    system, nb = w.build()
    ctx = H.Context(system, H.Integrator(H.VERLET, 0.001), platform)
    ctx.setPositions(w.positions)
    t0 = time.time()
    st = ctx.getState(getForces=True, getEnergy=True)
    print("%s platform: %d atoms, %.1f s, E = %.6f" % (platform, w.num_atoms, time.time() - t0, st.potentialEnergy), flush=True)
    alpha, nx, ny, nz = nb.getPMEParametersInContext(ctx)
    rng = np.random.default_rng(2024)
    idx = np.sort(rng.choice(w.num_atoms, size=min(sample, w.num_atoms), replace=False)).astype(np.int32)
    f = st.forces
    out = os.path.join(ROOT, "tests", "golden", "reference_forces_water%d_sample.npz" % w.num_atoms)
    np.savez_compressed(out, n_side=n_side, seed=1, indices=idx, forces=f[idx], energy=st.potentialEnergy,
                        rms_force=float(np.sqrt((f ** 2).sum(1).mean())), max_force=float(np.linalg.norm(f, axis=1).max()),
                        position_sum=w.positions.sum(0), position_sample=w.positions[idx[:64]], box=w.box,
                        pme=np.array([alpha, nx, ny, nz]), platform=platform,
                        source="tools/make_golden_water1m.py: %s platform of oracle/_ref, one evaluation" % platform)
    print("wrote", out, os.path.getsize(out), "bytes")
    ctx.close()


if __name__ == "__main__":
    main()
