"""What does the REFERENCE platform itself show at the benchmark's mutualInducedTargetEpsilon = 1e-5 D?  amoeba2009 DHFR (BASELINE.json
configs[4]) on the Reference platform solved to 1e-5 D, against the committed golden of the same platform solved to 1e-6 D
(tests/golden/reference_forces_amoeba_dhfr.npz): the truncation of the induced-dipole solve as a force error, in the units of
bench.py's force_parity (|dF| / max(|F_ref|, F_rms) per atom).  ~3 minutes on 8 cores; run in the build container.

    python tools/diag_reference_solver_truncation.py > profiles/r11/reference_platform_at_run_epsilon.txt
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from openmm_amd import harness as H, testsystems as T
    H.lib()
    H._check(H.lib().omm_load_plugin(os.path.join(H.HOST_LIB_DIR, "libOpenMMAmoebaReference.so").encode()))
    g = np.load(os.path.join(ROOT, "tests", "golden", "reference_forces_amoeba_dhfr.npz"))
    ref = g["forces_multipole"].astype(np.float64)
    rms = float(np.sqrt((ref ** 2).sum(1).mean()))
    for eps in (1e-5,):
        w = T.amoeba_dhfr(epsilon=eps, pin_grid=True)
        system, mp, vdw = w.build()
        H.lib().omm_force_set_group(vdw.h, 2)
        ctx = H.Context(system, H.Integrator(H.VERLET, 0.001), "Reference")
        ctx.setPositions(w.positions)
        t0 = time.time()
        f = ctx.getState(getForces=True, groups=2).forces
        rel = np.linalg.norm(f - ref, axis=1) / np.maximum(np.linalg.norm(ref, axis=1), rms)
        print("Reference platform, AmoebaMultipoleForce of amoeba2009 DHFR (23 558 atoms, PME 64^3), mutual epsilon %g against the same platform at 1e-6 (golden, float32-stored):" % eps)
        print("  max rel err %.3e   99.9th percentile %.3e   median %.3e   atoms above 1e-4: %d   above 5e-5: %d   (%.0f s)" % (
            rel.max(), np.percentile(rel, 99.9), np.median(rel), int((rel > 1e-4).sum()), int((rel > 5e-5).sum()), time.time() - t0), flush=True)
        ctx.close()


if __name__ == "__main__":
    main()
