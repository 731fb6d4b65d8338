"""Native LJPME: dispersion grid + direct-space correction + exclusion terms against the Reference platform
(ReferenceLJCoulombIxn.cpp:224-260,407-435,505-520; ReferencePME.cpp:518-614), with explicit grids on both sides (the Reference
platform keeps sizes the HIP platform rounds up to FFT-friendly ones).  Shared by the emulator test and the GPU test."""
import numpy as np

from openmm_amd import harness as H, testsystems as T


def run_ljpme_case(n_side=6):
    w = T.water_box(n_side, seed=11)
    w.method = H.LJPME
    # hydrogens get LJ parameters too, so that excluded pairs carry a C6 product
    w.sigma = np.where(w.epsilon > 0, w.sigma, 0.11)
    w.epsilon = np.where(w.epsilon > 0, w.epsilon, 0.07)
    alpha = float(np.sqrt(-np.log(2 * w.ewald_tol)) / w.cutoff)
    w.pme_params = (alpha, 18, 18, 18)
    w.ljpme_params = (2.6, 20, 18, 24)
    out = {}
    for plat in ("Reference", "HIP"):
        s, nb = w.build()
        nb.setReciprocalSpaceForceGroup(1)
        c = H.Context(s, H.Integrator(H.VERLET, 0.001), plat)
        c.setPositions(w.positions)
        out[plat] = [c.getState(getForces=True, getEnergy=True, groups=g) for g in (1, 2, 3)]
        if plat == "HIP":
            assert nb.getLJPMEParametersInContext(c) == (2.6, 20, 18, 24)
            assert c.getPlatformName() == "HIP"
        c.close()
    rms = np.sqrt((out["Reference"][2].forces ** 2).sum(1).mean())
    for k, name in enumerate(("direct", "reciprocal", "total")):
        ref, hip = out["Reference"][k], out["HIP"][k]
        err = np.abs(hip.forces - ref.forces).max() / rms
        print("LJPME %s: force max diff / rms %.3g, E %.6f vs %.6f" % (name, err, hip.potentialEnergy, ref.potentialEnergy))
        assert err < 5e-5
        assert abs(hip.potentialEnergy - ref.potentialEnergy) < 2e-6 * max(abs(ref.potentialEnergy), 5e4)
