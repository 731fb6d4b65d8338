"""How far do atoms travel along x between two re-sorts?  The domain decomposition's halo carries a drift margin (DESIGN (e));
this prints, for the bench workloads on ONE GPU, the largest displacement of any unit's first atom over windows of the length an
order lives (re-sort interval + lag), from the very start of the run (the lattice of `water1m` melts there) and later.

    python tools/check_drift.py [workload] [window steps] [windows]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from openmm_amd import harness as H  # noqa: E402


def main():
    workload = sys.argv[1] if len(sys.argv) > 1 else "water1m"
    window = int(sys.argv[2]) if len(sys.argv) > 2 else 628
    windows = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    H.load_hip_platform(emulated=False)
    w = bench.make_workload(workload, seed=1)
    system, nb, integ, ctx = bench.start_platform(w, "HIP", 0.002, 0, {}, seed=1, prepare=0)
    n = system.getNumParticles()
    st = ctx.getState(getPositions=True, getEnergy=True)
    L = np.diag(np.asarray(w.box, dtype=float)) if np.ndim(w.box) == 2 else np.asarray(w.box, dtype=float)
    ndof = 3 * n - (n if workload.startswith("water") else 0) - 3              # rigid waters: 3 constraints per molecule of 3 atoms
    ref = st.positions.copy()
    print("%s: %d atoms; window %d steps" % (workload, n, window))
    done = 0
    for k in range(windows):
        sub = [window // 4] * 3 + [window - 3 * (window // 4)]
        for s in sub:
            integ.step(s)
            done += s
            st = ctx.getState(getPositions=True, getEnergy=True)
            d = st.positions - ref
            d -= np.round(d / L) * L          # molecules may have been re-wrapped by a whole box vector in between
            ax = np.abs(d[:, 0])
            print("  step %5d (order age %4d): max |dx| %.3f nm, 99.99th pct %.3f nm, RMS %.4f nm, max |d| %.3f nm; T = %.0f K" % (
                done, done - k * window, ax.max(), np.quantile(ax, 0.9999), np.sqrt((d[:, 0] ** 2).mean()), np.sqrt((d ** 2).sum(axis=1)).max(),
                2.0 * st.kineticEnergy / (0.0083144626 * ndof)))
        ref = st.positions.copy()


if __name__ == "__main__":
    main()
