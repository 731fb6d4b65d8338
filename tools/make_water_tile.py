"""Generates tests/golden/water_tile_36501_equilibrated.npz: 23^3 TIP3P waters (rigid, PME, 0.9 nm cutoff, 33.4 molecules / nm^3)
equilibrated at 300 K on the HIP platform -- the tile the bench's 1M-atom workload is made of (3 x 3 x 3 copies = the same 985 527
atoms in the same 21.4 nm box as `water_box(69)`, but a liquid at 300 K instead of a lattice that melts at 1700 K).

Runs on a GPU box:   python tools/make_water_tile.py gpurun_out/water_tile_36501_equilibrated.npz
Protocol: lattice start (testsystems.water_box(23, seed=1)); 20 x (50 steps, velocities redrawn at 300 K) to take the heat of the
melting lattice out; 30 000 steps (60 ps) of LangevinMiddle at 300 K, 1 / ps, 2 fs.  Positions are stored as float32 with every
molecule's oxygen wrapped into the box, velocities as float16 (the loader re-applies the constraints).
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from openmm_amd import harness as H, testsystems as T  # noqa: E402


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/water_tile_36501_equilibrated.npz"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30000
    H.load_hip_platform(emulated=False)
    w = T.water_box(23, seed=1)
    system, nb = w.build()
    integ = H.Integrator(H.LANGEVIN_MIDDLE, 0.002, 300.0, 1.0, seed=11, constraintTolerance=1e-6)
    ctx = H.Context(system, integ, "HIP", {})
    ctx.setPositions(w.positions)
    ctx.applyConstraints(1e-6)
    n = system.getNumParticles()
    ndof = 3 * n - n - 3

    def temperature(st):
        return 2.0 * st.kineticEnergy / (0.0083144626 * ndof)

    for k in range(20):
        ctx.setVelocitiesToTemperature(300.0, 100 + k)
        integ.step(50)
    st = ctx.getState(getEnergy=True)
    print("after the quench: T = %.0f K, U = %.0f kJ/mol" % (temperature(st), st.potentialEnergy))
    for k in range(6):
        integ.step(steps // 6)
        st = ctx.getState(getEnergy=True)
        print("  %6d steps: T = %.1f K, U = %.0f kJ/mol (%.2f per molecule)" % ((k + 1) * (steps // 6), temperature(st), st.potentialEnergy, st.potentialEnergy / (n // 3)))
    st = ctx.getState(getPositions=True, getVelocities=True, getEnergy=True)
    L = float(w.box[0][0])
    pos = st.positions.reshape(-1, 3, 3)
    shift = np.floor(pos[:, 0, :] / L) * L
    pos = (pos - shift[:, None, :]).reshape(-1, 3)
    np.savez_compressed(out, positions=pos.astype(np.float32), velocities=st.velocities.astype(np.float16), box=np.float64(L), n_side=np.int32(23),
                        temperature=np.float64(temperature(st)), potential_energy=np.float64(st.potentialEnergy), steps=np.int32(steps))
    print("wrote %s (%d bytes)" % (out, os.path.getsize(out)))


if __name__ == "__main__":
    main()
