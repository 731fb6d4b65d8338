"""tests/golden/dhfr_5dfr_amber99sb_tip3p.npz: the System of examples/benchmark.py's `pme` test (DHFR, 23 558 atoms, amber99sb +
tip3p, PME 0.9 nm, HBonds, rigid water) as plain arrays, built by openmm_amd/forcefield.py from the reference's own input files
(examples/5dfr_solv-cube_equil.pdb, wrappers/python/openmm/app/data/{amber99sb,tip3p}.xml), plus positions/velocities after a short
LangevinMiddle run at 300 K on the reference's CPU platform (oracle/_ref) so that benchmarks start from an equilibrated state.
Run in the build container (needs /root/reference); the GPU box only reads the fixture (openmm_amd/testsystems.py:dhfr).

    python tools/make_dhfr_fixture.py [equilibration steps = 1500]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from openmm_amd import forcefield as FF, harness as H
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
    w = FF.dhfr()
    H.load_cpu_platform()
    system, nb = w.build()
    integ = H.Integrator(H.LANGEVIN_MIDDLE, 0.002, 300.0, 1.0, seed=1, constraintTolerance=1e-6)
    ctx = H.Context(system, integ, "CPU")
    ctx.setPositions(w.positions)
    ctx.applyConstraints(1e-6)
    e0 = ctx.getState(getEnergy=True).potentialEnergy
    ctx.setVelocitiesToTemperature(300.0, 1)
    integ.step(steps)
    st = ctx.getState(getPositions=True, getVelocities=True, getEnergy=True)
    print("E(pdb) = %.1f, after %d steps E = %.1f, KE = %.1f" % (e0, steps, st.potentialEnergy, st.kineticEnergy))
    out = os.path.join(ROOT, "tests", "golden", "dhfr_5dfr_amber99sb_tip3p.npz")
    np.savez_compressed(out, name=w.name, box=w.box, masses=w.masses.astype(np.float32), charge=w.charge.astype(np.float32),
                        sigma=w.sigma, epsilon=w.epsilon, exception_bonds=w.exception_bonds.astype(np.int32),
                        coulomb14=w.coulomb14, lj14=w.lj14, cutoff=w.cutoff,
                        bond_atoms=w.bonds[0].astype(np.int32), bond_length=w.bonds[1], bond_k=w.bonds[2],
                        angle_atoms=w.angles[0].astype(np.int32), angle_theta=w.angles[1], angle_k=w.angles[2],
                        torsion_atoms=w.torsions[0].astype(np.int32), torsion_n=w.torsions[1], torsion_phase=w.torsions[2], torsion_k=w.torsions[3],
                        constraint_atoms=w.constraints[0].astype(np.int32), constraint_length=w.constraints[1],
                        pdb_positions=w.positions.astype(np.float32), positions=st.positions, velocities=st.velocities.astype(np.float32),
                        pdb_potential_energy=e0,
                        source="tools/make_dhfr_fixture.py: openmm_amd/forcefield.py on examples/5dfr_solv-cube_equil.pdb + amber99sb.xml + tip3p.xml; "
                               "%d LangevinMiddle steps (2 fs, 300 K) on platforms/cpu of oracle/_ref" % steps)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
