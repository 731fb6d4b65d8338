"""Extract the reference's own golden inputs for this path into tests/golden/ (run in the build container,
where /root/reference exists):

  nacl_crystal.npz / nacl_amorph.npz   particle positions of tests/nacl_crystal.dat and tests/nacl_amorph.dat
                                        (the inputs of tests/TestEwald.h:49-96 testEwaldExact and :98-220 testEwaldPME),
                                        together with the known answers asserted there.
  reference_forces_*.npz               forces/energies computed by the real Reference platform (oracle/_ref) on
                                        seeded synthetic systems -- used on the GPU box where /root/reference is absent.
"""
import os
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden")


def parse_dat(path):
    pat = re.compile(r"positions\[(\d+)\]\s*=\s*Vec3\(([^,]+),([^,]+),([^)]+)\)")
    rows = {}
    for line in open(path):
        m = pat.search(line)
        if m:
            rows[int(m.group(1))] = [float(m.group(2)), float(m.group(3)), float(m.group(4))]
    return np.array([rows[i] for i in range(len(rows))])


def main():
    os.makedirs(OUT, exist_ok=True)
    avogadro = 6.02214076e23  # AVOGADRO in SimTKOpenMMRealType.h is 6.0221367e23; the test tolerance (1e-3) covers both
    crystal = parse_dat(os.path.join(REF, "tests", "nacl_crystal.dat"))
    np.savez_compressed(os.path.join(OUT, "nacl_crystal.npz"), positions=crystal, box=2.82, cutoff=1.0, ewald_tol=1e-5,
                        madelung_energy=-(1.7476 * 1.6022e-19 * 1.6022e-19 * 6.0221367e23 * 1000) / (1.112e-10 * 0.282e-9 * 2 * 1000),
                        source="tests/TestEwald.h:49-96 testEwaldExact; tests/nacl_crystal.dat")
    amorph = parse_dat(os.path.join(REF, "tests", "nacl_amorph.dat"))
    np.savez_compressed(os.path.join(OUT, "nacl_amorph.npz"), positions=amorph, box=3.00646, cutoff=1.2, ewald_tol=1e-5,
                        gromacs_energy=-3.82047e5, source="tests/TestEwald.h:98-220 testEwaldPME; tests/nacl_amorph.dat")
    print("nacl fixtures:", crystal.shape, amorph.shape)

    # Reference-platform outputs on seeded synthetic systems
    from openmm_amd import harness as H, testsystems as T
    for name, w in (("water648_pme", T.water_box(6, seed=11)), ("water3000_pme", T.water_box(10, seed=12)), ("argon864_nocutoff", T.argon_box())):
        if w.method == H.PME:
            alpha = float(np.sqrt(-np.log(2 * w.ewald_tol)) / w.cutoff)
            n = {"water648_pme": 18, "water3000_pme": 32}[name]
            w.pme_params = (alpha, n, n, n)
        system, nb = w.build()
        if w.method == H.PME:
            nb.setReciprocalSpaceForceGroup(1)
        ctx = H.Context(system, H.Integrator(H.VERLET, 0.001), "Reference")
        ctx.setPositions(w.positions)
        full = ctx.getState(getForces=True, getEnergy=True)
        out = dict(positions=w.positions, forces=full.forces, energy=full.potentialEnergy)
        if w.method == H.PME:
            d = ctx.getState(getForces=True, getEnergy=True, groups=1)
            r = ctx.getState(getForces=True, getEnergy=True, groups=2)
            out.update(direct_forces=d.forces, direct_energy=d.potentialEnergy, recip_forces=r.forces, recip_energy=r.potentialEnergy,
                       pme_params=np.array(w.pme_params))
        np.savez_compressed(os.path.join(OUT, "reference_forces_%s.npz" % name), **out)
        print(name, "E =", full.potentialEnergy)
        ctx.close()


if __name__ == "__main__":
    main()
