"""openmm_amd/forcefield.py + the DHFR fixture: the benchmark System of examples/benchmark.py without the SWIG layer."""
import os

import numpy as np
import pytest

from openmm_amd import harness as H, testsystems as T

REF_DATA = "/root/reference/wrappers/python/openmm/app/data"


def test_dhfr_fixture_is_the_benchmark_system():
    w = T.dhfr()
    assert w.num_atoms == 23558
    assert np.allclose(np.diag(w.box), 6.223)
    assert abs(w.charge.sum() + 11.0) < 1e-3                       # DHFR without counter-ions
    n_water = int((np.abs(w.charge + 0.834) < 1e-6).sum())
    assert n_water == 7023
    assert len(w.constraints[0]) == 3 * n_water + 1221              # rigid waters + the protein's 1221 bonds to hydrogen
    # HBonds: no bond term involves a hydrogen; waters carry no bond or angle terms
    light = w.masses < 1.5
    assert not light[w.bonds[0]].any()
    assert w.bonds[0].max() < 2489 and w.angles[0].max() < 2489 and w.torsions[0].max() < 2489
    assert abs(w.coulomb14 - 1 / 1.2) < 1e-6 and w.lj14 == 0.5
    # constraints are satisfied by the equilibrated coordinates
    pairs, dist = w.constraints
    d = np.linalg.norm(w.positions[pairs[:, 0]] - w.positions[pairs[:, 1]], axis=1)
    assert np.abs(d - dist).max() < 1e-5


def test_dhfr_energy_on_the_reference_platform_matches_the_generator():
    """One evaluation of the fixture's System at the PDB coordinates on the Reference platform reproduces the energy the generator
    recorded (platforms/cpu, single-precision pair arithmetic): the arrays in the fixture describe the System that was run."""
    w = T.dhfr()
    system, nb = w.build()
    ctx = H.Context(system, H.Integrator(H.VERLET, 0.001), "Reference")
    ctx.setPositions(w.pdb_positions)
    ctx.applyConstraints(1e-6)
    e = ctx.getState(getEnergy=True).potentialEnergy
    ctx.close()
    assert abs(e - w.pdb_potential_energy) < 2e-4 * abs(e), (e, w.pdb_potential_energy)
    assert -3.1e5 < e < -2.9e5


@pytest.mark.skipif(not os.path.isdir(REF_DATA), reason="needs the reference's force-field files (build container only)")
def test_reader_reproduces_the_fixture_from_the_reference_files():
    from openmm_amd import forcefield as FF
    w, f = FF.dhfr(), T.dhfr()
    assert set(w.template_names) >= {"NMET", "CARG", "HID", "HOH"}             # terminal variants and the histidine tautomer are found by graph matching
    assert np.array_equal(w.exception_bonds, f.exception_bonds)
    assert np.allclose(w.charge, f.charge, atol=1e-6) and np.allclose(w.sigma, f.sigma) and np.allclose(w.epsilon, f.epsilon)
    for a, b in ((w.bonds, f.bonds), (w.angles, f.angles), (w.torsions, f.torsions), (w.constraints, f.constraints)):
        assert all(np.allclose(x, y) for x, y in zip(a, b))
    # spot checks against amber99sb.xml / tip3p.xml: backbone N-H constraint length, a peptide improper, TIP3P geometry
    assert np.isclose(f.constraints[1][-1], 0.15139, atol=1e-5)                # H-H of the last water: 2 * 0.09572 * sin(104.52 deg / 2)
    k = dict(zip(map(tuple, f.torsions[0]), f.torsions[3]))
    assert any(abs(v - 4.6024) < 1e-9 for v in k.values()) and any(abs(v - 43.932) < 1e-9 for v in k.values())   # amber general / peptide impropers
