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


def _reference_forces(system, positions):
    ctx = H.Context(system, H.Integrator(H.VERLET, 0.001), "Reference")
    ctx.setPositions(positions)
    st = ctx.getState(getForces=True, getEnergy=True)
    ctx.close()
    return st


@pytest.mark.skipif(not os.path.isdir(REF_DATA), reason="needs the reference's force-field files and test PDB (build container only)")
def test_reader_is_pinned_to_the_reference_held_lysozyme_forces(golden):
    """SURVEY.md §8(f)2 / VERDICT r4 "missing" 3: T4 lysozyme built by openmm_amd/forcefield.py from amber99sb.xml + amber99_obc.xml with
    createSystem's defaults (NoCutoff, no constraints, GBSA-OBC), evaluated on the Reference platform, against the golden forces the
    reference's own Python test keeps (wrappers/python/tests/systems/lysozyme-implicit-forces.xml, copied into
    tests/golden/forcefield_reference_forces.npz by tools/make_forcefield_goldens.py) under that test's own criterion
    (TestForceField.py:296-301): fewer than N/20 atoms may differ by more than 0.1 kJ/mol/nm AND 1e-3 relative."""
    from openmm_amd import forcefield as FF
    w = FF.lysozyme_implicit()
    g = golden("forcefield_reference_forces.npz")
    assert w.num_atoms == 2603 and np.abs(w.positions - g["lysozyme_positions"]).max() < 1e-12
    assert {"NMET", "CLEU", "HIE"} & set(w.template_names) and w.gbsa is not None and len(w.constraints[0]) == 0
    system, nb = w.build()
    f = _reference_forces(system, w.positions).forces
    ref = g["lysozyme_forces"]
    diff = np.linalg.norm(f - ref, axis=1)
    differences = int(((diff > 0.1) & (diff / np.linalg.norm(f, axis=1) > 1e-3)).sum())
    print("lysozyme: %d atoms differ (allowed < %d); median |dF| %.3f, 99th percentile %.2f kJ/mol/nm at an RMS force of %.0f"
          % (differences, w.num_atoms // 20, np.median(diff), np.percentile(diff, 99), np.sqrt((ref ** 2).sum(1).mean())))
    assert differences < w.num_atoms / 20
    assert np.median(diff) < 0.1 and np.percentile(diff, 98) < 1.0          # tighter than the reference asks: 98 % of the atoms within 1e-3 of the RMS force


def test_system_xml_is_what_the_reference_serializer_writes_and_loads():
    """§8(f)2 deliverable: openmm_amd/system_xml.py writes the System as XML.  (a) `XmlSerializer::deserialize<System>` of the reference loads it
    and the loaded System gives the Reference platform's forces and energy of the System built call by call -- to the last bit; (b) the text
    equals, line for line, what `XmlSerializer::serialize` writes for that System (every exception of createExceptionsFromBonds in the
    reference's order), for the small solvated chain and for the 23 558-atom DHFR benchmark System."""
    from openmm_amd import system_xml as X
    w = T.small_solvated_chain(seed=3)
    w.pme_params = (3.0, 24, 24, 24)
    text = X.workload_to_xml(w)
    built, _ = w.build()
    loaded = H.System.from_xml(text)
    assert loaded.getNumParticles() == w.num_atoms and loaded.getNumForces() == built.getNumForces() == 5
    assert loaded.getNumConstraints() == len(w.constraints[0])
    a, b = _reference_forces(built, w.positions), _reference_forces(loaded, w.positions)
    assert np.array_equal(a.forces, b.forces) and a.potentialEnergy == b.potentialEnergy

    def same_text(w):
        system, _ = w.build()
        ours = [l.strip() for l in X.workload_to_xml(w).splitlines()]
        theirs = [l.strip() for l in system.to_xml().splitlines()]
        assert len(ours) == len(theirs)
        return [(x, y) for x, y in zip(ours, theirs) if x != y and not x.startswith("<System openmmVersion")]
    assert same_text(w) == []
    d = T.dhfr()
    assert same_text(d) == []
    # GBSA-OBC section (lysozyme-style System): loads, and carries the reaction-field switch-off of GBSAOBCGenerator.postprocessSystem
    g = T.water_box(3, seed=1, method=H.NoCutoff)
    g.box = None
    g.gbsa = (g.charge, np.full(g.num_atoms, 0.15), np.full(g.num_atoms, 0.8))
    g.reaction_field_dielectric = 1.0
    assert same_text(g) == []
    built, _ = g.build()
    a, b = _reference_forces(built, g.positions), _reference_forces(H.System.from_xml(X.workload_to_xml(g)), g.positions)
    assert np.array_equal(a.forces, b.forces)
