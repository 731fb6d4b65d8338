"""openmm_amd/forcefield_amoeba.py: amoeba2009.xml + the DHFR PDB -> the System of examples/benchmark.py's `amoebapme` test (BASELINE.json
configs[4]).  The SWIG Python layer of the reference cannot be built here, so the reader is pinned three ways: the parameters it assigns to
water against the numbers of the reference's own C++ test (plugins/amoeba/tests/TestAmoebaMultipoleForce.h:1180-1217, AMOEBA_WATER in
testsystems.py), structural invariants of the force field, and the Reference platform's energies of the System it describes against the
committed golden values (tools/make_amoeba_dhfr_fixture.py)."""
import os

import numpy as np
import pytest

from conftest import ROOT

GOLDEN = os.path.join(ROOT, "tests", "golden")
XML = "/root/reference/wrappers/python/openmm/app/data/amoeba2009.xml"
PDB = "/root/reference/examples/5dfr_solv-cube_equil.pdb"
PROTEIN_ATOMS = 2489


@pytest.fixture(scope="module")
def description():
    from openmm_amd import forcefield_amoeba as A
    return A.load_description(os.path.join(GOLDEN, "amoeba_dhfr_5dfr_amoeba2009.npz"))


@pytest.mark.skipif(not (os.path.exists(XML) and os.path.exists(PDB)), reason="reference tree not present")
def test_fixture_is_what_the_reader_makes_of_the_reference_files(description):
    from openmm_amd import forcefield_amoeba as A
    d = A.dhfr()
    for key in ("bonds", "angles", "inplane_angles", "opbends", "stretch_bends", "urey_bradleys", "torsions", "pi_torsions"):
        for a, b in zip(d[key][:2], description[key][:2]):
            assert np.array_equal(np.asarray(a), np.asarray(b)), key
    assert np.array_equal(d["torsion_torsions"][0], description["torsion_torsions"][0])
    for k in ("charge", "dipole", "quadrupole", "axes", "thole", "damping", "polarity"):
        assert np.array_equal(d["multipoles"][k], description["multipoles"][k]), k
    assert all(a[0] == b[0] and a[1] == b[1] and list(a[2]) == list(b[2]) for a, b in zip(d["multipoles"]["covalent_maps"], description["multipoles"]["covalent_maps"]))
    assert all(list(a) == list(b) for a, b in zip(d["vdw"]["exclusions"], description["vdw"]["exclusions"]))


def test_water_gets_the_parameters_of_the_reference_test(description):
    from openmm_amd import forcefield_amoeba as A, testsystems as T
    d, a = description, T.AMOEBA_WATER
    n = len(d["masses"])
    assert n == 23558 and (n - PROTEIN_ATOMS) % 3 == 0
    o = np.arange(PROTEIN_ATOMS, n, 3)
    m = d["multipoles"]
    assert np.allclose(m["charge"][o], a["qO"], atol=0, rtol=1e-12) and np.allclose(m["charge"][o + 1], a["qH"]) and np.allclose(m["charge"][o + 2], a["qH"])
    assert np.allclose(m["dipole"][o], a["dO"], rtol=1e-12, atol=0) and np.allclose(m["dipole"][o + 1], a["dH"], rtol=1e-12, atol=0)
    assert np.allclose(m["quadrupole"][o].reshape(-1, 3, 3), a["QO"], rtol=1e-12, atol=0) and np.allclose(m["quadrupole"][o + 2].reshape(-1, 3, 3), a["QH"], rtol=1e-12, atol=0)
    assert np.allclose(m["polarity"][o], a["polO"]) and np.allclose(m["polarity"][o + 1], a["polH"]) and np.allclose(m["thole"][o], a["thole"])
    assert np.allclose(m["damping"][o], a["polO"] ** (1 / 6.0))
    # frames: the oxygen bisects its hydrogens, a hydrogen looks at the oxygen and then at the other hydrogen
    assert (m["axes"][o, 0] == A.Bisector).all() and (np.sort(m["axes"][o, 1:3], axis=1) == np.stack([o + 1, o + 2], -1)).all()
    assert (m["axes"][o + 1] == np.stack([np.full(len(o), A.ZThenX), o, o + 2, np.full(len(o), -1)], -1)).all()
    v = d["vdw"]
    assert np.allclose(v["sigma"][o], a["sigO"]) and np.allclose(v["sigma"][o + 1], a["sigH"]) and np.allclose(v["epsilon"][o], a["epsO"]) and np.allclose(v["epsilon"][o + 2], a["epsH"])
    assert np.allclose(v["reduction"][o + 1], a["redH"]) and (v["parent"][o + 1] == o).all() and (v["parent"][o] == o).all()
    assert (v["sigma_rule"], v["epsilon_rule"]) == ("CUBIC-MEAN", "HHG")
    # valence: two bonds, one angle and one Urey-Bradley term per water
    ba, bp = d["bonds"][:2]
    w = ba[:, 0] >= PROTEIN_ATOMS
    assert w.sum() == 2 * len(o) and np.allclose(bp[w, 0], a["dOH"]) and np.allclose(2 * bp[w, 1], a["kBond"])
    aa, ap = d["angles"][:2]
    w = aa[:, 1] >= PROTEIN_ATOMS
    assert w.sum() == len(o) and np.allclose(np.deg2rad(ap[w, 0]), a["angle"])
    assert (d["urey_bradleys"][0][:, 0] >= PROTEIN_ATOMS).all() and len(d["urey_bradleys"][0]) == len(o)


def test_structure_of_the_force_field(description):
    from openmm_amd import forcefield_amoeba as A
    d = description
    n = len(d["masses"])
    m = d["multipoles"]
    assert abs(m["charge"].sum() - round(m["charge"].sum())) < 1e-9 and round(m["charge"].sum()) == -11          # DHFR at pH 7 in this file, no counter-ions
    assert np.allclose(np.trace(m["quadrupole"].reshape(-1, 3, 3), axis1=1, axis2=2), 0, atol=1e-9)               # traceless
    assert np.allclose(m["quadrupole"].reshape(-1, 3, 3), m["quadrupole"].reshape(-1, 3, 3).transpose(0, 2, 1))
    # every frame atom is a 1-2 or 1-3 neighbour of its atom
    maps = {}
    for atom, kind, l in m["covalent_maps"]:
        maps[(atom, kind)] = set(int(x) for x in l)
    for i in range(n):
        near = maps[(i, A.Covalent12)] | maps[(i, A.Covalent13)]
        assert all(a < 0 or int(a) in near for a in m["axes"][i, 1:]), i
        assert i in maps[(i, A.PolarizationCovalent11)]
        for j in maps[(i, A.PolarizationCovalent11)]:
            assert maps[(j, A.PolarizationCovalent11)] == maps[(i, A.PolarizationCovalent11)]      # a partition
        for kind in (A.Covalent12, A.Covalent13, A.Covalent14, A.Covalent15):
            assert all(i in maps[(j, kind)] for j in maps[(i, kind)])                              # symmetric
    # shells are disjoint
    for i in range(0, PROTEIN_ATOMS, 7):
        shells = [maps[(i, k)] for k in (A.Covalent12, A.Covalent13, A.Covalent14, A.Covalent15)]
        assert sum(len(s) for s in shells) == len(set().union(*shells))
    # three out-of-plane bends and three in-plane angles at every trigonal centre that has any
    centres, counts = np.unique(d["opbends"][0][:, 1], return_counts=True)
    assert (counts == 3).all()
    c2, n2 = np.unique(d["inplane_angles"][0][:, 1], return_counts=True)
    assert set(c2) <= set(centres) and (n2 <= 3).all()
    assert not set(d["angles"][0][:, 1]) & set(centres)
    # a backbone of 159 residues: one torsion-torsion (phi, psi) map per residue that has both neighbours, none for glycine / proline specials beyond the table
    assert 100 < len(d["torsion_torsions"][0]) <= 157
    grid = d["torsion_torsions"][2][0]
    assert grid.shape == (25, 25, 3) and grid[0, 0, 0] == -180 and grid[1, 0, 0] == -165 and grid[0, 1, 1] == -165


def test_reference_platform_reproduces_the_golden_valence_energy_and_forces(description):
    """The same fixture through the harness -> the Reference platform's energy of all valence terms (Custom*Forces with the reference's
    expressions, AmoebaTorsionTorsionForce, ...) as committed with the golden forces."""
    from openmm_amd import harness as H, testsystems as T
    if not os.path.exists(os.path.join(H.HOST_LIB_DIR, "libOpenMMAmoebaReference.so")):
        pytest.skip("host OpenMM not built")
    H.lib()
    H._check(H.lib().omm_load_plugin(os.path.join(H.HOST_LIB_DIR, "libOpenMMAmoebaReference.so").encode()))
    g = np.load(os.path.join(GOLDEN, "reference_forces_amoeba_dhfr.npz"))
    w = T.amoeba_dhfr()
    s, mp, vdw = w.build(nonbonded=False)
    ctx = H.Context(s, H.Integrator(H.VERLET, 0.001), "Reference")
    ctx.setPositions(w.positions)
    st = ctx.getState(getEnergy=True, getForces=True)
    assert abs(st.potentialEnergy - float(g["energy_valence"])) < 1e-6 * abs(float(g["energy_valence"]))
    assert np.abs(st.forces - g["forces_valence"]).max() < 1e-3          # the golden forces are stored as float32
    ctx.close()


def test_numpy_valence_energies_are_the_reference_platforms(description):
    """oracle/valence.py (the checker of the kernel-level tests of kernels/valence.hip) pinned: its energies of the DHFR System's valence terms,
    kind by kind, are the Reference platform's energies of the corresponding Custom*Forces."""
    from openmm_amd import harness as H, testsystems as T
    from oracle import valence as OV
    if not os.path.exists(os.path.join(H.HOST_LIB_DIR, "libOpenMMAmoebaReference.so")):
        pytest.skip("host OpenMM not built")
    H.lib()
    H._check(H.lib().omm_load_plugin(os.path.join(H.HOST_LIB_DIR, "libOpenMMAmoebaReference.so").encode()))
    d = description
    w = T.amoeba_dhfr()
    s, mp, vdw = w.build(nonbonded=False)
    names = list(w.handles)
    for g, name in enumerate(names):
        H.lib().omm_force_set_group(w.handles[name], g)
    ctx = H.Context(s, H.Integrator(H.VERLET, 0.001), "Reference")
    ctx.setPositions(w.positions)
    ref = {name: ctx.getState(getEnergy=True, groups=1 << g).potentialEnergy for g, name in enumerate(names)}
    ctx.close()
    pos = d["positions"]
    rad = 180.0 / np.pi
    poly = [float(v) for v in d["angles"][2]]
    ours = {"AmoebaBond": OV.poly_bond(pos, d["bonds"][0], d["bonds"][1], [float(d["bonds"][2]), float(d["bonds"][3])]).sum(),
            "AmoebaAngle": OV.poly_angle(pos, d["angles"][0], d["angles"][1], poly + [rad]).sum(),
            "AmoebaInPlaneAngle": OV.inplane_angle(pos, d["inplane_angles"][0], d["inplane_angles"][1], poly + [rad]).sum(),
            "AmoebaOutOfPlaneBend": OV.out_of_plane_bend(pos, d["opbends"][0], d["opbends"][1][:, None], list(d["opbends"][2]) + [rad]).sum(),
            "AmoebaStretchBend": OV.stretch_bend(pos, d["stretch_bends"][0], d["stretch_bends"][1], [rad]).sum(),
            "AmoebaPiTorsion": OV.pi_torsion(pos, d["pi_torsions"][0], d["pi_torsions"][1][:, None], []).sum()}
    for name, e in ours.items():
        assert abs(e - ref[name]) < 1e-9 * abs(ref[name]), (name, e, ref[name])


def test_subset_is_a_closed_system(description):
    from openmm_amd import forcefield_amoeba as A
    p = A.subset(description, PROTEIN_ATOMS)
    assert len(p["masses"]) == PROTEIN_ATOMS and len(p["urey_bradleys"][0]) == 0 and len(p["torsion_torsions"][0]) == len(description["torsion_torsions"][0])
    with pytest.raises(ValueError):
        A.subset(description, PROTEIN_ATOMS + 1)


@pytest.mark.skipif(not os.path.exists("/root/reference/wrappers/python/openmm/app/data/amoeba2013_gk.xml"), reason="reference tree not present")
def test_reader_is_pinned_to_the_reference_held_alanine_dipeptide_forces():
    """SURVEY.md §8(f)2 / VERDICT r4 "missing" 3 -- the pin of the AMOEBA reader against the reference's own app layer: alanine dipeptide built from
    amoeba2013.xml + amoeba2013_gk.xml with createSystem(polarization='direct') (NoCutoff; bonds, angles, in-plane angles, out-of-plane
    bends, stretch-bends, AmoebaTorsionForce torsions, pi-torsions, vdW, multipoles with their frames and covalent maps,
    generalized Kirkwood, WCA dispersion), evaluated on the Reference platform, against the golden forces of the reference's Python test
    (wrappers/python/tests/systems/alanine-dipeptide-amoeba-forces.xml -> tests/golden/forcefield_reference_forces.npz) under that test's
    criterion (TestForceField.py:1258-1262: every atom within 0.1 kJ/mol/nm or 1e-3) -- and in fact to 1e-7: every frame choice, scale
    factor and parameter lookup of the reader is the app layer's."""
    from openmm_amd import forcefield_amoeba as A, harness as H, testsystems as T
    H.load_amoeba_plugins(native=False)
    d = A.alanine_dipeptide_implicit()
    assert d["template_names"] == ["ACE", "ALA", "NME"] and len(d["masses"]) == 22
    w = T.AmoebaWorkload(d, polarization=H.Direct, no_cutoff=True)
    w.cm_remover = True
    system, mp, vdw = w.build()
    assert len(d["torsion_torsions"][0]) == 0          # (this molecule has none: the torsion-torsion lookup is pinned by the DHFR fixture and TestAmoebaTorsionTorsionForce only)
    assert {"AmoebaGeneralizedKirkwood", "AmoebaWcaDispersion", "AmoebaPiTorsion", "AmoebaStretchBend", "AmoebaOutOfPlaneBend", "AmoebaInPlaneAngle"} <= set(w.handles)
    ctx = H.Context(system, H.Integrator(H.VERLET, 0.001), "Reference")
    ctx.setPositions(w.positions)
    f = ctx.getState(getForces=True).forces
    ctx.close()
    g = np.load(os.path.join(GOLDEN, "forcefield_reference_forces.npz"))
    assert np.abs(w.positions - g["alanine_dipeptide_amoeba_positions"]).max() < 1e-12
    ref = g["alanine_dipeptide_amoeba_forces"]
    diff = np.linalg.norm(f - ref, axis=1)
    assert np.all((diff < 0.1) | (diff / np.linalg.norm(f, axis=1) < 1e-3))          # the reference's criterion
    assert diff.max() < 1e-4 and (diff / np.linalg.norm(ref, axis=1)).max() < 1e-7, diff.max()
