"""Synthetic workloads: argon box (HelloArgon parameters), TIP3P water boxes and a
DHFR-sized solvated-chain system.  Everything is generated from a seed -- no files from the
reference are read at run time.  Each builder returns a `Workload` with numpy arrays only; `build()`
turns it into harness objects (System, NonbondedForce, ...) for any platform.
"""
import numpy as np

from . import harness as H

# TIP3P (wrappers/python/openmm/app/data/tip3p.xml): charges, LJ, geometry
TIP3P = dict(qO=-0.834, qH=0.417, sigO=0.315075, epsO=0.635968, dOH=0.09572, dHH=0.15139, mO=15.99943, mH=1.007947)


class Workload:
    def __init__(self, name):
        self.name = name
        self.box = None                 # 3x3 or None
        self.masses = None
        self.positions = None
        self.charge = self.sigma = self.epsilon = None
        self.exception_bonds = None     # bonds used for createExceptionsFromBonds (or None)
        self.exceptions = None          # explicit (pairs, qq, sigma, eps)
        self.constraints = None         # (pairs, distances)
        self.bonds = self.angles = self.torsions = None
        self.method = H.PME
        self.cutoff = 0.9
        self.ewald_tol = 5e-4
        self.dispersion = True
        self.cm_remover = False
        self.pme_params = None

    @property
    def num_atoms(self):
        return len(self.masses)

    def build(self):
        """-> (System, NonbondedForce)"""
        s = H.System()
        s.addParticles(self.masses)
        if self.box is not None:
            s.setDefaultPeriodicBoxVectors(*self.box)
        nb = H.NonbondedForce(s, self.method, self.cutoff, self.ewald_tol, self.dispersion)
        nb.addParticles(self.charge, self.sigma, self.epsilon)
        if self.exception_bonds is not None and len(self.exception_bonds):
            nb.createExceptionsFromBonds(self.exception_bonds, getattr(self, "coulomb14", 1.0 / 1.2), getattr(self, "lj14", 0.5))
        if self.exceptions is not None and len(self.exceptions[0]):
            nb.addExceptions(*self.exceptions)
        if getattr(self, "reaction_field_dielectric", None) is not None:
            nb.setReactionFieldDielectric(self.reaction_field_dielectric)
        if getattr(self, "gbsa", None) is not None:               # (charge, radius, scale): GBSAOBCForce with the NonbondedForce's method and cutoff
            s.addGBSAOBCForce(self.gbsa[0], self.gbsa[1], self.gbsa[2], 0 if self.method == H.NoCutoff else (1 if self.method == H.CutoffNonPeriodic else 2), self.cutoff)
        if self.pme_params is not None:
            nb.setPMEParameters(*self.pme_params)
        if getattr(self, "ljpme_params", None) is not None:
            nb.setLJPMEParameters(*self.ljpme_params)
        if self.constraints is not None and len(self.constraints[0]):
            s.addConstraints(*self.constraints)
        if self.bonds is not None and len(self.bonds[0]):
            s.addHarmonicBondForce(*self.bonds)
        if self.angles is not None and len(self.angles[0]):
            s.addHarmonicAngleForce(*self.angles)
        if self.torsions is not None and len(self.torsions[0]):
            s.addPeriodicTorsionForce(*self.torsions)
        if self.cm_remover:
            s.addCMMotionRemover(1)
        if getattr(self, "barostat", None) is not None:          # (pressure in bar, temperature, frequency, seed)
            s.addMonteCarloBarostat(*self.barostat)
        return s, nb


def _random_rotations(rng, n):
    q = rng.normal(size=(n, 4))
    q /= np.linalg.norm(q, axis=1)[:, None]
    a, b, c, d = q.T
    return np.stack([
        np.stack([a * a + b * b - c * c - d * d, 2 * (b * c - a * d), 2 * (b * d + a * c)], -1),
        np.stack([2 * (b * c + a * d), a * a - b * b + c * c - d * d, 2 * (c * d - a * b)], -1),
        np.stack([2 * (b * d - a * c), 2 * (c * d + a * b), a * a - b * b - c * c + d * d], -1)], 1)


def water_sites():
    t = TIP3P
    half = 0.5 * t["dHH"]
    h = np.sqrt(t["dOH"] ** 2 - half ** 2)
    return np.array([[0.0, 0.0, 0.0], [half, h, 0.0], [-half, h, 0.0]])


def water_box(n_side, seed=0, density=33.4, rigid=True, method=H.PME, cutoff=0.9):
    """n_side^3 TIP3P waters on a jittered cubic lattice with random orientations."""
    rng = np.random.default_rng(seed)
    nw = n_side ** 3
    L = (nw / density) ** (1.0 / 3.0)
    spacing = L / n_side
    g = np.stack(np.meshgrid(*[np.arange(n_side)] * 3, indexing="ij"), -1).reshape(-1, 3)
    centers = (g + 0.5 + 0.15 * (rng.random((nw, 3)) - 0.5)) * spacing
    rot = _random_rotations(rng, nw)
    sites = water_sites()
    pos = centers[:, None, :] + np.einsum("nij,kj->nki", rot, sites)
    w = Workload("water-%d" % (3 * nw))
    w.positions = pos.reshape(-1, 3)
    w.box = np.eye(3) * L
    t = TIP3P
    w.masses = np.tile([t["mO"], t["mH"], t["mH"]], nw)
    w.charge = np.tile([t["qO"], t["qH"], t["qH"]], nw)
    w.sigma = np.tile([t["sigO"], 1.0, 1.0], nw)
    w.epsilon = np.tile([t["epsO"], 0.0, 0.0], nw)
    o = 3 * np.arange(nw)
    pairs = np.stack([np.stack([o, o + 1], -1), np.stack([o, o + 2], -1), np.stack([o + 1, o + 2], -1)], 1).reshape(-1, 2)
    w.exceptions = (pairs, np.zeros(len(pairs)), np.ones(len(pairs)), np.zeros(len(pairs)))
    if rigid:
        w.constraints = (pairs, np.tile([t["dOH"], t["dOH"], t["dHH"]], nw))
    else:
        ob = np.stack([np.stack([o, o + 1], -1), np.stack([o, o + 2], -1)], 1).reshape(-1, 2)
        w.bonds = (ob, np.full(len(ob), t["dOH"]), np.full(len(ob), 462750.4))
        w.angles = (np.stack([o + 1, o, o + 2], -1), np.full(nw, 1.82421813418), np.full(nw, 836.8))
    w.method, w.cutoff = method, cutoff
    return w


# AMOEBA water (wrappers/python/openmm/app/data/amoeba2009.xml: Multipole types 402 / 403 :13229-13230, Polarize :13642-13643, Vdw classes
# 73 / 74 :12733-12734 with radiussize DIAMETER (sigma halved, forcefield.py AmoebaVdwGenerator), Bond :5851, Angle :6161; the same numbers
# as plugins/amoeba/tests/TestAmoebaMultipoleForce.h:1180-1217).  Units: e, e nm, e nm^2, nm^3.
AMOEBA_WATER = dict(
    qO=-0.51966, qH=0.25983,
    dO=(0.0, 0.0, 0.00755612136146), dH=(-0.00204209484795, 0.0, -0.00307875299958),
    QO=((0.000354030721139, 0.0, 0.0), (0.0, -0.000390257077096, 0.0), (0.0, 0.0, 3.62263559571e-05)),
    QH=((-3.42848248983e-05, 0.0, -1.89485963908e-06), (0.0, -0.000100240875193, 0.0), (-1.89485963908e-06, 0.0, 0.000134525700091)),
    polO=0.000837, polH=0.000496, thole=0.39,
    sigO=0.5 * 0.3405, epsO=0.46024, redO=1.0, sigH=0.5 * 0.2655, epsH=0.056484, redH=0.91,
    mO=15.999, mH=1.008, dOH=0.09572, kBond=2 * 221584.64, angle=np.deg2rad(108.5), kAngle=2 * 0.0433973816335 * (180.0 / np.pi) ** 2)


class AmoebaWaterWorkload:
    """Water with the AMOEBA multipole (PME, mutual or direct polarization) and buffered 14-7 vdW forces -- the two AMOEBA forces the HIP
    platform computes natively -- on given O, H, H coordinates.  AMOEBA's anharmonic bond / angle / Urey-Bradley terms are replaced by their
    harmonic parts (HarmonicBondForce / HarmonicAngleForce, native as well): this System times the nonbonded AMOEBA path, it is not the
    amoeba2009 water model to the letter."""

    def __init__(self, positions, box, cutoff=0.7, vdw_cutoff=0.9, polarization=H.Mutual, epsilon=1e-5, ewald_tol=7.5e-4, grid=None, a_ewald=0.0, bonded=True):
        self.positions = np.asarray(positions, dtype=np.float64)
        self.box = np.asarray(box, dtype=np.float64)
        self.cutoff, self.vdw_cutoff, self.polarization, self.epsilon, self.ewald_tol, self.grid, self.a_ewald, self.bonded = cutoff, vdw_cutoff, polarization, epsilon, ewald_tol, grid, a_ewald, bonded
        self.name = "amoeba-water-%d" % len(self.positions)
        nw = len(self.positions) // 3
        a = AMOEBA_WATER
        self.masses = np.tile([a["mO"], a["mH"], a["mH"]], nw)

    @property
    def num_atoms(self):
        return len(self.positions)

    def build(self):
        """-> (System, AmoebaMultipoleForce, AmoebaVdwForce)"""
        a = AMOEBA_WATER
        nw = self.num_atoms // 3
        o = 3 * np.arange(nw)
        s = H.System()
        s.addParticles(self.masses)
        s.setDefaultPeriodicBoxVectors(*self.box)
        mp = H.AmoebaMultipoleForce(s, H.AmoebaMultipoleForce.PME, self.polarization, self.cutoff, self.a_ewald, self.grid, self.ewald_tol, self.epsilon, 100)
        axes = np.stack([np.stack([np.full(nw, H.Bisector), o + 1, o + 2, np.full(nw, -1)], -1),
                         np.stack([np.full(nw, H.ZThenX), o, o + 2, np.full(nw, -1)], -1),
                         np.stack([np.full(nw, H.ZThenX), o, o + 1, np.full(nw, -1)], -1)], 1).reshape(-1, 4)
        pol = np.tile([a["polO"], a["polH"], a["polH"]], nw)
        mp.addMultipoles(np.tile([a["qO"], a["qH"], a["qH"]], nw), np.tile(np.array([a["dO"], a["dH"], a["dH"]]), (nw, 1)),
                         np.tile(np.array([a["QO"], a["QH"], a["QH"]]), (nw, 1, 1)), axes, np.full(3 * nw, a["thole"]), pol ** (1.0 / 6.0), pol)
        atoms, types, lists = [], [], []
        for w in range(nw):
            i = 3 * w
            for atom, t, l in ((i, H.Covalent12, (i + 1, i + 2)), (i + 1, H.Covalent12, (i,)), (i + 2, H.Covalent12, (i,)),
                               (i + 1, H.Covalent13, (i + 2,)), (i + 2, H.Covalent13, (i + 1,)),
                               (i, H.PolarizationCovalent11, (i, i + 1, i + 2)), (i + 1, H.PolarizationCovalent11, (i, i + 1, i + 2)), (i + 2, H.PolarizationCovalent11, (i, i + 1, i + 2))):
                atoms.append(atom); types.append(t); lists.append(l)
        mp.setCovalentMaps(atoms, types, lists)
        vdw = H.AmoebaVdwForce(s, "CUBIC-MEAN", "HHG", H.AmoebaVdwForce.CutoffPeriodic, self.vdw_cutoff, True)
        vdw.addParticles(np.stack([o, o, o], -1).reshape(-1), np.tile([a["sigO"], a["sigH"], a["sigH"]], nw), np.tile([a["epsO"], a["epsH"], a["epsH"]], nw),
                         np.tile([a["redO"], a["redH"], a["redH"]], nw))
        vdw.setParticleExclusions([(3 * (i // 3), 3 * (i // 3) + 1, 3 * (i // 3) + 2) for i in range(3 * nw)])
        if self.bonded:
            ob = np.stack([np.stack([o, o + 1], -1), np.stack([o, o + 2], -1)], 1).reshape(-1, 2)
            s.addHarmonicBondForce(ob, np.full(len(ob), a["dOH"]), np.full(len(ob), a["kBond"]))
            s.addHarmonicAngleForce(np.stack([o + 1, o, o + 2], -1), np.full(nw, a["angle"]), np.full(nw, a["kAngle"]))
        return s, mp, vdw


class AmoebaWorkload:
    """A System with the complete AMOEBA force field from a description of forcefield_amoeba.create_description() -- the Forces that
    wrappers/python/openmm/app/forcefield.py creates for amoeba2009.xml, in the force groups of examples/benchmark.py:69-73 (multipoles and
    vdW in group 1, all valence terms in group 0).  The energy expressions of the Custom*Forces are the reference's (forcefield.py:3340,
    3503, 3567-3576, 3732-3741, 4428, 4064-4072): they define the force field."""

    RAD = 180.0 / np.pi

    def __init__(self, description, cutoff=0.7, vdw_cutoff=0.9, polarization=H.Mutual, epsilon=1e-5, ewald_tol=7.5e-4, grid=None, a_ewald=0.0, no_cutoff=False):
        self.d = description
        self.no_cutoff = no_cutoff          # createSystem's default nonbondedMethod (implicit-solvent Systems): multipoles and vdW over all pairs
        self.name = description["name"]
        self.positions, self.box, self.masses = description["positions"], description["box"], description["masses"]
        self.cutoff, self.vdw_cutoff, self.polarization, self.epsilon, self.ewald_tol, self.grid, self.a_ewald = cutoff, vdw_cutoff, polarization, epsilon, ewald_tol, grid, a_ewald

    @property
    def num_atoms(self):
        return len(self.positions)

    def build(self, valence=True, nonbonded=True):
        """-> (System, AmoebaMultipoleForce or None, AmoebaVdwForce or None)"""
        d = self.d
        s = H.System()
        s.addParticles(self.masses)
        s.setDefaultPeriodicBoxVectors(*self.box)
        self.handles = handles = {}          # force name -> handle (omm_force_set_group for per-term diagnostics)

        def named(handle, name):
            H.lib().omm_force_set_name(handle, name.encode())
            handles[name] = handle
        plane = ("projx = x2-nx*dot; projy = y2-ny*dot; projz = z2-nz*dot; dot = nx*(x2-x3) + ny*(y2-y3) + nz*(z2-z3); nx = px/norm; ny = py/norm; nz = pz/norm; "
                 "norm = sqrt(px*px + py*py + pz*pz); px = (d1y*d2z-d1z*d2y); py = (d1z*d2x-d1x*d2z); pz = (d1x*d2y-d1y*d2x); "
                 "d1x = x1-x4; d1y = y1-y4; d1z = z1-z4; d2x = x3-x4; d2y = y3-y4; d2z = z3-z4")
        if valence:
            atoms, par, cubic, quartic = d["bonds"]
            named(s.addCustomBondForce("k*(d^2 + %s*d^3 + %s*d^4); d=r-r0" % (cubic, quartic), ["r0", "k"], atoms, par), "AmoebaBond")
            atoms, par, poly = d["angles"]
            sextic = "k*(d^2 + %s*d^3 + %s*d^4 + %s*d^5 + %s*d^6)" % tuple(poly)
            named(s.addCustomAngleForce(sextic + "; d=%.15g*theta-theta0" % self.RAD, ["theta0", "k"], atoms, par), "AmoebaAngle")
            atoms, par = d["inplane_angles"]
            if len(atoms):
                e = sextic + "; d=theta-theta0; theta = %.15g*pointangle(x1, y1, z1, projx, projy, projz, x3, y3, z3); " % self.RAD + plane
                named(s.addCustomCompoundBondForce(4, e, ["theta0", "k"], atoms, par), "AmoebaInPlaneAngle")
            atoms, k, poly = d["opbends"]
            if len(atoms):
                e = ("k*(theta^2 + %s*theta^3 + %s*theta^4 + %s*theta^5 + %s*theta^6); " % tuple(repr(float(v)) for v in poly) +
                     "theta = %.15g*pointangle(x2, y2, z2, x4, y4, z4, projx, projy, projz); " % self.RAD + plane)
                named(s.addCustomCompoundBondForce(4, e, ["k"], atoms, k), "AmoebaOutOfPlaneBend")
            atoms, par = d["stretch_bends"]
            if len(atoms):
                e = "(k1*(distance(p1,p2)-r12) + k2*(distance(p2,p3)-r23))*(%.15g*(angle(p1,p2,p3)-theta0))" % self.RAD
                named(s.addCustomCompoundBondForce(3, e, ["r12", "r23", "theta0", "k1", "k2"], atoms, par), "AmoebaStretchBend")
            atoms, par = d["urey_bradleys"]
            if len(atoms):
                named(s.addHarmonicBondForce(atoms, par[:, 0], par[:, 1]), "AmoebaUreyBradley")
            atoms, par = d["torsions"]
            if len(atoms):
                named(s.addPeriodicTorsionForce(atoms, par[:, 0].astype(np.int32), par[:, 1], par[:, 2]), "PeriodicTorsion")
            atoms, k = d["pi_torsions"]
            if len(atoms):
                e = ("2*k*sin(phi)^2; phi = pointdihedral(x3+c1x, y3+c1y, z3+c1z, x3, y3, z3, x4, y4, z4, x4+c2x, y4+c2y, z4+c2z); "
                     "c1x = (d14y*d24z-d14z*d24y); c1y = (d14z*d24x-d14x*d24z); c1z = (d14x*d24y-d14y*d24x); "
                     "c2x = (d53y*d63z-d53z*d63y); c2y = (d53z*d63x-d53x*d63z); c2z = (d53x*d63y-d53y*d63x); "
                     "d14x = x1-x4; d14y = y1-y4; d14z = z1-z4; d24x = x2-x4; d24y = y2-y4; d24z = z2-z4; "
                     "d53x = x5-x3; d53y = y5-y3; d53z = z5-z3; d63x = x6-x3; d63y = y6-y3; d63z = z6-z3")
                named(s.addCustomCompoundBondForce(6, e, ["k"], atoms, k), "AmoebaPiTorsion")
            atoms, grid_index, grids = d["torsion_torsions"]
            if len(atoms):
                handles["AmoebaTorsionTorsion"] = H.AmoebaTorsionTorsionForce(s, atoms, grid_index, grids).h
        mp = vdw = None
        if nonbonded:
            m = d["multipoles"]
            mp = H.AmoebaMultipoleForce(s, H.AmoebaMultipoleForce.NoCutoff if self.no_cutoff else H.AmoebaMultipoleForce.PME, self.polarization, self.cutoff, self.a_ewald, self.grid, self.ewald_tol, self.epsilon, 100)
            mp.addMultipoles(m["charge"], m["dipole"], m["quadrupole"].reshape(-1, 3, 3), m["axes"], m["thole"], m["damping"], m["polarity"])
            mp.setCovalentMaps([e[0] for e in m["covalent_maps"]], [e[1] for e in m["covalent_maps"]], [e[2] for e in m["covalent_maps"]])
            v = d["vdw"]
            vdw = H.AmoebaVdwForce(s, v["sigma_rule"], v["epsilon_rule"], H.AmoebaVdwForce.NoCutoff if self.no_cutoff else H.AmoebaVdwForce.CutoffPeriodic, self.vdw_cutoff, True)
            vdw.addParticles(v["parent"], v["sigma"], v["epsilon"], v["reduction"])
            vdw.setParticleExclusions(v["exclusions"])
            if "gk" in d:          # generalized Kirkwood + WCA dispersion (implicit solvent; forcefield.py:5287-5617)
                g, wca = d["gk"], d["wca"]
                handles["AmoebaGeneralizedKirkwood"] = H.addAmoebaGeneralizedKirkwoodForce(s, g["charge"], g["radius"], g["scale"], g["solventDielectric"], g["soluteDielectric"],
                                                                                         int(g["includeCavityTerm"]), g["probeRadius"], g["surfaceAreaFactor"])
                handles["AmoebaWcaDispersion"] = H.addAmoebaWcaDispersionForce(s, wca["radius"], wca["epsilon"], *[wca[k] for k in ("epso", "epsh", "rmino", "rminh", "awater", "slevy", "dispoff", "shctd")])
            handles["AmoebaMultipole"], handles["AmoebaVdw"] = mp.h, vdw.h
            H.lib().omm_force_set_group(mp.h, 1)
            H.lib().omm_force_set_group(vdw.h, 1)
        if getattr(self, "cm_remover", False):
            s.addCMMotionRemover(1)
        return s, mp, vdw


def amoeba_dhfr(pin_grid=False, **kw):
    """examples/benchmark.py `amoebapme` (BASELINE.json configs[4]): DHFR in water (23 558 atoms), amoeba2009, multipole PME cutoff 0.7 nm /
    tolerance 7.5e-4, vdW cutoff 0.9 nm, no constraints, mutual polarization at epsilon 1e-5, CMMotionRemover -- from the fixture
    tests/golden/amoeba_dhfr_5dfr_amoeba2009.npz (tools/make_amoeba_dhfr_fixture.py).  pin_grid: write the 64^3 grid and the alpha that
    the tolerance rule gives into the force, so that every platform computes the same sum whatever its rounding of grid sizes."""
    import os
    from . import forcefield_amoeba as A
    d = A.load_description(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "amoeba_dhfr_5dfr_amoeba2009.npz"))
    kw.setdefault("cutoff", 0.7)
    kw.setdefault("ewald_tol", 7.5e-4)
    if pin_grid:
        kw["grid"] = (64, 64, 64)
        kw["a_ewald"] = float(np.sqrt(-np.log(2 * kw["ewald_tol"])) / kw["cutoff"])
    w = AmoebaWorkload(d, **kw)
    w.cm_remover = True
    return w


def amoeba_water_box(n_side, seed=0, **kw):
    """n_side^3 AMOEBA waters on the jittered lattice of water_box()."""
    w = water_box(n_side, seed=seed)
    return AmoebaWaterWorkload(w.positions, w.box, **kw)


def amoeba_water_tile(**kw):
    """12 167 AMOEBA waters (36 501 atoms) on the coordinates of the equilibrated TIP3P tile (tests/golden/water_tile_36501_equilibrated.npz)."""
    import os
    d = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "water_tile_36501_equilibrated.npz"))
    return AmoebaWaterWorkload(d["positions"], np.eye(3) * float(d["box"]), **kw)


def apoa1_like(seed=0):
    """BASELINE.json configs[2] stand-in (examples/apoa1.pdb is not in the reference tree): 92 224
    atoms in the apoa1 box 10.8861 x 10.8861 x 7.7758 nm -- 30 741 TIP3P waters on a jittered 36 x 35 x 25 lattice with
    759 sites left empty, plus one argon atom on an empty site.  PME at cutoff 0.9 nm / 5e-4 on a 98 x 98 x 70 grid."""
    rng = np.random.default_rng(seed)
    box = np.array([10.8861, 10.8861, 7.7758])
    dims = np.array([36, 35, 25])
    nw = 30741
    g = np.stack(np.meshgrid(*[np.arange(d) for d in dims], indexing="ij"), -1).reshape(-1, 3)
    order = rng.permutation(len(g))
    spacing = box / dims
    centers = (g[order[:nw]] + 0.5 + 0.15 * (rng.random((nw, 3)) - 0.5)) * spacing
    rot = _random_rotations(rng, nw)
    pos = (centers[:, None, :] + np.einsum("nij,kj->nki", rot, water_sites())).reshape(-1, 3)
    argon = (g[order[nw]] + 0.5) * spacing
    w = Workload("apoa1-sized-water-92224")
    w.positions = np.vstack([pos, argon[None, :]])
    w.box = np.diag(box)
    t = TIP3P
    w.masses = np.concatenate([np.tile([t["mO"], t["mH"], t["mH"]], nw), [39.95]])
    w.charge = np.concatenate([np.tile([t["qO"], t["qH"], t["qH"]], nw), [0.0]])
    w.sigma = np.concatenate([np.tile([t["sigO"], 1.0, 1.0], nw), [0.3350]])
    w.epsilon = np.concatenate([np.tile([t["epsO"], 0.0, 0.0], nw), [0.996]])
    o = 3 * np.arange(nw)
    pairs = np.stack([np.stack([o, o + 1], -1), np.stack([o, o + 2], -1), np.stack([o + 1, o + 2], -1)], 1).reshape(-1, 2)
    w.exceptions = (pairs, np.zeros(len(pairs)), np.ones(len(pairs)), np.zeros(len(pairs)))
    w.constraints = (pairs, np.tile([t["dOH"], t["dOH"], t["dHH"]], nw))
    # The tolerance rule gives 97 x 97 x 70 on the Reference platform (any size is legal for fftpack) and 98 x 98 x 70 on
    # platforms that round up to FFT-friendly sizes (CudaFFT3D.cpp:127-142, and ours); pin the grid so both compute the same sum.
    w.pme_params = (float(np.sqrt(-np.log(2 * w.ewald_tol)) / w.cutoff), 98, 98, 70)
    return w


def argon_box(n_cells=6, seed=0, method=H.NoCutoff):
    """864-atom (6x6x6 fcc) argon box with the HelloArgon parameters (examples/HelloArgon.cpp:36-44)."""
    rng = np.random.default_rng(seed)
    a = 0.54
    basis = np.array([[0, 0, 0], [0.5, 0.5, 0], [0.5, 0, 0.5], [0, 0.5, 0.5]])
    g = np.stack(np.meshgrid(*[np.arange(n_cells)] * 3, indexing="ij"), -1).reshape(-1, 1, 3)
    pos = ((g + basis[None]) * a).reshape(-1, 3) + rng.uniform(-0.01, 0.01, (4 * n_cells ** 3, 3))
    n = len(pos)
    w = Workload("argon-%d" % n)
    w.positions = pos
    w.masses = np.full(n, 39.95)
    w.charge = np.zeros(n)
    w.sigma = np.full(n, 0.3350)
    w.epsilon = np.full(n, 0.996)
    w.method = method
    w.dispersion = False
    if method != H.NoCutoff:
        w.box = np.eye(3) * (n_cells * a)
        w.cutoff = 1.0
    return w


def _load_fixture(w, filename):
    """Replace the lattice start by the equilibrated coordinates/velocities of tests/golden/<filename>
    (made by tools/make_workload_fixtures.py on the reference's CPU platform), if that file exists."""
    import os
    path = os.path.join(H.ROOT, "tests", "golden", filename)
    w.velocities = None
    w.relaxed = False
    if os.path.exists(path):
        data = np.load(path)
        if data["positions"].shape == w.positions.shape:
            w.positions = data["positions"].astype(np.float64)
            w.velocities = data["velocities"].astype(np.float64)
            w.relaxed = True
    return w


def with_barostat(w, pressure, temperature, frequency, seed):
    """The same workload with a MonteCarloBarostat (pressure in bar)."""
    w.barostat = (pressure, temperature, frequency, seed)
    return w


def water_tiled(reps=3):
    """reps^3 copies of an equilibrated water box: tests/golden/water_tile_36501_equilibrated.npz holds 23^3 rigid TIP3P waters after
    60 ps at 300 K on the HIP platform (tools/make_water_tile.py).  reps = 3 gives the 985 527 atoms and the 21.4 nm box of
    `water_box(69)` -- BASELINE.json configs[3]'s size -- as a liquid at 300 K with thermal velocities, where the lattice start of
    `water_box` melts at several hundred kelvin above that.  A periodic tiling of an equilibrium configuration is one itself; the
    copies decorrelate within the first steps of a Langevin run (independent noise per atom)."""
    import os
    d = np.load(os.path.join(H.ROOT, "tests", "golden", "water_tile_36501_equilibrated.npz"))
    L, n_side = float(d["box"]), int(d["n_side"])
    w = water_box(n_side * reps, seed=0)                 # parameters, constraints, exceptions (the positions are replaced below)
    assert abs(float(w.box[0][0]) - reps * L) < 1e-9 * L
    shifts = np.stack(np.meshgrid(*[np.arange(reps)] * 3, indexing="ij"), -1).reshape(-1, 3) * L
    w.positions = (d["positions"].astype(np.float64)[None, :, :] + shifts[:, None, :]).reshape(-1, 3)
    w.velocities = np.tile(d["velocities"].astype(np.float64), (reps ** 3, 1))
    w.name = "water-%d-equilibrated" % len(w.positions)
    w.prepare_steps = 200                                # untimed steps before a measurement: the copies go their own ways
    return w


def water_row(n_side, reps, seed=0, cutoff=0.4):
    """`reps` copies of `water_box(n_side)` side by side along x: a long thin box (reps L x L x L) for runs on MANY ranks, whose x-slabs
    must each be wider than the halo -- 8 slabs of a cubic box small enough for the CPU emulator would not be.  The copies
    are identical at the start, which periodic boundaries allow; independent Langevin noise separates them."""
    c = water_box(n_side, seed=seed, cutoff=cutoff)
    L, n = float(c.box[0][0]), len(c.positions)
    w = Workload("water-row-%d" % (n * reps))
    w.positions = (c.positions[None, :, :] + np.arange(reps)[:, None, None] * np.array([L, 0.0, 0.0])).reshape(-1, 3)
    w.box = np.diag([reps * L, L, L])
    for name in ("masses", "charge", "sigma", "epsilon"):
        setattr(w, name, np.tile(getattr(c, name), reps))
    offs = (np.arange(reps) * n)[:, None, None]
    pairs, qq, sig, eps = c.exceptions
    w.exceptions = ((pairs[None] + offs).reshape(-1, 2), np.tile(qq, reps), np.tile(sig, reps), np.tile(eps, reps))
    cp, cd = c.constraints
    w.constraints = ((cp[None] + offs).reshape(-1, 2), np.tile(cd, reps))
    w.method, w.cutoff = c.method, c.cutoff
    return w


def dhfr():
    """The real DHFR benchmark System (examples/benchmark.py `pme`: 5dfr_solv-cube_equil.pdb, amber99sb + tip3p, PME 0.9 nm, HBonds,
    rigid water, CMMotionRemover) from the fixture tests/golden/dhfr_5dfr_amber99sb_tip3p.npz, which tools/make_dhfr_fixture.py
    builds from the reference's own input files with openmm_amd/forcefield.py; equilibrated positions and velocities."""
    import os
    d = np.load(os.path.join(H.ROOT, "tests", "golden", "dhfr_5dfr_amber99sb_tip3p.npz"))
    w = Workload(str(d["name"]))
    w.box = d["box"]
    w.masses, w.charge = d["masses"].astype(np.float64), d["charge"].astype(np.float64)
    w.sigma, w.epsilon = d["sigma"], d["epsilon"]
    w.exception_bonds = d["exception_bonds"].astype(np.int64)
    w.coulomb14, w.lj14, w.cutoff = float(d["coulomb14"]), float(d["lj14"]), float(d["cutoff"])
    w.bonds = (d["bond_atoms"].astype(np.int64), d["bond_length"], d["bond_k"])
    w.angles = (d["angle_atoms"].astype(np.int64), d["angle_theta"], d["angle_k"])
    w.torsions = (d["torsion_atoms"].astype(np.int64), d["torsion_n"].astype(np.int32), d["torsion_phase"], d["torsion_k"])
    w.constraints = (d["constraint_atoms"].astype(np.int64), d["constraint_length"])
    w.positions, w.velocities = d["positions"], d["velocities"].astype(np.float64)
    w.pdb_positions, w.pdb_potential_energy = d["pdb_positions"].astype(np.float64), float(d["pdb_potential_energy"])
    w.method, w.dispersion, w.cm_remover, w.relaxed = H.PME, True, True, True
    return w


def with_cutoff(w, cutoff):
    """The same workload with another nonbonded cutoff (tests of the domain decomposition want halos thinner than their small boxes)."""
    w.cutoff = cutoff
    return w


def sheared(w, bx, cx, cy, affine=False):
    """A workload of a cubic box in the triclinic box a = (L, 0, 0), b = (bx, L, 0), c = (cx, cy, L): every molecule (the atoms connected by
    bonds and constraints) is moved as a whole by the shear of its first atom, so the density and the molecules' shapes stay what they were.
    That suits small molecules; a large one would be left in a cavity that no longer has its shape, so `affine` shears every atom instead
    (bonds and angles a little distorted -- the caller applies the constraints; fine for tests that start unrelaxed anyway)."""
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components
    L = float(w.box[0][0])
    n = len(w.positions)
    edges = [np.asarray(t[0]).reshape(-1, 2) for t in (w.bonds, w.constraints) if t is not None and len(t[0])]
    edges = np.concatenate(edges) if edges else np.zeros((0, 2), int)
    _, molecule = connected_components(coo_matrix((np.ones(len(edges)), (edges[:, 0], edges[:, 1])), shape=(n, n)), directed=False)
    first = np.full(molecule.max() + 1, n)
    np.minimum.at(first, molecule, np.arange(n))
    o = w.positions if affine else w.positions[first[molecule]]
    shift = np.zeros_like(w.positions)
    shift[:, 0] = o[:, 1] / L * bx + o[:, 2] / L * cx
    shift[:, 1] = o[:, 2] / L * cy
    w.positions = w.positions + shift
    w.box = np.array([[L, 0.0, 0.0], [bx, L, 0.0], [cx, cy, L]])
    w.name = getattr(w, "name", "workload") + "-triclinic"
    return w


def small_solvated_chain(seed=3):
    """The same construction at test size: a 150-atom chain (bonds, angles, torsions, 1-4s, X-H clusters) in ~660 waters."""
    return dhfr_like(seed=seed, n_side=9, chain_atoms=150, relaxed=False, L=2.75, n_target=2130, radius=0.9)


def dhfr_like(seed=1, n_side=22, chain_atoms=2489, relaxed=True, L=6.223, n_target=23558, radius=2.2):
    """A DHFR-sized stand-in: 23 558 atoms = a 2 489-atom flexible heteropolymer
    (bonds, angles, torsions, 1-4 exceptions, X-H constraints) solvated in TIP3P water in a 6.223 nm cube,
    PME, cutoff 0.9 nm, default Ewald tolerance -> alpha 2.92 / grid 56^3, as examples/benchmark.py 'pme'.
    """
    rng = np.random.default_rng(seed)
    # --- chain: a compact self-avoiding walk of heavy atoms (0.15 nm bonds, >= 0.28 nm between atoms more
    #     than three bonds apart), each carrying 0-2 hydrogens placed away from every other heavy atom
    n_heavy = int(chain_atoms / 2.0)
    step, min_dist = 0.15, 0.28
    center = np.array([L / 2, L / 2, L / 2])
    heavy = np.zeros((n_heavy, 3))
    heavy[0] = center
    count = 1
    while count < n_heavy:
        placed = False
        for _ in range(200):
            d = rng.normal(size=3)
            d /= np.linalg.norm(d)
            q = heavy[count - 1] + step * d
            if np.linalg.norm(q - center) > radius:
                continue
            if count > 3 and np.min(np.linalg.norm(heavy[:count - 3] - q, axis=1)) < min_dist:
                continue
            # backbone angle between 94 and 130 degrees: torsion forces diverge as 1/sin^2 when three consecutive atoms
            # line up, which no force field allows either
            if count > 1 and not (0.22 < np.linalg.norm(heavy[count - 2] - q) < 0.272):
                continue
            heavy[count] = q
            count += 1
            placed = True
            break
        if not placed:                      # dead end: back up a few atoms and try again
            count = max(1, count - 5)
    pos, masses, charge, sigma, eps, bonds, is_h = [], [], [], [], [], [], []
    heavy_index = []
    for i, hp in enumerate(heavy):
        idx = len(pos)
        heavy_index.append(idx)
        pos.append(hp); masses.append(12.011 if i % 4 else 14.007); is_h.append(False)
        charge.append(rng.normal(0, 0.25)); sigma.append(0.30 if i % 4 else 0.29); eps.append(0.36 if i % 4 else 0.71)
        if i > 0:
            bonds.append((heavy_index[i - 1], idx))
        n_h = 0 if len(pos) >= chain_atoms else (1 if i % 3 == 0 else (2 if i % 3 == 1 else 0))
        others = np.delete(heavy, i, axis=0)
        placed_h = []
        for _ in range(n_h):
            if len(pos) >= chain_atoms:
                break
            best, best_d = None, -1.0
            for _try in range(30):
                d = rng.normal(size=3); d /= np.linalg.norm(d)
                q = hp + 0.109 * d
                dm = np.min(np.linalg.norm(others - q, axis=1))
                for ph in placed_h:
                    dm = min(dm, np.linalg.norm(ph - q) + 0.05)
                if dm > best_d:
                    best, best_d = q, dm
            placed_h.append(best)
            bonds.append((idx, len(pos)))
            pos.append(best); masses.append(1.008); is_h.append(True)
            # hydrogens carry positive partial charges only (as in protein force fields): a negative hydrogen has no
            # LJ core to stop it from collapsing onto a positive one
            charge.append(0.03 + abs(rng.normal(0, 0.1))); sigma.append(0.107); eps.append(0.066)
    pos = np.array(pos)[:chain_atoms]
    nc = len(pos)
    masses, charge, sigma, eps, is_h = [np.array(x)[:nc] for x in (masses, charge, sigma, eps, is_h)]
    bonds = np.array([b for b in bonds if b[0] < nc and b[1] < nc])
    charge[~is_h] -= charge.sum() / np.count_nonzero(~is_h)      # neutral chain, compensated on the heavy atoms
    # --- waters on a lattice, skipping sites that overlap the chain
    n_water = (n_target - nc) // 3
    spacing = L / n_side
    g = np.stack(np.meshgrid(*[np.arange(n_side)] * 3, indexing="ij"), -1).reshape(-1, 3)
    centers = (g + 0.5) * spacing
    from scipy.spatial import cKDTree
    tree = cKDTree(pos)
    dmin, _ = tree.query(centers)
    order = np.argsort(-dmin)[:n_water]
    centers = centers[np.sort(order)] + 0.05 * (rng.random((n_water, 3)) - 0.5)
    rot = _random_rotations(rng, n_water)
    wpos = (centers[:, None, :] + np.einsum("nij,kj->nki", rot, water_sites())).reshape(-1, 3)
    t = TIP3P
    w = Workload("dhfr-like-%d" % (nc + 3 * n_water))
    w.positions = np.concatenate([pos, wpos])
    w.box = np.eye(3) * L
    w.masses = np.concatenate([masses, np.tile([t["mO"], t["mH"], t["mH"]], n_water)])
    w.charge = np.concatenate([charge, np.tile([t["qO"], t["qH"], t["qH"]], n_water)])
    w.sigma = np.concatenate([sigma, np.tile([t["sigO"], 1.0, 1.0], n_water)])
    w.epsilon = np.concatenate([eps, np.tile([t["epsO"], 0.0, 0.0], n_water)])
    o = nc + 3 * np.arange(n_water)
    wpairs = np.stack([np.stack([o, o + 1], -1), np.stack([o, o + 2], -1), np.stack([o + 1, o + 2], -1)], 1).reshape(-1, 2)
    w.exception_bonds = bonds
    w.exceptions = (wpairs, np.zeros(len(wpairs)), np.ones(len(wpairs)), np.zeros(len(wpairs)))
    # constraints: waters (SETTLE) + X-H bonds of the chain (SHAKE clusters)
    hb = np.array([b for b in bonds if is_h[b[0]] or is_h[b[1]]])
    hb_len = np.linalg.norm(pos[hb[:, 0]] - pos[hb[:, 1]], axis=1)
    w.constraints = (np.concatenate([hb, wpairs]), np.concatenate([hb_len, np.tile([t["dOH"], t["dOH"], t["dHH"]], n_water)]))
    hv = np.array([b for b in bonds if not (is_h[b[0]] or is_h[b[1]])])
    hv_len = np.linalg.norm(pos[hv[:, 0]] - pos[hv[:, 1]], axis=1)
    w.bonds = (hv, hv_len, np.full(len(hv), 250000.0))
    # angles and torsions along the heavy-atom backbone at their current values (a relaxed structure)
    hi = np.array(heavy_index)[: np.searchsorted(heavy_index, nc)]
    ang = np.stack([hi[:-2], hi[1:-1], hi[2:]], -1)
    v1 = pos[ang[:, 0]] - pos[ang[:, 1]]
    v2 = pos[ang[:, 2]] - pos[ang[:, 1]]
    cosang = np.einsum("ij,ij->i", v1, v2) / np.linalg.norm(v1, axis=1) / np.linalg.norm(v2, axis=1)
    # every hydrogen also gets H-X-Y angle terms (tetrahedral), as a force field would give it
    nbrs = [[] for _ in range(nc)]
    for a, b in bonds:
        nbrs[a].append(b); nbrs[b].append(a)
    h_ang = []
    for x in range(nc):
        if is_h[x]:
            continue
        for h in nbrs[x]:
            if not is_h[h]:
                continue
            for y in nbrs[x]:
                if y != h and (not is_h[y] or h < y):
                    h_ang.append((h, x, y))
    h_ang = np.array(h_ang, dtype=np.int64).reshape(-1, 3)
    ang_all = np.concatenate([ang, h_ang])
    theta0 = np.concatenate([np.arccos(np.clip(cosang, -1, 1)), np.full(len(h_ang), 1.911)])
    w.angles = (ang_all, theta0, np.concatenate([np.full(len(ang), 400.0), np.full(len(h_ang), 300.0)]))
    tor = np.stack([hi[:-3], hi[1:-2], hi[2:-1], hi[3:]], -1)
    w.torsions = (tor, np.full(len(tor), 3, dtype=np.int32), np.zeros(len(tor)), np.full(len(tor), 1.0))
    w.cm_remover = True
    w.velocities = None
    w.relaxed = False
    if relaxed:
        _load_fixture(w, "dhfr_like_seed%d_equilibrated.npz" % seed)
    return w


def constraint_zoo(seed=5, ccma_heavy=24):
    """Every constraint algorithm of the platform in one System: the solvated chain of
    `small_solvated_chain` -- rigid waters (SETTLE), X-H clusters (SHAKE) -- with the first `ccma_heavy` heavy-atom bonds of the chain
    constrained as well (an "AllBonds" stretch: those atoms and their hydrogens form one connected constraint graph, which the
    Reference platform and this one hand to CCMA; its coupling matrix takes the angles from the HarmonicAngleForce)."""
    w = small_solvated_chain(seed=seed)
    hv, hv_len, hv_k = w.bonds
    take = np.arange(len(hv)) < ccma_heavy
    w.constraints = (np.concatenate([w.constraints[0], hv[take]]), np.concatenate([w.constraints[1], hv_len[take]]))
    w.bonds = (hv[~take], hv_len[~take], hv_k[~take])
    w.name = "constraint-zoo-%d" % w.num_atoms
    w.cm_remover = False
    return w


def nacl_amorph():
    """The amorphous NaCl system of the reference's tests/TestEwald.h:98-220 (894 ions, Ewald summation, tolerance 1e-5) from the
    committed fixture tests/golden/nacl_amorph.npz (positions, box, cutoff and the Gromacs energy copied from that test by
    tools/make_golden_from_reference.py)."""
    import os
    g = np.load(os.path.join(H.ROOT, "tests", "golden", "nacl_amorph.npz"))
    pos = g["positions"].astype(np.float64)
    n = len(pos)
    w = Workload("nacl-amorph-%d" % n)
    w.positions, w.box = pos, np.eye(3) * float(g["box"])
    w.masses = np.concatenate([np.full(n // 2, 22.99), np.full(n // 2, 35.45)])
    w.charge, w.sigma, w.epsilon = np.concatenate([np.ones(n // 2), -np.ones(n // 2)]), np.ones(n), np.zeros(n)
    w.method, w.cutoff, w.ewald_tol, w.dispersion = H.Ewald, float(g["cutoff"]), float(g["ewald_tol"]), False
    w.gromacs_energy = float(g["gromacs_energy"])
    return w
