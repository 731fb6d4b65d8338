"""amoeba2009.xml -> the System of examples/benchmark.py's `amoebapme` test (BASELINE.json configs[4]: DHFR in water, AMOEBA 2009, PME)
without the SWIG Python layer: the counterpart of forcefield.py for the polarizable force field.  Pure Python + numpy.

It restates what wrappers/python/openmm/app/forcefield.py does with this file for ForceField('amoeba2009.xml').createSystem(topology,
nonbondedMethod=PME, nonbondedCutoff=0.7, vdwCutoff=0.9, constraints=None, ...) (examples/benchmark.py:58-68) -- generator by generator:

  AmoebaBondForce            quartic bonds as a CustomBondForce                                   (forcefield.py:3324-3394)
  AmoebaAngleForce           sextic angles (CustomAngleForce); at trigonal centres with out-of-plane parameters the IN-PLANE angle
                             (CustomCompoundBondForce over 4 atoms); ideal angle by the number of hydrogens on the centre  (:3426-3607)
  AmoebaOutOfPlaneBendForce  Allinger out-of-plane bends at trigonal centres, three per centre, and the partition of the angles into
                             in-plane and ordinary ones                                            (:3616-3885)
  AmoebaStretchBendForce     CustomCompoundBondForce over every angle with parameters              (:4362-4503)
  PeriodicTorsionForce       as in forcefield.py of this package
  AmoebaPiTorsionForce       bonds between two trigonal atoms, six-atom CustomCompoundBondForce    (:3998-4116)
  AmoebaTorsionTorsionForce  five-atom chains a-b-c-d-e, chirality marker, spline grids            (:4121-4357)
  AmoebaUreyBradleyForce     a HarmonicBondForce between the end atoms of an angle (water H-H)     (:5622-5686)
  AmoebaVdwForce             per class sigma / epsilon / reduction, hydrogens reduced towards their heavy atom, 1-2 and 1-3 excluded (:4508-4670)
  AmoebaMultipoleForce       multipoles by type AND by the types of the frame-defining neighbours (four stages of the search), axis types
                             from the signs of kz / kx / ky, covalent maps 1-2 ... 1-5, polarization groups by pgrp types  (:4675-5282)

Where the reference's result depends on the iteration order of Python sets of atom indices (which of two equivalent hydrogens defines an
x axis), this reader takes the lowest index; the Systems are then physically equivalent, individual frames may be mirror choices of the
same definition.  The energy expressions of the Custom*Forces are the reference's strings (they ARE the force-field definition).
"""
import itertools
import math
import os
import xml.etree.ElementTree as ET
from collections import defaultdict

import numpy as np

from . import forcefield as FF

# AmoebaMultipoleForce::MultipoleAxisTypes
ZThenX, Bisector, ZBisect, ThreeFold, ZOnly, NoAxisType = range(6)
# AmoebaMultipoleForce::CovalentType
Covalent12, Covalent13, Covalent14, Covalent15, PolarizationCovalent11, PolarizationCovalent12, PolarizationCovalent13, PolarizationCovalent14 = range(8)

RAD_TO_DEG = 180.0 / math.pi


class AmoebaForceField:
    """The parameter tables of one AMOEBA force-field file, keyed by atom CLASS (valence terms, vdW) or TYPE (multipoles)."""

    def __init__(self, path, implicit_path=None):
        """path: the force-field file (amoeba2009.xml, amoeba2013.xml); implicit_path: its generalized-Kirkwood companion (amoeba2013_gk.xml)"""
        root = ET.parse(path).getroot()
        self.base = FF.ForceField(path)          # atom types, residue templates and torsions: the plain reader
        self.class_of = {name: t[0] for name, t in self.base.types.items()}
        e = root.find("AmoebaBondForce")
        self.bond_cubic, self.bond_quartic = e.attrib["bond-cubic"], e.attrib["bond-quartic"]
        self.bonds = {}
        for b in e.findall("Bond"):
            self.bonds[(b.attrib["class1"], b.attrib["class2"])] = (float(b.attrib["length"]), float(b.attrib["k"]))
        e = root.find("AmoebaAngleForce")
        self.angle_poly = tuple(e.attrib["angle-" + k] for k in ("cubic", "quartic", "pentic", "sextic"))
        self.angles = []          # in file order: the first match wins
        for a in e.findall("Angle"):
            values = [float(a.attrib[k]) for k in ("angle1", "angle2", "angle3") if k in a.attrib]
            self.angles.append(((a.attrib["class1"], a.attrib["class2"], a.attrib["class3"]), values, float(a.attrib["k"])))
        e = root.find("AmoebaOutOfPlaneBendForce")
        self.opbend_poly = tuple(float(e.attrib["opbend-" + k]) for k in ("cubic", "quartic", "pentic", "sextic"))
        self.opbends = [((a.attrib["class1"], a.attrib["class2"]), float(a.attrib["k"])) for a in e.findall("Angle")]      # (partner class, centre class)
        e = root.find("AmoebaStretchBendForce")
        self.stretch_bends = [((s.attrib["class1"], s.attrib["class2"], s.attrib["class3"]), float(s.attrib["k1"]), float(s.attrib["k2"])) for s in e.findall("StretchBend")]
        e = root.find("AmoebaPiTorsionForce")
        self.pi_torsions = [((p.attrib["class1"], p.attrib["class2"]), float(p.attrib["k"])) for p in e.findall("PiTorsion")]
        e = root.find("AmoebaTorsionTorsionForce")
        self.torsion_torsions = [(tuple(t.attrib["class%d" % i] for i in range(1, 6)), int(t.attrib["grid"])) for t in e.findall("TorsionTorsion")]
        self.tt_grids = {}
        for g in e.findall("TorsionTorsionGrid"):
            nx, ny = int(g.attrib["nx"]), int(g.attrib["ny"])
            entries = g.findall("Grid")
            # amoeba2009.xml tabulates only f: AmoebaTorsionTorsionForce then derives fx, fy, fxy by cubic splines (AmoebaTorsionTorsionForce.cpp:113)
            keys = ("angle1", "angle2", "f", "fx", "fy", "fxy") if "fx" in entries[0].attrib else ("angle1", "angle2", "f")
            rows = [[float(p.attrib[k]) for k in keys] for p in entries]
            # file order: angle1 runs fastest; the force wants grid[x][y] with x = the first angle
            arr = np.array(rows).reshape(ny, nx, len(keys)).transpose(1, 0, 2)
            self.tt_grids[int(g.attrib["grid"])] = np.ascontiguousarray(arr)
        e = root.find("AmoebaUreyBradleyForce")
        self.urey_bradleys = [((u.attrib["class1"], u.attrib["class2"], u.attrib["class3"]), float(u.attrib["d"]), float(u.attrib["k"])) for u in e.findall("UreyBradley")]
        e = root.find("AmoebaVdwForce")
        self.vdw_attrib = dict(e.attrib)
        self.vdw = {v.attrib["class"]: (float(v.attrib["sigma"]), float(v.attrib["epsilon"]), float(v.attrib["reduction"])) for v in e.findall("Vdw")}
        e = root.find("AmoebaMultipoleForce")
        self.multipoles = defaultdict(list)          # type -> definitions in file order
        for m in e.findall("Multipole"):
            k = [int(m.attrib[key]) if m.attrib.get(key) else 0 for key in ("kz", "kx", "ky")]
            q = {key: float(m.attrib[key]) for key in ("q11", "q21", "q22", "q31", "q32", "q33")}
            self.multipoles[m.attrib["type"]].append(dict(
                k=k, axis=axis_type(*k), charge=float(m.attrib["c0"]), dipole=[float(m.attrib["d%d" % i]) for i in (1, 2, 3)],
                quadrupole=[q["q11"], q["q21"], q["q31"], q["q21"], q["q22"], q["q32"], q["q31"], q["q32"], q["q33"]]))
        self.polarize = {}
        for p in e.findall("Polarize"):
            alpha, thole = float(p.attrib["polarizability"]), float(p.attrib["thole"])
            groups = {int(p.attrib["pgrp%d" % i]) for i in range(1, 7) if "pgrp%d" % i in p.attrib}
            self.polarize[p.attrib["type"]] = (alpha, thole, 0.0 if thole == 0 else alpha ** (1.0 / 6.0), groups)
        # amoeba2013.xml writes its torsions as <AmoebaTorsionForce> (three amplitude / phase pairs per definition) where amoeba2009.xml has a
        # <PeriodicTorsionForce>; AmoebaTorsionGenerator (forcefield.py:3890-3993) turns them into PeriodicTorsionForce terms all the same
        self.amoeba_torsions = None
        e = root.find("AmoebaTorsionForce")
        if e is not None:
            self.amoeba_torsions = [(tuple(t.attrib["class%d" % i] for i in range(1, 5)),
                                     [(i, float(t.attrib["angle%d" % i]), float(t.attrib["amp%d" % i])) for i in (1, 2, 3)]) for t in e.findall("Torsion")]
        # implicit solvent: AmoebaGeneralizedKirkwoodGenerator (forcefield.py:5359-5617) + AmoebaWcaDispersionGenerator (:5287-5354)
        self.gk = self.wca = None
        if implicit_path is not None:
            iroot = ET.parse(implicit_path).getroot()
            e = iroot.find("AmoebaGeneralizedKirkwoodForce")
            self.gk = {k: float(e.attrib[k]) for k in ("solventDielectric", "soluteDielectric", "includeCavityTerm", "probeRadius", "surfaceAreaFactor")}
            e = iroot.find("AmoebaWcaDispersionForce")
            self.wca = ({k: float(e.attrib[k]) for k in ("epso", "epsh", "rmino", "rminh", "awater", "slevy", "dispoff", "shctd")},
                        {w.attrib["class"]: (float(w.attrib["radius"]), float(w.attrib["epsilon"])) for w in e.findall("WcaDispersion")})


def axis_type(kz, kx, ky):
    """AmoebaMultipoleGenerator.setAxisType (forcefield.py:4689-4725): the frame convention from the signs of the neighbour types."""
    axis = ZThenX
    if kz == 0:
        axis = NoAxisType
    if kz != 0 and kx == 0:
        axis = ZOnly
    if kz < 0 or kx < 0:
        axis = Bisector
    if kx < 0 and ky < 0:
        axis = ZBisect
    if kz < 0 and kx < 0 and ky < 0:
        axis = ThreeFold
    return axis


def _first(table, key_forms):
    for entry in table:
        if entry[0] in key_forms:
            return entry
    return None


def create_description(pdb_path, ff, name):
    """-> dict of numpy arrays / lists describing every force of the System (see testsystems.AmoebaProteinWorkload.build)."""
    pdb = FF.read_pdb(pdb_path)
    n = len(pdb["names"])
    bonds = sorted(set(FF.perceive_bonds(pdb)))
    bonded = [[] for _ in range(n)]
    for a, b in bonds:
        bonded[a].append(b); bonded[b].append(a)
    for l in bonded:
        l.sort()
    res = pdb["resids"]
    starts = np.flatnonzero(np.diff(res, prepend=-1))
    ends = np.append(starts[1:], n)
    atom_type, template_names, cache = [None] * n, [], {}
    for s, e in zip(starts, ends):
        local = {i: i - s for i in range(s, e)}
        lb, ext = [], [0] * (e - s)
        for i in range(s, e):
            for j in bonded[i]:
                if s <= j < e:
                    if i < j:
                        lb.append((local[i], local[j]))
                else:
                    ext[local[i]] += 1
        t, m = ff.base.match_residue([pdb["elements"][i] for i in range(s, e)], lb, ext, cache)
        template_names.append(t["name"])
        for i in range(s, e):
            atom_type[i] = t["types"][m[local[i]]]
    cls = [ff.class_of[t] for t in atom_type]
    masses = np.array([ff.base.types[t][2] for t in atom_type])
    d = dict(name=name, positions=pdb["positions"], box=np.diag(pdb["box"] if pdb["box"] is not None else [2.0, 2.0, 2.0]), masses=masses, atom_type=np.array([int(t) for t in atom_type]),
             template_names=template_names, bonds_topology=np.array(bonds, dtype=np.int64))

    # ---- bonds
    bond_length = {}
    b_atoms, b_par = [], []
    for a, b in bonds:
        p = ff.bonds.get((cls[a], cls[b])) or ff.bonds.get((cls[b], cls[a]))
        if p is None:
            raise KeyError("no AMOEBA bond parameters for classes %s-%s" % (cls[a], cls[b]))
        bond_length[(a, b)] = bond_length[(b, a)] = p[0]
        if p[1] != 0:
            b_atoms.append((a, b)); b_par.append(p)
    d["bonds"] = (np.array(b_atoms, dtype=np.int32).reshape(-1, 2), np.array(b_par).reshape(-1, 2), ff.bond_cubic, ff.bond_quartic)

    # ---- angles, partitioned by the out-of-plane-bend rule: a centre with exactly three bonds ALL of which have out-of-plane parameters
    #      gets three out-of-plane bends and its angles become in-plane angles (the fourth atom = the centre's remaining partner)
    angles = [(i, j, k) for j in range(n) for i, k in itertools.combinations(bonded[j], 2)]
    op_atoms, op_k = [], []
    trigonal = {}
    for j in range(n):
        if len(bonded[j]) != 3:
            continue
        ks = []
        for p in bonded[j]:
            hit = _first(ff.opbends, ((cls[p], cls[j]),))
            if hit is None:
                break
            ks.append(hit[1])
        if len(ks) == 3:
            p0, p1, p2 = bonded[j]
            # (atom whose out-of-plane motion is measured is listed last; its own parameter applies: forcefield.py:3781-3783)
            op_atoms += [(p0, j, p1, p2), (p0, j, p2, p1), (p1, j, p2, p0)]
            op_k += [ks[2], ks[1], ks[0]]
            trigonal[j] = (p0, p1, p2)
    d["opbends"] = (np.array(op_atoms, dtype=np.int32).reshape(-1, 4), np.array(op_k), ff.opbend_poly)
    a_atoms, a_par, ip_atoms, ip_par = [], [], [], []
    ideal = {}
    for i, j, k in angles:
        hit = _first(ff.angles, ((cls[i], cls[j], cls[k]), (cls[k], cls[j], cls[i])))
        if hit is None:
            continue
        _, values, kf = hit
        if j in trigonal:
            ideal[(i, j, k)] = values[0]
            if kf != 0:
                fourth = [p for p in trigonal[j] if p != i and p != k][0]
                ip_atoms.append((i, j, k, fourth)); ip_par.append((values[0], kf))
            continue
        if kf == 0:
            continue
        theta = values[0]
        if len(values) > 1:
            # the ideal angle depends on how many hydrogens the centre carries besides the two angle atoms (Tinker's kangle.f; forcefield.py:3539-3556)
            nh = sum(1 for p in bonded[j] if p != i and p != k and masses[p] < 1.9)
            if nh >= len(values):
                raise ValueError("angle %d-%d-%d: %d hydrogens on the centre, %d angle values" % (i, j, k, nh, len(values)))
            theta = values[nh]
        ideal[(i, j, k)] = theta
        a_atoms.append((i, j, k)); a_par.append((theta, kf))
    d["angles"] = (np.array(a_atoms, dtype=np.int32).reshape(-1, 3), np.array(a_par).reshape(-1, 2), ff.angle_poly)
    d["inplane_angles"] = (np.array(ip_atoms, dtype=np.int32).reshape(-1, 4), np.array(ip_par).reshape(-1, 2))

    # ---- stretch-bend: every angle with an ideal value (the class of the first angle atom may match class1 or class3: k1 stays with the first bond, forcefield.py:4455)
    sb_atoms, sb_par = [], []
    for (i, j, k), theta in ideal.items():
        hit = None
        for classes, k1, k2 in ff.stretch_bends:
            if cls[j] == classes[1] and ((cls[i] == classes[0] and cls[k] == classes[2]) or (cls[k] == classes[0] and cls[i] == classes[2])):
                hit = (k1, k2)
                break
        if hit is not None:
            sb_atoms.append((i, j, k)); sb_par.append((bond_length[(i, j)], bond_length[(k, j)], theta / RAD_TO_DEG, hit[0], hit[1]))
    d["stretch_bends"] = (np.array(sb_atoms, dtype=np.int32).reshape(-1, 3), np.array(sb_par).reshape(-1, 5))

    # ---- Urey-Bradley (HarmonicBondForce with 2 k, forcefield.py:5682)
    ub_atoms, ub_par = [], []
    for i, j, k in angles:
        hit = _first(ff.urey_bradleys, ((cls[i], cls[j], cls[k]), (cls[k], cls[j], cls[i])))
        if hit is not None:
            ub_atoms.append((i, k)); ub_par.append((hit[1], 2.0 * hit[2]))
    d["urey_bradleys"] = (np.array(ub_atoms, dtype=np.int32).reshape(-1, 2), np.array(ub_par).reshape(-1, 2))

    # ---- proper torsions (PeriodicTorsionForce; the matching rules of forcefield.py of this package: first definition without wildcards, else the first)
    propers = set()
    for i, j, k in angles:
        for x in bonded[i]:
            if x not in (i, j, k):
                propers.add((x, i, j, k) if x < k else (k, j, i, x))
        for x in bonded[k]:
            if x not in (i, j, k):
                propers.add((i, j, k, x) if x > i else (x, k, j, i))
    t_atoms, t_par, cache_p = [], [], {}
    for tor in sorted(propers):
        c = tuple(cls[x] for x in tor)
        key = min(c, c[::-1])
        if key not in cache_p and ff.amoeba_torsions is not None:
            # AmoebaTorsionGenerator.createForce (forcefield.py:3966-3991): the FIRST definition that fits forwards or backwards
            match = None
            for classes, terms in ff.amoeba_torsions:
                if all(classes[q] in ("", c[q]) for q in range(4)) or all(classes[q] in ("", c[3 - q]) for q in range(4)):
                    match = terms
                    break
            cache_p[key] = match
        if key not in cache_p:
            match = None
            for classes, terms in ff.base.propers:
                fwd = all(classes[q] in ("", c[q]) for q in range(4))
                rev = all(classes[q] in ("", c[3 - q]) for q in range(4))
                if fwd or rev:
                    wild = "" in classes
                    if match is None or not wild:
                        match = terms
                    if not wild:
                        break
            cache_p[key] = match
        for per, phase, kf in cache_p[key] or ():
            if kf != 0:
                t_atoms.append(tor); t_par.append((per, phase, kf))
    d["torsions"] = (np.array(t_atoms, dtype=np.int32).reshape(-1, 4), np.array(t_par).reshape(-1, 3))

    # ---- pi-torsions: a bond between two trigonal atoms with parameters for the class pair
    pt_atoms, pt_k = [], []
    for a, b in bonds:
        if len(bonded[a]) == 3 and len(bonded[b]) == 3:
            hit = _first(ff.pi_torsions, ((cls[a], cls[b]), (cls[b], cls[a])))
            if hit is not None:
                oa = [p for p in bonded[a] if p != b]
                ob = [p for p in bonded[b] if p != a]
                pt_atoms.append((oa[0], oa[1], a, b, ob[0], ob[1])); pt_k.append(hit[1])
    d["pi_torsions"] = (np.array(pt_atoms, dtype=np.int32).reshape(-1, 6), np.array(pt_k))

    # ---- torsion-torsions: chains a-b-c-d-e over every angle b-c-d (Tinker's bitors), types in order or reversed
    tt_atoms, tt_grid = [], []
    for ib, ic, id_ in angles:
        for ia in bonded[ib]:
            if ia in (ic, id_):
                continue
            for ie in bonded[id_]:
                if ie in (ic, ib, ia):
                    continue
                c5 = (cls[ia], cls[ib], cls[ic], cls[id_], cls[ie])
                for classes, grid in ff.torsion_torsions:
                    if c5 == classes:
                        tt_atoms.append((ia, ib, ic, id_, ie, _chiral_marker(ib, ic, id_, bonded, d["atom_type"]))); tt_grid.append(grid)
                    elif c5[::-1] == classes:
                        tt_atoms.append((ie, id_, ic, ib, ia, _chiral_marker(ib, ic, id_, bonded, d["atom_type"]))); tt_grid.append(grid)
    d["torsion_torsions"] = (np.array(tt_atoms, dtype=np.int32).reshape(-1, 6), np.array(tt_grid, dtype=np.int32), ff.tt_grids)

    # ---- covalent neighbourhoods
    b12 = [set(l) for l in bonded]

    def next_shell(prev, *inner):
        out = []
        for i in range(n):
            s = set()
            for j in prev[i]:
                s |= b12[j]
            s.discard(i)
            for shell in inner:
                s -= shell[i]
            out.append(s)
        return out
    b13 = next_shell(b12, b12)
    b14 = next_shell(b13, b12, b13)
    b15 = next_shell(b14, b12, b13, b14)

    # ---- vdW
    va = ff.vdw_attrib
    scale = 1.0
    if va["radiustype"] == "SIGMA":
        scale = 1.122462048309372
    if va["radiussize"] == "DIAMETER":
        scale = 0.5
    parent = np.arange(n, dtype=np.int32)
    for i in range(n):
        if pdb["elements"][i] == "H" and len(bonded[i]) == 1:
            parent[i] = bonded[i][0]
    vp = np.array([ff.vdw[c] for c in cls])
    excl = []
    for i in range(n):
        s = set(b12[i])
        if float(va["vdw-13-scale"]) == 0.0:
            s |= b13[i]
        s.add(i)
        excl.append(sorted(s))
    d["vdw"] = dict(parent=parent, sigma=vp[:, 0] * scale, epsilon=vp[:, 1], reduction=vp[:, 2], exclusions=excl, sigma_rule=va["radiusrule"], epsilon_rule=va["epsilonrule"],
                    potential=va["type"])

    # ---- multipoles: for every atom the first definition of its type whose frame-defining neighbour TYPES are found, searching in four stages
    ty = d["atom_type"]
    axes = np.full((n, 4), -1, dtype=np.int32)
    chosen = [None] * n
    for i in range(n):
        defs = ff.multipoles.get(str(ty[i]))
        if not defs:
            raise KeyError("no multipole definition for atom type %d" % ty[i])
        hit = None
        n12 = sorted(b12[i])
        n13 = sorted(b13[i])
        # stage 1: z and x (and y) among the 1-2 neighbours
        for m in defs:
            kz, kx, ky = (abs(v) for v in m["k"])
            if kz == 0 or kx == 0:
                continue
            for z in n12:
                if ty[z] != kz:
                    continue
                xs = [x for x in n12 if x != z and ty[x] == kx]
                if not xs:
                    continue
                if ky == 0:
                    x = xs[0]
                    if ty[x] == ty[z] and x < z:
                        z, x = x, z
                    hit = (m, z, x, -1)
                else:
                    for x in xs:
                        ys = [y for y in n12 if y not in (z, x) and ty[y] == ky]
                        if ys:
                            hit = (m, z, x, ys[0])
                            break
                if hit:
                    break
            if hit:
                break
        # stage 2: z among the 1-2 neighbours, x (and y) among the 1-3 neighbours bonded to z
        if hit is None:
            for m in defs:
                kz, kx, ky = (abs(v) for v in m["k"])
                if kz == 0 or kx == 0:
                    continue
                for z in n12:
                    if ty[z] != kz:
                        continue
                    xs = [x for x in n13 if ty[x] == kx and z in b12[x]]
                    if not xs:
                        continue
                    if ky == 0:
                        hit = (m, z, xs[0], -1)
                    else:
                        for x in xs:
                            ys = [y for y in n13 if y != x and ty[y] == ky and z in b12[y]]
                            if ys:
                                hit = (m, z, x, ys[0])
                                break
                    if hit:
                        break
                if hit:
                    break
        # stage 3: only a z-defining atom; stage 4: no frame at all
        if hit is None:
            for m in defs:
                kz, kx, _ = (abs(v) for v in m["k"])
                zs = [z for z in n12 if kx == 0 and kz != 0 and ty[z] == kz]
                if zs:
                    hit = (m, zs[0], -1, -1)
                    break
        if hit is None:
            for m in defs:
                if m["k"][0] == 0:
                    hit = (m, -1, -1, -1)
                    break
        if hit is None:
            raise ValueError("atom %d (%s of %s, type %d): no multipole definition fits its neighbours" % (i, pdb["names"][i], pdb["resnames"][i], ty[i]))
        chosen[i] = hit[0]
        axes[i] = (hit[0]["axis"], hit[1], hit[2], hit[3])
    pol = [ff.polarize[str(t)] for t in ty]
    # polarization groups: atoms joined through bonds to neighbours whose TYPE is listed in the atom's pgrp attributes (either direction)
    link = [set([i]) for i in range(n)]
    for i in range(n):
        for j in b12[i]:
            if int(ty[j]) in pol[i][3]:
                link[i].add(j); link[j].add(i)
    group_of = [None] * n
    for i in range(n):
        if group_of[i] is not None:
            continue
        group, stack = set(), [i]
        while stack:
            a = stack.pop()
            if a in group:
                continue
            group.add(a)
            stack.extend(link[a] - group)
        g = sorted(group)
        for a in g:
            group_of[a] = g
    p11 = [set(group_of[i]) for i in range(n)]

    def next_groups(prev, *inner):
        out = []
        for i in range(n):
            s = set()
            for a in prev[i]:
                for b in b12[a]:
                    s |= p11[b]
            for shell in inner:
                s -= shell[i]
            out.append(s)
        return out
    p12 = next_groups(p11, p11)
    p13 = next_groups(p12, p11, p12)
    p14 = next_groups(p13, p11, p12, p13)
    maps = []
    for i in range(n):
        for kind, sets in ((Covalent12, b12), (Covalent13, b13), (Covalent14, b14), (Covalent15, b15),
                           (PolarizationCovalent11, p11), (PolarizationCovalent12, p12), (PolarizationCovalent13, p13), (PolarizationCovalent14, p14)):
            maps.append((i, kind, sorted(sets[i])))
    if ff.gk is not None:
        # AmoebaGeneralizedKirkwoodGenerator.createForce: the multipole's charge, the Bondi radius of the element x 1.03, overlap scale 0.69
        bondi = {"H": 0.12, "He": 0.14, "B": 0.18, "C": 0.170, "N": 0.155, "O": 0.152, "F": 0.147, "Ne": 0.154, "Si": 0.210, "P": 0.180, "S": 0.180, "Cl": 0.175}
        d["gk"] = dict(ff.gk, charge=np.array([m["charge"] for m in chosen]), radius=np.array([bondi[e] * 1.03 for e in pdb["elements"]]), scale=np.full(n, 0.69))
        globals_, table = ff.wca
        d["wca"] = dict(globals_, radius=np.array([table[c][0] for c in cls]), epsilon=np.array([table[c][1] for c in cls]))
    d["multipoles"] = dict(charge=np.array([m["charge"] for m in chosen]), dipole=np.array([m["dipole"] for m in chosen]), quadrupole=np.array([m["quadrupole"] for m in chosen]),
                           axes=axes, thole=np.array([p[1] for p in pol]), damping=np.array([p[2] for p in pol]), polarity=np.array([p[0] for p in pol]), covalent_maps=maps)
    return d


def _chiral_marker(ib, ic, id_, bonded, atom_type):
    """AmoebaTorsionTorsionGenerator.getChiralAtomIndex (forcefield.py:4216-4262): when the central atom has four bonds, the one of its two
    other partners with the higher atom TYPE (the mass comparison that follows in the reference compares an atom with itself), else -1."""
    if len(bonded[ic]) != 4:
        return -1
    others = [p for p in bonded[ic] if p != ib and p != id_]
    if len(others) != 2:
        return -1
    e, f = others
    if atom_type[e] > atom_type[f]:
        return e
    if atom_type[f] > atom_type[e]:
        return f
    return -1


def dhfr(data_dir=None, pdb_path="/root/reference/examples/5dfr_solv-cube_equil.pdb"):
    """The `amoebapme` test of examples/benchmark.py:58-68: DHFR in water, amoeba2009.xml."""
    data_dir = data_dir or next((p for p in FF.DATA_DIR_CANDIDATES if os.path.isdir(p)), None)
    ff = AmoebaForceField(os.path.join(data_dir, "amoeba2009.xml"))
    return create_description(pdb_path, ff, "dhfr-23558 (5dfr_solv-cube_equil.pdb, amoeba2009)")


# ------------------------------------------------------------------------------------------------ fixtures
def _csr(lists):
    start = np.zeros(len(lists) + 1, dtype=np.int64)
    start[1:] = np.cumsum([len(l) for l in lists])
    flat = np.concatenate([np.asarray(l, dtype=np.int32) for l in lists]) if start[-1] else np.zeros(0, dtype=np.int32)
    return start, flat.astype(np.int32)


def save_description(d, path):
    """The description as one compressed .npz (lists of index lists as CSR)."""
    out = dict(name=np.array(d["name"]), positions=d["positions"], box=d["box"], masses=d["masses"], atom_type=d["atom_type"], bonds_topology=d["bonds_topology"])
    out["bonds_atoms"], out["bonds_par"] = d["bonds"][0], d["bonds"][1]
    out["bonds_poly"] = np.array([d["bonds"][2], d["bonds"][3]])
    out["angles_atoms"], out["angles_par"] = d["angles"][0], d["angles"][1]
    out["angles_poly"] = np.array(d["angles"][2])
    out["inplane_atoms"], out["inplane_par"] = d["inplane_angles"]
    out["opbend_atoms"], out["opbend_k"] = d["opbends"][0], d["opbends"][1]
    out["opbend_poly"] = np.array(d["opbends"][2])
    out["sb_atoms"], out["sb_par"] = d["stretch_bends"]
    out["ub_atoms"], out["ub_par"] = d["urey_bradleys"]
    out["torsion_atoms"], out["torsion_par"] = d["torsions"]
    out["pitors_atoms"], out["pitors_k"] = d["pi_torsions"]
    out["tt_atoms"], out["tt_grid_index"] = d["torsion_torsions"][0], d["torsion_torsions"][1]
    for index, grid in d["torsion_torsions"][2].items():
        out["tt_grid_%d" % index] = grid
    v = d["vdw"]
    for k in ("parent", "sigma", "epsilon", "reduction"):
        out["vdw_" + k] = v[k]
    out["vdw_rules"] = np.array([v["sigma_rule"], v["epsilon_rule"], v["potential"]])
    out["vdw_excl_start"], out["vdw_excl"] = _csr(v["exclusions"])
    m = d["multipoles"]
    for k in ("charge", "dipole", "quadrupole", "axes", "thole", "damping", "polarity"):
        out["mp_" + k] = m[k]
    out["mp_map_atom"] = np.array([e[0] for e in m["covalent_maps"]], dtype=np.int32)
    out["mp_map_kind"] = np.array([e[1] for e in m["covalent_maps"]], dtype=np.int8)
    out["mp_map_start"], out["mp_map_list"] = _csr([e[2] for e in m["covalent_maps"]])
    np.savez_compressed(path, **out)


def load_description(path):
    z = np.load(path)

    def lists(start, flat):
        return [flat[start[i]:start[i + 1]] for i in range(len(start) - 1)]
    d = dict(name=str(z["name"]), positions=z["positions"], box=z["box"], masses=z["masses"], atom_type=z["atom_type"], bonds_topology=z["bonds_topology"])
    d["bonds"] = (z["bonds_atoms"], z["bonds_par"], str(z["bonds_poly"][0]), str(z["bonds_poly"][1]))
    d["angles"] = (z["angles_atoms"], z["angles_par"], tuple(str(v) for v in z["angles_poly"]))
    d["inplane_angles"] = (z["inplane_atoms"], z["inplane_par"])
    d["opbends"] = (z["opbend_atoms"], z["opbend_k"], tuple(float(v) for v in z["opbend_poly"]))
    d["stretch_bends"] = (z["sb_atoms"], z["sb_par"])
    d["urey_bradleys"] = (z["ub_atoms"], z["ub_par"])
    d["torsions"] = (z["torsion_atoms"], z["torsion_par"])
    d["pi_torsions"] = (z["pitors_atoms"], z["pitors_k"])
    d["torsion_torsions"] = (z["tt_atoms"], z["tt_grid_index"], {int(k[8:]): z[k] for k in z.files if k.startswith("tt_grid_") and k != "tt_grid_index"})
    rules = [str(v) for v in z["vdw_rules"]]
    d["vdw"] = dict(parent=z["vdw_parent"], sigma=z["vdw_sigma"], epsilon=z["vdw_epsilon"], reduction=z["vdw_reduction"], sigma_rule=rules[0], epsilon_rule=rules[1], potential=rules[2],
                    exclusions=lists(z["vdw_excl_start"], z["vdw_excl"]))
    maps = list(zip(z["mp_map_atom"].tolist(), z["mp_map_kind"].tolist(), lists(z["mp_map_start"], z["mp_map_list"])))
    d["multipoles"] = dict(charge=z["mp_charge"], dipole=z["mp_dipole"], quadrupole=z["mp_quadrupole"], axes=z["mp_axes"], thole=z["mp_thole"], damping=z["mp_damping"],
                           polarity=z["mp_polarity"], covalent_maps=maps)
    return d


def subset(d, num_atoms):
    """The first `num_atoms` atoms of a description with every term among them (cut at a molecule boundary: the solute without the water,
    say) -- a small System with every kind of term for tests."""
    def keep(atoms, *rest):
        atoms = np.asarray(atoms)
        m = (atoms < num_atoms).all(axis=1) if len(atoms) else np.zeros(0, dtype=bool)
        # (the chirality marker of a torsion-torsion is -1 or an atom)
        return (atoms[m],) + tuple(np.asarray(r)[m] for r in rest)
    n = num_atoms
    out = dict(name="%s, first %d atoms" % (d["name"], n), positions=d["positions"][:n], box=d["box"], masses=d["masses"][:n], atom_type=d["atom_type"][:n],
               bonds_topology=keep(d["bonds_topology"])[0])
    out["bonds"] = keep(d["bonds"][0], d["bonds"][1]) + tuple(d["bonds"][2:])
    out["angles"] = keep(d["angles"][0], d["angles"][1]) + (d["angles"][2],)
    out["inplane_angles"] = keep(*d["inplane_angles"])
    out["opbends"] = keep(d["opbends"][0], d["opbends"][1]) + (d["opbends"][2],)
    for k in ("stretch_bends", "urey_bradleys", "torsions", "pi_torsions"):
        out[k] = keep(*d[k])
    out["torsion_torsions"] = keep(d["torsion_torsions"][0], d["torsion_torsions"][1]) + (d["torsion_torsions"][2],)
    v = d["vdw"]
    out["vdw"] = dict(v, parent=v["parent"][:n], sigma=v["sigma"][:n], epsilon=v["epsilon"][:n], reduction=v["reduction"][:n], exclusions=v["exclusions"][:n])
    m = d["multipoles"]
    out["multipoles"] = dict({k: m[k][:n] for k in ("charge", "dipole", "quadrupole", "axes", "thole", "damping", "polarity")},
                             covalent_maps=[e for e in m["covalent_maps"] if e[0] < n])
    for lists in (out["vdw"]["exclusions"], [e[2] for e in out["multipoles"]["covalent_maps"]]):
        for l in lists:
            if len(l) and max(l) >= n:
                raise ValueError("the cut at atom %d goes through a molecule" % n)
    if (out["multipoles"]["axes"][:, 1:] >= n).any() or (out["vdw"]["parent"] >= n).any():
        raise ValueError("the cut at atom %d goes through a molecule" % n)
    return out


def alanine_dipeptide_implicit(data_dir=None, pdb_path="/root/reference/wrappers/python/tests/systems/alanine-dipeptide-implicit.pdb"):
    """The System of wrappers/python/tests/TestForceField.py:1246-1262 (AmoebaTestForceField.test_Forces): alanine dipeptide,
    ForceField('amoeba2013.xml', 'amoeba2013_gk.xml').createSystem(topology, polarization='direct') -- NoCutoff, generalized Kirkwood +
    WCA dispersion.  The reference keeps golden forces of it (tests/systems/alanine-dipeptide-amoeba-forces.xml): the pin of this reader."""
    data_dir = data_dir or next((p for p in FF.DATA_DIR_CANDIDATES if os.path.isdir(p)), None)
    ff = AmoebaForceField(os.path.join(data_dir, "amoeba2013.xml"), os.path.join(data_dir, "amoeba2013_gk.xml"))
    return create_description(pdb_path, ff, "alanine-dipeptide-implicit (amoeba2013 + amoeba2013_gk)")
