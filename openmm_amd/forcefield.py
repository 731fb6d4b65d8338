"""A small, dependency-free reader of OpenMM force-field XML (the subset amber99sb.xml + tip3p.xml use) that turns a PDB
file into a testsystems.Workload -- the real DHFR benchmark System of examples/benchmark.py without the SWIG Python layer
(SURVEY.md §8f-2).  Pure Python + numpy; nothing of the `openmm` package is imported.

It restates the behaviour of wrappers/python/openmm/app/forcefield.py (ForceField.createSystem) for what that file does on
this input:
  * residue templates are matched to residues as graphs -- elements, bonds, number of external bonds (forcefield.py:_matchResidue);
  * HarmonicBond / HarmonicAngle parameters by atom class (forcefield.py:1989-2110); bonds to hydrogen become constraints with
    constraints=HBonds and are left out of the bond force, rigid water gets its H-H constraint from the angle (forcefield.py:1302-1340);
  * proper torsions: the first definition without wildcards, otherwise the first one with (PeriodicTorsionGenerator.createForce,
    forcefield.py:2156-2192); impropers with the "default" ordering rule of _matchImproper (forcefield.py:1835-1866);
  * NonbondedForce: per-type charge / sigma / epsilon, exceptions from the bond graph with the 1-4 scale factors of the file
    (NonbondedGenerator, forcefield.py:2347-2445), dispersion correction on;
  * CMMotionRemover (removeCMMotion=True is the default of createSystem).
Bonds are perceived from the coordinates (covalent-radius test inside a residue, C-N links between consecutive residues); that
every residue then matches exactly one amber99sb template is the check that the perception was right.  Where the reference
picks among chemically equivalent atoms by the order of its topology's bond list (the two oxygens of a carboxylate in an
improper), this reader may pick the other one -- the System is equally valid, individual improper angles can differ.
"""
import itertools
import os
import xml.etree.ElementTree as ET
from collections import defaultdict

import numpy as np

DATA_DIR_CANDIDATES = ("/root/reference/wrappers/python/openmm/app/data",)
ELEMENT_MASS = {"H": 1.007947, "C": 12.01078, "N": 14.00672, "O": 15.99943, "S": 32.0655}
COVALENT_RADIUS = {"H": 0.031, "C": 0.076, "N": 0.071, "O": 0.066, "S": 0.105}      # nm


# ------------------------------------------------------------------------------------------------ PDB
def read_pdb(path):
    """-> dict(names, resnames, resids (sequential residue index), elements, positions [nm], box [3 lengths, nm])"""
    names, resnames, resids, pos = [], [], [], []
    box = None
    last, index = None, -1
    for line in open(path):
        if line.startswith("CRYST1"):
            box = np.array([float(line[6:15]), float(line[15:24]), float(line[24:33])]) * 0.1
        elif line.startswith(("ATOM", "HETATM")):
            key = (line[17:21].strip(), line[21], line[22:27], line[72:76].strip())
            if key != last:
                index += 1
                last = key
            names.append(line[12:16].strip())
            resnames.append(line[17:21].strip())
            resids.append(index)
            pos.append((float(line[30:38]), float(line[38:46]), float(line[46:54])))
    elements = [n.lstrip("0123456789")[0] for n in names]
    return dict(names=names, resnames=resnames, resids=np.array(resids), elements=elements, positions=np.array(pos) * 0.1, box=box)


def perceive_bonds(pdb):
    """Covalent bonds from geometry: inside a residue every pair closer than 1.3 x the sum of the covalent radii (hydrogens bond
    only to their nearest heavy atom); between consecutive non-water residues the peptide C-N link."""
    pos, el, res = pdb["positions"], pdb["elements"], pdb["resids"]
    bonds = []
    starts = np.flatnonzero(np.diff(res, prepend=-1))
    ends = np.append(starts[1:], len(res))
    prev_c = None
    for s, e in zip(starts, ends):
        idx = np.arange(s, e)
        water = pdb["resnames"][s] in ("HOH", "WAT", "TIP3")
        heavy = [i for i in idx if el[i] != "H"]
        hyd = [i for i in idx if el[i] == "H"]
        for a, b in itertools.combinations(heavy, 2):
            if np.linalg.norm(pos[a] - pos[b]) < 1.3 * (COVALENT_RADIUS[el[a]] + COVALENT_RADIUS[el[b]]):
                bonds.append((a, b))
        for h in hyd:
            d = [np.linalg.norm(pos[h] - pos[x]) for x in heavy]
            bonds.append((heavy[int(np.argmin(d))], h))
        if water:
            prev_c = None
            continue
        n_atoms = [i for i in idx if pdb["names"][i] == "N"]
        if prev_c is not None and n_atoms and np.linalg.norm(pos[prev_c] - pos[n_atoms[0]]) < 0.17:
            bonds.append((prev_c, n_atoms[0]))
        c_atoms = [i for i in idx if pdb["names"][i] == "C"]
        prev_c = c_atoms[0] if c_atoms else None
    return [(min(a, b), max(a, b)) for a, b in bonds]


# ------------------------------------------------------------------------------------------------ force field files
class ForceField:
    def __init__(self, *files, data_dir=None):
        data_dir = data_dir or next((d for d in DATA_DIR_CANDIDATES if os.path.isdir(d)), None)
        self.types = {}              # type name -> (class, element, mass)
        self.templates = []          # dict(name, types[], elements[], bonds[(i, j)], external[count per atom])
        self.bonds = {}              # (class, class) -> (length, k)
        self.angles = {}             # (c1, c2, c3) -> (angle, k)
        self.propers, self.impropers = [], []      # (classes[4], [(periodicity, phase, k)])
        self.nonbonded = {}          # type -> (charge, sigma, epsilon)
        self.coulomb14, self.lj14 = 1.0, 1.0
        self.gbsa = {}               # type -> (charge, radius, scale)   (<GBSAOBCForce>, e.g. amber99_obc.xml)
        for f in files:
            self._load(f if os.path.isabs(f) else os.path.join(data_dir, f))

    def _load(self, path):
        root = ET.parse(path).getroot()
        for t in root.findall("AtomTypes/Type"):
            self.types[t.attrib["name"]] = (t.attrib["class"], t.attrib.get("element"), float(t.attrib["mass"]))
        for r in root.findall("Residues/Residue"):
            atoms = r.findall("Atom")
            names = [a.attrib["name"] for a in atoms]
            types = [a.attrib["type"] for a in atoms]

            def index(b, key_i, key_n):
                return int(b.attrib[key_i]) if key_i in b.attrib else names.index(b.attrib[key_n])
            bonds = [(index(b, "from", "atomName1"), index(b, "to", "atomName2")) for b in r.findall("Bond")]
            external = [0] * len(atoms)
            for b in r.findall("ExternalBond"):
                external[index(b, "from", "atomName")] += 1
            self.templates.append(dict(name=r.attrib["name"], names=names, types=types, elements=[self.types[t][1] for t in types],
                                       bonds=bonds, external=external))
        for b in root.findall("HarmonicBondForce/Bond"):
            self.bonds[(b.attrib["class1"], b.attrib["class2"])] = (float(b.attrib["length"]), float(b.attrib["k"]))
        for a in root.findall("HarmonicAngleForce/Angle"):
            self.angles[(a.attrib["class1"], a.attrib["class2"], a.attrib["class3"])] = (float(a.attrib["angle"]), float(a.attrib["k"]))
        for kind, store in (("Proper", self.propers), ("Improper", self.impropers)):
            for t in root.findall("PeriodicTorsionForce/" + kind):
                classes = [t.attrib["class%d" % i] for i in range(1, 5)]
                terms, i = [], 1
                while "phase%d" % i in t.attrib:
                    terms.append((int(t.attrib["periodicity%d" % i]), float(t.attrib["phase%d" % i]), float(t.attrib["k%d" % i])))
                    i += 1
                store.append((classes, terms))
        nb = root.find("NonbondedForce")
        if nb is not None:
            self.coulomb14, self.lj14 = float(nb.attrib["coulomb14scale"]), float(nb.attrib["lj14scale"])
            for a in nb.findall("Atom"):
                self.nonbonded[a.attrib["type"]] = (float(a.attrib["charge"]), float(a.attrib["sigma"]), float(a.attrib["epsilon"]))
        for a in root.findall("GBSAOBCForce/Atom"):                   # GBSAOBCGenerator (forcefield.py:2676-2723)
            self.gbsa[a.attrib["type"]] = (float(a.attrib["charge"]), float(a.attrib["radius"]), float(a.attrib["scale"]))

    # ---- residue <-> template graph matching (forcefield.py:_matchResidue)
    def match_residue(self, elements, bonds, external, cache):
        """elements[n], bonds [(i, j)] local indices, external[n] -> (template, mapping residue atom -> template atom)"""
        sig = (tuple(sorted(elements)), len(bonds), tuple(sorted(external)))
        matches = []
        for t in cache.setdefault("by_signature", {}).get(sig, None) or self._templates_with_signature(sig, cache):
            m = _match_graph(elements, bonds, external, t)
            if m is not None:
                matches.append((t, m))
        if len(matches) != 1:
            raise ValueError("residue matches %d templates (%s)" % (len(matches), ", ".join(t["name"] for t, _ in matches)))
        return matches[0]

    def _templates_with_signature(self, sig, cache):
        out = [t for t in self.templates if (tuple(sorted(t["elements"])), len(t["bonds"]), tuple(sorted(t["external"]))) == sig]
        cache["by_signature"][sig] = out
        return out


def _match_graph(elements, bonds, external, template):
    n = len(elements)
    adj = [set() for _ in range(n)]
    for a, b in bonds:
        adj[a].add(b); adj[b].add(a)
    tadj = [set() for _ in range(n)]
    for a, b in template["bonds"]:
        tadj[a].add(b); tadj[b].add(a)
    cand = [[j for j in range(n) if template["elements"][j] == elements[i] and len(tadj[j]) == len(adj[i]) and template["external"][j] == external[i]]
            for i in range(n)]
    order = sorted(range(n), key=lambda i: len(cand[i]))
    # visit atoms so that each (after the first) is bonded to an earlier one where possible: prunes early
    seen, ordered = set(), []
    for start in order:
        if start in seen:
            continue
        stack = [start]
        while stack:
            i = stack.pop()
            if i in seen:
                continue
            seen.add(i); ordered.append(i)
            stack.extend(sorted(adj[i] - seen, key=lambda x: -len(cand[x])))
    mapping, used = {}, set()

    def place(k):
        if k == n:
            return True
        i = ordered[k]
        for j in cand[i]:
            if j in used:
                continue
            if all((mapping[x] in tadj[j]) for x in adj[i] if x in mapping):
                mapping[i] = j; used.add(j)
                if place(k + 1):
                    return True
                del mapping[i]; used.discard(j)
        return False
    return dict(mapping) if place(0) else None


# ------------------------------------------------------------------------------------------------ System
def create_workload(pdb_path, ff, name, cutoff=0.9, constraints_hbonds=True, rigid_water=True, method=4):
    """-> testsystems.Workload with the arrays ForceField.createSystem(method, cutoff, HBonds / None) would put into the System.
    method: harness.NonbondedForce constants (0 NoCutoff ... 4 PME); a force field with a <GBSAOBCForce> section adds `w.gbsa`."""
    from .testsystems import Workload
    pdb = read_pdb(pdb_path)
    n = len(pdb["names"])
    bonds = sorted(set(perceive_bonds(pdb)))
    bonded = [[] for _ in range(n)]
    for a, b in bonds:
        bonded[a].append(b); bonded[b].append(a)
    res = pdb["resids"]
    # ---- atom types by template matching
    atom_type = [None] * n
    starts = np.flatnonzero(np.diff(res, prepend=-1))
    ends = np.append(starts[1:], n)
    cache = {}
    is_water = np.zeros(n, bool)
    template_names = []
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
        t, m = ff.match_residue([pdb["elements"][i] for i in range(s, e)], lb, ext, cache)
        template_names.append(t["name"])
        for i in range(s, e):
            atom_type[i] = t["types"][m[local[i]]]
        if t["name"] == "HOH":
            is_water[s:e] = True
    cls = [ff.types[t][0] for t in atom_type]
    el = pdb["elements"]
    w = Workload(name)
    w.template_names = template_names
    w.positions = pdb["positions"]
    w.box = np.diag(pdb["box"]) if pdb["box"] is not None else None
    w.masses = np.array([ff.types[t][2] for t in atom_type])
    w.charge = np.array([ff.nonbonded[t][0] for t in atom_type])
    w.sigma = np.array([ff.nonbonded[t][1] for t in atom_type])
    w.epsilon = np.array([ff.nonbonded[t][2] for t in atom_type])
    w.method, w.cutoff, w.dispersion, w.cm_remover = method, cutoff, True, True
    if ff.gbsa:
        # (charge, radius, scale) per atom + the NonbondedForce's reaction field switched off (GBSAOBCGenerator.postprocessSystem)
        w.gbsa = tuple(np.array([ff.gbsa[t][k] for t in atom_type]) for k in range(3))
        w.reaction_field_dielectric = 1.0
    w.exception_bonds = np.array(bonds, dtype=np.int64)
    w.coulomb14, w.lj14 = ff.coulomb14, ff.lj14

    def bond_params(a, b):
        p = ff.bonds.get((cls[a], cls[b])) or ff.bonds.get((cls[b], cls[a]))
        if p is None:
            raise KeyError("no bond parameters for %s-%s" % (cls[a], cls[b]))
        return p
    # ---- bonds / constraints
    cons_pairs, cons_len, b_atoms, b_len, b_k = [], [], [], [], []
    constrained = set()
    for a, b in bonds:
        length, k = bond_params(a, b)
        rigid = rigid_water and is_water[a]
        if rigid or (constraints_hbonds and (el[a] == "H" or el[b] == "H")):
            cons_pairs.append((a, b)); cons_len.append(length); constrained.add((a, b))
        else:
            b_atoms.append((a, b)); b_len.append(length); b_k.append(k)
    # ---- angles
    angles = set()
    for j in range(n):
        for i, k in itertools.combinations(sorted(bonded[j]), 2):
            angles.add((i, j, k))
    a_atoms, a_theta, a_k = [], [], []
    for i, j, k in sorted(angles):
        p = ff.angles.get((cls[i], cls[j], cls[k])) or ff.angles.get((cls[k], cls[j], cls[i]))
        if p is None:
            raise KeyError("no angle parameters for %s-%s-%s" % (cls[i], cls[j], cls[k]))
        if rigid_water and is_water[j]:
            # H-H distance of the rigid water from the two bond lengths and the angle (forcefield.py:1325-1338)
            l1, l2 = bond_params(i, j)[0], bond_params(j, k)[0]
            cons_pairs.append((i, k)); cons_len.append(float(np.sqrt(l1 * l1 + l2 * l2 - 2 * l1 * l2 * np.cos(p[0]))))
            continue
        a_atoms.append((i, j, k)); a_theta.append(p[0]); a_k.append(p[1])
    # ---- proper torsions
    propers = set()
    for i, j, k in sorted(angles):
        for x in bonded[i]:
            if x not in (i, j, k):
                propers.add((x, i, j, k) if x < k else (k, j, i, x))
        for x in bonded[k]:
            if x not in (i, j, k):
                propers.add((i, j, k, x) if x > i else (x, k, j, i))
    t_atoms, t_n, t_phase, t_k = [], [], [], []

    def fits(c, pattern):
        return pattern == "" or pattern == c
    cache_p = {}
    for tor in sorted(propers):
        c = tuple(cls[x] for x in tor)
        key = min(c, c[::-1])
        if key not in cache_p:
            match = None
            for classes, terms in ff.propers:
                fwd = all(fits(c[i], classes[i]) for i in range(4))
                rev = all(fits(c[3 - i], classes[i]) for i in range(4))
                if fwd or rev:
                    wild = "" in classes
                    if match is None or not wild:
                        match = terms
                    if not wild:
                        break
            cache_p[key] = match
        if cache_p[key]:
            for per, phase, k in cache_p[key]:
                if k != 0:
                    t_atoms.append(tor); t_n.append(per); t_phase.append(phase); t_k.append(k)
    # ---- improper torsions ("default" ordering of forcefield.py:_matchImproper)
    for centre in range(n):
        if len(bonded[centre]) < 3:
            continue
        for subset in itertools.combinations(bonded[centre], 3):
            tor = (centre,) + subset
            match = None
            for classes, terms in ff.impropers:
                wild = "" in classes
                if match is not None and wild:
                    continue
                if not fits(cls[centre], classes[0]):
                    continue
                for perm in itertools.permutations((1, 2, 3)):
                    if all(fits(cls[tor[perm[i]]], classes[i + 1]) for i in range(3)):
                        a1, a2 = tor[perm[0]], tor[perm[1]]
                        e1, e2 = el[a1], el[a2]
                        if e1 == e2 and a1 > a2:
                            a1, a2 = a2, a1
                        elif e1 != "C" and (e2 == "C" or ELEMENT_MASS[e1] < ELEMENT_MASS[e2]):
                            a1, a2 = a2, a1
                        match = ((a1, a2, centre, tor[perm[2]]), terms)
                        break
            if match is not None:
                for per, phase, k in match[1]:
                    if k != 0:
                        t_atoms.append(match[0]); t_n.append(per); t_phase.append(phase); t_k.append(k)
    w.bonds = (np.array(b_atoms, dtype=np.int64).reshape(-1, 2), np.array(b_len), np.array(b_k))
    w.angles = (np.array(a_atoms, dtype=np.int64).reshape(-1, 3), np.array(a_theta), np.array(a_k))
    w.torsions = (np.array(t_atoms, dtype=np.int64).reshape(-1, 4), np.array(t_n, dtype=np.int32), np.array(t_phase), np.array(t_k))
    w.constraints = (np.array(cons_pairs, dtype=np.int64).reshape(-1, 2), np.array(cons_len))
    w.velocities = None
    return w


def dhfr(data_dir=None, pdb_path="/root/reference/examples/5dfr_solv-cube_equil.pdb"):
    """The `pme` test of examples/benchmark.py:87-90,133-138: DHFR in TIP3P water, amber99sb, PME 0.9 nm, HBonds, rigid water."""
    ff = ForceField("amber99sb.xml", "tip3p.xml", data_dir=data_dir)
    return create_workload(pdb_path, ff, "dhfr-23558 (5dfr_solv-cube_equil.pdb, amber99sb + tip3p)")


def lysozyme_implicit(data_dir=None, pdb_path="/root/reference/wrappers/python/tests/systems/lysozyme-implicit.pdb"):
    """The System of wrappers/python/tests/TestForceField.py:285-301 (test_Forces): T4 lysozyme, amber99sb + amber99_obc (GBSA-OBC),
    createSystem's defaults -- NoCutoff, no constraints, CMMotionRemover.  The reference keeps golden forces of it
    (tests/systems/lysozyme-implicit-forces.xml): the pin of this reader against the app layer."""
    ff = ForceField("amber99sb.xml", "amber99_obc.xml", data_dir=data_dir)
    return create_workload(pdb_path, ff, "lysozyme-implicit (amber99sb + amber99_obc)", cutoff=1.0, constraints_hbonds=False, rigid_water=True, method=0)
