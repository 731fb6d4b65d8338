"""System XML for the reference's XmlSerializer from a testsystems.Workload ("the real DHFR System without the app
layer", delivered as a file `XmlSerializer::deserialize<System>` loads -- serialization/include/openmm/serialization/XmlSerializer.h:74-76).

Pure Python, no OpenMM import: the writer follows the proxies' own formats
  serialization/src/SystemProxy.cpp:46-105           <System version=1> PeriodicBoxVectors / Particles / Constraints / Forces
  serialization/src/HarmonicBondForceProxy.cpp:44-57, HarmonicAngleForceProxy.cpp:44-57, PeriodicTorsionForceProxy.cpp:44-57
  serialization/src/NonbondedForceProxy.cpp:43-103   version 4, every particle and EVERY exception written out
  serialization/src/GBSAOBCForceProxy.cpp:44-60, CMMotionRemoverProxy.cpp:44-50
and restates NonbondedForce::createExceptionsFromBonds (openmmapi/src/NonbondedForce.cpp:207-257: 1-2 and 1-3 pairs excluded, 1-4 pairs
scaled) because a serialized force carries its exceptions explicitly.  The force order is the one testsystems.Workload.build uses.
tests/test_forcefield.py round-trips the file through the real deserializer and compares forces on the Reference platform.
"""
import numpy as np


def _f(x):
    """Shortest round-trip decimal in the style of the reference's writer (serialization/src/g_fmt.cpp: no leading zero before the point,
    no trailing ".0", two-digit exponents) -- so that a file written here and one written by XmlSerializer::serialize of the same
    System are the same text."""
    r = repr(float(x))
    if r in ("inf", "-inf", "nan"):
        return r
    mantissa, _, exponent = r.partition("e")
    if mantissa.endswith(".0"):
        mantissa = mantissa[:-2]
    if mantissa.startswith("0."):
        mantissa = mantissa[1:]
    elif mantissa.startswith("-0."):
        mantissa = "-" + mantissa[2:]
    return mantissa + ("e" + exponent if exponent else "")


def exceptions_from_bonds(bonds, charge, sigma, epsilon, coulomb14, lj14):
    """-> list of (p1, p2, chargeProd, sigma, epsilon) in the order NonbondedForce::createExceptionsFromBonds adds them."""
    n = len(charge)
    bonded = [set() for _ in range(n)]
    for a, b in np.asarray(bonds, dtype=np.int64).reshape(-1, 2):
        bonded[a].add(int(b)); bonded[b].add(int(a))
    out = []
    for i in range(n):
        if not bonded[i]:
            continue
        within2 = set()
        for j in bonded[i]:
            within2.add(j)
            within2.update(bonded[j])
        within3 = set(within2)
        for j in within2:
            within3.update(bonded[j])
        within2.discard(i); within3.discard(i)
        for j in sorted(within3):
            if j < i:
                if j in within2:
                    out.append((j, i, 0.0, 1.0, 0.0))
                else:
                    out.append((j, i, coulomb14 * charge[j] * charge[i], 0.5 * (sigma[j] + sigma[i]), lj14 * float(np.sqrt(epsilon[j] * epsilon[i]))))
    return out


def workload_to_xml(w, openmm_version="8.0"):
    """The System Workload.build() would create, as XmlSerializer text."""
    box = w.box if w.box is not None else np.diag([2.0, 2.0, 2.0])          # System's default box
    L = []
    add = L.append
    add('<?xml version="1.0" ?>')
    add('<System openmmVersion="%s" type="System" version="1">' % openmm_version)
    add('\t<PeriodicBoxVectors>')
    for name, v in zip("ABC", np.asarray(box, dtype=np.float64)):
        add('\t\t<%s x="%s" y="%s" z="%s"/>' % (name, _f(v[0]), _f(v[1]), _f(v[2])))
    add('\t</PeriodicBoxVectors>')
    add('\t<Particles>')
    for m in w.masses:
        add('\t\t<Particle mass="%s"/>' % _f(m))
    add('\t</Particles>')
    add('\t<Constraints>')
    if w.constraints is not None:
        for (a, b), d in zip(w.constraints[0], w.constraints[1]):
            add('\t\t<Constraint d="%s" p1="%d" p2="%d"/>' % (_f(d), a, b))
    add('\t</Constraints>')
    add('\t<Forces>')
    # ---- NonbondedForce
    periodic_method = w.method
    alpha, nx, ny, nz = w.pme_params if w.pme_params is not None else (0.0, 0, 0, 0)
    rf = getattr(w, "reaction_field_dielectric", None)
    add('\t\t<Force alpha="%s" cutoff="%s" dispersionCorrection="%d" ewaldTolerance="%s" exceptionsUsePeriodic="0" forceGroup="0" includeDirectSpace="1" '
        'ljAlpha="0" ljnx="0" ljny="0" ljnz="0" method="%d" name="NonbondedForce" nx="%d" ny="%d" nz="%d" recipForceGroup="-1" rfDielectric="%s" '
        'switchingDistance="-1" type="NonbondedForce" useSwitchingFunction="0" version="4">'
        % (_f(alpha), _f(w.cutoff), 1 if w.dispersion else 0, _f(w.ewald_tol), periodic_method, nx, ny, nz, _f(78.3 if rf is None else rf)))
    add('\t\t\t<GlobalParameters/>')
    add('\t\t\t<ParticleOffsets/>')
    add('\t\t\t<ExceptionOffsets/>')
    add('\t\t\t<Particles>')
    for q, s, e in zip(w.charge, w.sigma, w.epsilon):
        add('\t\t\t\t<Particle eps="%s" q="%s" sig="%s"/>' % (_f(e), _f(q), _f(s)))
    add('\t\t\t</Particles>')
    add('\t\t\t<Exceptions>')
    exceptions = []
    if w.exception_bonds is not None and len(w.exception_bonds):
        exceptions += exceptions_from_bonds(w.exception_bonds, w.charge, w.sigma, w.epsilon, getattr(w, "coulomb14", 1.0 / 1.2), getattr(w, "lj14", 0.5))
    if w.exceptions is not None and len(w.exceptions[0]):
        exceptions += [(int(p[0]), int(p[1]), q, s, e) for p, q, s, e in zip(*w.exceptions)]
    for a, b, q, s, e in exceptions:
        add('\t\t\t\t<Exception eps="%s" p1="%d" p2="%d" q="%s" sig="%s"/>' % (_f(e), a, b, _f(q), _f(s)))
    add('\t\t\t</Exceptions>')
    add('\t\t</Force>')
    if getattr(w, "gbsa", None) is not None:
        method = 0 if w.method == 0 else (1 if w.method == 1 else 2)
        add('\t\t<Force cutoff="%s" forceGroup="0" method="%d" name="GBSAOBCForce" soluteDielectric="1" solventDielectric="78.3" surfaceAreaEnergy="2.25936" type="GBSAOBCForce" version="2">'
            % (_f(w.cutoff), method))
        add('\t\t\t<Particles>')
        for q, r, sc in zip(*w.gbsa):
            add('\t\t\t\t<Particle q="%s" r="%s" scale="%s"/>' % (_f(q), _f(r), _f(sc)))
        add('\t\t\t</Particles>')
        add('\t\t</Force>')
    if w.bonds is not None and len(w.bonds[0]):
        add('\t\t<Force forceGroup="0" name="HarmonicBondForce" type="HarmonicBondForce" usesPeriodic="0" version="2">')
        add('\t\t\t<Bonds>')
        for (a, b), d, k in zip(*w.bonds):
            add('\t\t\t\t<Bond d="%s" k="%s" p1="%d" p2="%d"/>' % (_f(d), _f(k), a, b))
        add('\t\t\t</Bonds>')
        add('\t\t</Force>')
    if w.angles is not None and len(w.angles[0]):
        add('\t\t<Force forceGroup="0" name="HarmonicAngleForce" type="HarmonicAngleForce" usesPeriodic="0" version="2">')
        add('\t\t\t<Angles>')
        for (a, b, c), t, k in zip(*w.angles):
            add('\t\t\t\t<Angle a="%s" k="%s" p1="%d" p2="%d" p3="%d"/>' % (_f(t), _f(k), a, b, c))
        add('\t\t\t</Angles>')
        add('\t\t</Force>')
    if w.torsions is not None and len(w.torsions[0]):
        add('\t\t<Force forceGroup="0" name="PeriodicTorsionForce" type="PeriodicTorsionForce" usesPeriodic="0" version="2">')
        add('\t\t\t<Torsions>')
        for (a, b, c, d), n, ph, k in zip(*w.torsions):
            add('\t\t\t\t<Torsion k="%s" p1="%d" p2="%d" p3="%d" p4="%d" periodicity="%d" phase="%s"/>' % (_f(k), a, b, c, d, n, _f(ph)))
        add('\t\t\t</Torsions>')
        add('\t\t</Force>')
    if w.cm_remover:
        add('\t\t<Force forceGroup="0" frequency="1" name="CMMotionRemover" type="CMMotionRemover" version="1"/>')
    add('\t</Forces>')
    add('</System>')
    return "\n".join(L) + "\n"
