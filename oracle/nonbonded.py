"""oracle/nonbonded.py -- TEST INFRASTRUCTURE ONLY (never imported by the product path).

A plain numpy (float64) restatement of the Reference platform's NonbondedForce arithmetic, used as
the checker for the HIP kernels.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg may import this module.

Parity pinning: tests/test_oracle_vs_reference.py checks every function here against the real
reference (oracle/_ref/libOpenMM.so, Reference platform) on seeded inputs and against the golden
values of the reference's own tests (tests/TestNonbondedForce.h, tests/TestEwald.h).

All pair sums are O(N^2) dense numpy: intended for N up to a few thousand atoms.

Reference sources restated (paths relative to the OpenMM tree):
  platforms/reference/src/ReferenceKernels.cpp:1077-1121      parameter combination (sigma/2, 2 sqrt(eps))
  platforms/reference/src/SimTKReference/ReferenceForce.cpp:90-101   minimum image (triclinic-aware)
  platforms/reference/src/SimTKReference/ReferenceLJCoulombIxn.cpp:543-639  cutoff / no-cutoff pair ixn
  platforms/reference/src/SimTKReference/ReferenceLJCoulombIxn.cpp:190-233  Ewald self energy
  platforms/reference/src/SimTKReference/ReferenceLJCoulombIxn.cpp:272-367  Ewald reciprocal k-sum
  platforms/reference/src/SimTKReference/ReferenceLJCoulombIxn.cpp:379-457  Ewald direct sum
  platforms/reference/src/SimTKReference/ReferenceLJCoulombIxn.cpp:462-523  exclusion correction
  platforms/reference/src/SimTKReference/ReferenceLJCoulomb14.cpp           1-4 exceptions
"""
import numpy as np
from scipy.special import erf, erfc

ONE_4PI_EPS0 = 138.93545764438198  # 1/(4 pi EPSILON0), platforms/reference/include/SimTKOpenMMRealType.h:74-89 (CODATA 2018)

NoCutoff, CutoffNonPeriodic, CutoffPeriodic, Ewald, PME, LJPME = range(6)


def min_image(d, box):
    """d[...,3] displacement(s); box = 3x3 reduced box vectors (rows a,b,c).  ReferenceForce.cpp:90-101."""
    d = np.array(d, dtype=np.float64, copy=True)
    box = np.asarray(box, dtype=np.float64)
    for k in (2, 1, 0):
        s = np.floor(d[..., k] / box[k, k] + 0.5)
        d -= s[..., None] * box[k]
    return d


def switch_function(r, rs, rc):
    """ReferenceLJCoulombIxn.cpp:388-392."""
    t = np.clip((r - rs) / (rc - rs), 0.0, None)
    sw = 1 + t ** 3 * (-10 + t * (15 - t * 6))
    dsw = t * t * (-30 + t * (60 - t * 30)) / (rc - rs)
    sw = np.where(r > rs, sw, 1.0)
    dsw = np.where(r > rs, dsw, 0.0)
    return sw, dsw


def reaction_field_constants(cutoff, dielectric):
    """ReferenceLJCoulombIxn.cpp:74-80 (setUseCutoff)."""
    krf = (1.0 / cutoff ** 3) * (dielectric - 1.0) / (2.0 * dielectric + 1.0)
    crf = (1.0 / cutoff) * (3.0 * dielectric) / (2.0 * dielectric + 1.0)
    return krf, crf


def direct_space(pos, charge, sigma, epsilon, method, cutoff=None, box=None, exclusions=(), alpha=0.0,
                 rf_dielectric=78.3, switch_distance=None):
    """Direct-space forces [N,3] and energy of all non-excluded pairs.

    exclusions: iterable of (i, j) pairs that are skipped entirely.
    Returns (forces, energy).  Pair formulae: ReferenceLJCoulombIxn.cpp:379-457 (Ewald/PME) and :586-639.
    """
    pos = np.asarray(pos, dtype=np.float64)
    n = len(pos)
    q = np.asarray(charge, dtype=np.float64)
    hs = 0.5 * np.asarray(sigma, dtype=np.float64)
    se = 2.0 * np.sqrt(np.asarray(epsilon, dtype=np.float64))
    d = pos[None, :, :] - pos[:, None, :]        # d[i,j] = pos[j]-pos[i]
    periodic = method in (CutoffPeriodic, Ewald, PME)
    if periodic:
        d = min_image(d, box)
    r2 = np.einsum("ijk,ijk->ij", d, d)
    iu = np.triu(np.ones((n, n), dtype=bool), 1)
    mask = iu.copy()
    for (i, j) in exclusions:
        mask[min(i, j), max(i, j)] = False
    if method != NoCutoff:
        mask &= r2 < cutoff * cutoff
    r2s = np.where(mask, r2, 1.0)
    r = np.sqrt(r2s)
    inv_r = 1.0 / r
    sig = hs[:, None] + hs[None, :]
    eps = se[:, None] * se[None, :]
    sig6 = (sig * inv_r) ** 6
    lj_f = eps * (12.0 * sig6 - 6.0) * sig6
    lj_e = eps * (sig6 - 1.0) * sig6
    if switch_distance is not None and method != NoCutoff:
        sw, dsw = switch_function(r, switch_distance, cutoff)
        lj_f = sw * lj_f - lj_e * dsw * r
        lj_e = lj_e * sw
    qq = ONE_4PI_EPS0 * q[:, None] * q[None, :]
    if method in (Ewald, PME):
        ar = alpha * r
        c_f = qq * inv_r * (erfc(ar) + 2.0 * ar * np.exp(-ar * ar) / np.sqrt(np.pi))
        c_e = qq * inv_r * erfc(ar)
    elif method == NoCutoff:
        c_f = qq * inv_r
        c_e = qq * inv_r
    else:
        krf, crf = reaction_field_constants(cutoff, rf_dielectric)
        c_f = qq * (inv_r - 2.0 * krf * r2s)
        c_e = qq * (inv_r + krf * r2s - crf)
    dedr = np.where(mask, (lj_f + c_f) * inv_r * inv_r, 0.0)
    fpair = dedr[:, :, None] * d                  # force on j from i; i gets the negative
    forces = fpair.sum(axis=0) - fpair.sum(axis=1)
    energy = float(np.where(mask, lj_e + c_e, 0.0).sum())
    return forces, energy


def exceptions_14(pos, exceptions, box=None, periodic=False):
    """1-4 exceptions: plain LJ + Coulomb, no cutoff (ReferenceLJCoulomb14.cpp).

    exceptions: iterable of (i, j, chargeProd, sigma, epsilon).
    """
    pos = np.asarray(pos, dtype=np.float64)
    forces = np.zeros_like(pos)
    energy = 0.0
    for (i, j, qq, sig, eps) in exceptions:
        d = pos[j] - pos[i]
        if periodic:
            d = min_image(d, box)
        r2 = d @ d
        inv_r = 1.0 / np.sqrt(r2)
        s6 = (sig * inv_r) ** 6
        e4 = 4.0 * eps
        dedr = (e4 * (12.0 * s6 - 6.0) * s6 + ONE_4PI_EPS0 * qq * inv_r) * inv_r * inv_r
        energy += e4 * (s6 - 1.0) * s6 + ONE_4PI_EPS0 * qq * inv_r
        forces[j] += dedr * d
        forces[i] -= dedr * d
    return forces, energy


def ewald_self_energy(charge, alpha):
    """ReferenceLJCoulombIxn.cpp:220-233."""
    q = np.asarray(charge, dtype=np.float64)
    return float(-ONE_4PI_EPS0 * alpha / np.sqrt(np.pi) * np.sum(q * q))


def ewald_exclusion_correction(pos, charge, exclusions, alpha, box=None, periodic=False):
    """Subtract erf(alpha r)/r for every excluded pair (ReferenceLJCoulombIxn.cpp:462-523)."""
    pos = np.asarray(pos, dtype=np.float64)
    q = np.asarray(charge, dtype=np.float64)
    forces = np.zeros_like(pos)
    energy = 0.0
    for (i, j) in exclusions:
        d = pos[j] - pos[i]
        if periodic:
            d = min_image(d, box)
        r = np.sqrt(d @ d)
        ar = alpha * r
        qq = ONE_4PI_EPS0 * q[i] * q[j]
        if erf(ar) > 1e-6:
            inv_r = 1.0 / r
            dedr = qq * inv_r ** 3 * (erf(ar) - 2.0 * ar * np.exp(-ar * ar) / np.sqrt(np.pi))
            forces[j] -= dedr * d
            forces[i] += dedr * d
            energy -= qq * inv_r * erf(ar)
        else:
            energy -= alpha * 2.0 / np.sqrt(np.pi) * qq
    return forces, energy


def ewald_reciprocal(pos, charge, box, alpha, kmax):
    """Classic Ewald k-sum for rectangular boxes (ReferenceLJCoulombIxn.cpp:272-367).

    kmax = (kx, ky, kz) as returned by NonbondedForceImpl::calcEwaldParameters.
    """
    pos = np.asarray(pos, dtype=np.float64)
    q = np.asarray(charge, dtype=np.float64)
    L = np.array([box[0][0], box[1][1], box[2][2]], dtype=np.float64)
    recip = 2.0 * np.pi / L
    volume = L[0] * L[1] * L[2]
    coeff = ONE_4PI_EPS0 * 4.0 * np.pi / volume
    factor = -1.0 / (4.0 * alpha * alpha)
    forces = np.zeros_like(pos)
    energy = 0.0
    nx, ny, nz = kmax
    lowry, lowrz = 0, 1
    for rx in range(nx):
        for ry in range(lowry, ny):
            for rz in range(lowrz, nz):
                k = np.array([rx, ry, rz], dtype=np.float64) * recip
                phase = pos @ k
                c = q * np.cos(phase)
                s = q * np.sin(phase)
                cs, ss = c.sum(), s.sum()
                k2 = k @ k
                ak = np.exp(k2 * factor) / k2
                f = ak * (cs * s - ss * c)
                forces += 2.0 * coeff * f[:, None] * k[None, :]
                energy += coeff * ak * (cs * cs + ss * ss)
                lowrz = 1 - nz
            lowry = 1 - ny
    return forces, float(energy)
