"""TEST INFRASTRUCTURE -- numpy restatement of the AMOEBA valence energies that kernels/valence.hip evaluates, for kernel-level parity
tests through the C ABI (ommhip_valence_forces).  Energies only, written directly from the reference's expressions; forces are central
differences of these energies in double precision, so nothing here shares a derivation (or a gradient formula) with the kernels.

  poly_bond          wrappers/python/openmm/app/forcefield.py:3368   k (d^2 + c0 d^3 + c1 d^4), d = r - r0
  poly_angle         :3502   k (d^2 + c0 d^3 + c1 d^4 + c2 d^5 + c3 d^6), d = c4 theta - theta0
  inplane_angle      :3565   the same polynomial of the angle 1-P-3, P = atom 2 projected onto the plane through 1, 3, 4
  out_of_plane_bend  :3730   k (t^2 + ...), t = c4 x the angle at atom 4 between atom 2 and P
  stretch_bend       :4428   (k1 (r12 - r12_0) + k2 (r23 - r23_0)) c0 (theta - theta0)
  pi_torsion         :4039   2 k sin^2(phi), phi = pointdihedral(3 + c1, 3, 4, 4 + c2), c1 = (1 - 4) x (2 - 4), c2 = (5 - 3) x (6 - 3)
  torsion_torsion    plugins/amoeba/platforms/reference/src/SimTKReference/AmoebaReferenceTorsionTorsionForce.cpp:283-530: a function of the
                     two signed dihedrals (degrees), both negated when (marker - c) . ((b - c) x (d - c)) < 0; here the map is ANY callable
                     f(phi, psi) -- the tests tabulate a bicubic polynomial, which the bicubic patch of the kernel must reproduce exactly

Parity pinned: these functions give the Reference platform's energies of the amoeba2009 DHFR System term by term (tests/test_forcefield_amoeba.py
compares the sum of each kind with the Reference platform through the harness).
"""
import numpy as np


def _angle(u, w):
    c = (u * w).sum(-1) / np.sqrt((u * u).sum(-1) * (w * w).sum(-1))
    return np.arccos(np.clip(c, -1.0, 1.0))


def _poly(c, x):
    return x ** 2 + c[0] * x ** 3 + c[1] * x ** 4 + c[2] * x ** 5 + c[3] * x ** 6


def poly_bond(pos, atoms, params, c):
    d = np.linalg.norm(pos[atoms[:, 1]] - pos[atoms[:, 0]], axis=1) - params[:, 0]
    return params[:, 1] * (d ** 2 + c[0] * d ** 3 + c[1] * d ** 4)


def poly_angle(pos, atoms, params, c):
    theta = _angle(pos[atoms[:, 0]] - pos[atoms[:, 1]], pos[atoms[:, 2]] - pos[atoms[:, 1]])
    return params[:, 1] * _poly(c, c[4] * theta - params[:, 0])


def _projection(pos, atoms):
    x1, x2, x3, x4 = (pos[atoms[:, k]] for k in range(4))
    p = np.cross(x1 - x4, x3 - x4)
    n = p / np.linalg.norm(p, axis=1)[:, None]
    return x2 - n * ((n * (x2 - x3)).sum(1))[:, None]


def inplane_angle(pos, atoms, params, c):
    proj = _projection(pos, atoms)
    theta = _angle(pos[atoms[:, 0]] - proj, pos[atoms[:, 2]] - proj)
    return params[:, 1] * _poly(c, c[4] * theta - params[:, 0])


def out_of_plane_bend(pos, atoms, params, c):
    proj = _projection(pos, atoms)
    theta = _angle(pos[atoms[:, 1]] - pos[atoms[:, 3]], proj - pos[atoms[:, 3]])
    return params[:, 0] * _poly(c, c[4] * theta)


def stretch_bend(pos, atoms, params, c):
    u, w = pos[atoms[:, 0]] - pos[atoms[:, 1]], pos[atoms[:, 2]] - pos[atoms[:, 1]]
    return (params[:, 3] * (np.linalg.norm(u, axis=1) - params[:, 0]) + params[:, 4] * (np.linalg.norm(w, axis=1) - params[:, 1])) * c[0] * (_angle(u, w) - params[:, 2])


def _dihedral(p1, p2, p3, p4):
    """Lepton's pointdihedral / the signed dihedral of four points (radians), positive when (p2 - p1) . ((p3 - p2) x (p4 - p3)) >= 0"""
    ba, cb, dc = p2 - p1, p3 - p2, p4 - p3
    t, u = np.cross(ba, cb), np.cross(cb, dc)
    phi = _angle(t, u)
    return np.where((ba * u).sum(-1) < 0, -phi, phi)


def pi_torsion(pos, atoms, params, c):
    x = [pos[atoms[:, k]] for k in range(6)]
    c1, c2 = np.cross(x[0] - x[3], x[1] - x[3]), np.cross(x[4] - x[2], x[5] - x[2])
    phi = _dihedral(x[2] + c1, x[2], x[3], x[3] + c2)
    return 2 * params[:, 0] * np.sin(phi) ** 2


def torsion_torsion(pos, atoms, surface):
    """surface(phi, psi) with the angles in degrees"""
    a, b, cc, d, e = (pos[atoms[:, k]] for k in range(5))
    phi, psi = np.degrees(_dihedral(a, b, cc, d)), np.degrees(_dihedral(b, cc, d, e))
    marker = atoms[:, 5]
    m = pos[np.where(marker >= 0, marker, 0)]
    volume = ((m - cc) * np.cross(b - cc, d - cc)).sum(1)
    sign = np.where((marker >= 0) & (volume < 0), -1.0, 1.0)
    return surface(sign * phi, sign * psi)


def forces(energy, pos, h=1e-5):
    """-grad of sum(energy(pos)) by central differences; energy: positions -> per-term energies"""
    f = np.zeros_like(pos)
    for i in range(len(pos)):
        for k in range(3):
            p = pos.copy(); p[i, k] += h
            m = pos.copy(); m[i, k] -= h
            f[i, k] = -(energy(p).sum() - energy(m).sum()) / (2 * h)
    return f
