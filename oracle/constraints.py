"""oracle/constraints.py -- TEST INFRASTRUCTURE ONLY (never imported by the product path).

A numpy (float64) restatement of the Reference platform's constraint algorithms and of its Verlet step, the
checker for ommhip_settle / ommhip_shake / ommhip_ccma_iterations / ommhip_integrate_* (include/openmm_hip_kernels.h).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Parity pinning: tests/test_oracle_vs_reference.py drives the real Reference platform (build/openmm/lib/libOpenMM.so)
through Context.applyConstraints / applyVelocityConstraints and through force-free VerletIntegrator steps (which hand
distinct "before" and "trial" positions to the constraint algorithms) and compares at 1e-12 ... 1e-9.

Reference sources restated (paths relative to the OpenMM tree):
  platforms/reference/src/SimTKReference/ReferenceSETTLEAlgorithm.cpp:54-195    SETTLE, positions (Miyamoto & Kollman 1992)
  platforms/reference/src/SimTKReference/ReferenceSETTLEAlgorithm.cpp:197-244   SETTLE, velocities (general masses)
  platforms/reference/src/SimTKReference/ReferenceCCMAAlgorithm.cpp:42-196      CCMA coupling matrix and its thresholded inverse
  platforms/reference/src/SimTKReference/ReferenceCCMAAlgorithm.cpp:224-311     CCMA iteration (positions and velocities)
  platforms/reference/src/SimTKReference/ReferenceConstraints.cpp:150-206       SETTLE clusters first, CCMA for the rest, order of application
  platforms/reference/src/SimTKReference/ReferenceVerletDynamics.cpp:76-119     Verlet step around the constraints
  platforms/common/src/kernels/integrationUtilities.cc (applyShakeToHydrogens)  SHAKE on a centre atom with up to three satellites
      (Ryckaert, Ciccotti & Berendsen 1977): the Reference platform has no SHAKE -- it hands such clusters to CCMA -- so the SHAKE
      restatement is pinned to the Reference platform's CCMA result at a tolerance where both have converged to the same solution.

The SETTLE velocity stage is restated as what it solves -- three impulses along the edges of the triangle that cancel the
relative velocity along each edge -- through a 3 x 3 linear solve per water, not through the reference's closed form.
"""
import numpy as np


def _unit(v):
    return v / np.linalg.norm(v, axis=-1, keepdims=True)


def settle_positions(pos, trial, mass, clusters, d_leg, d_base):
    """SETTLE: `trial` positions of the waters `clusters` (int[n, 3]: apex, leg atom, leg atom) are reset so that apex-leg = d_leg[n]
    and leg-leg = d_base[n]; pos = the constrained positions before the step.  Returns the corrected copy of `trial`.
    ReferenceSETTLEAlgorithm.cpp:54-195, written with whole arrays of 3-vectors."""
    pos, out = np.asarray(pos, np.float64), np.array(trial, np.float64, copy=True)
    cl = np.asarray(clusters, np.int64).reshape(-1, 3)
    if len(cl) == 0:
        return out
    d_leg, d_base = np.broadcast_to(np.asarray(d_leg, np.float64), (len(cl),)), np.broadcast_to(np.asarray(d_base, np.float64), (len(cl),))
    m = np.asarray(mass, np.float64)[cl]                                   # [n, 3]
    p = pos[cl]                                                            # [n, 3 atoms, 3]
    q = out[cl]
    b0, c0 = p[:, 1] - p[:, 0], p[:, 2] - p[:, 0]
    # trial positions relative to the old apex, then to their centre of mass
    t = q - p[:, :1]
    inv_m = 1.0 / m.sum(1)
    com = (t * m[:, :, None]).sum(1) * inv_m[:, None]
    a1, b1, c1 = (t[:, k] - com for k in range(3))
    # frame: z normal to the old triangle, x = a1 x z, y = z x x
    ez = np.cross(b0, c0)
    ex = np.cross(a1, ez)
    ey = np.cross(ez, ex)
    ex, ey, ez = _unit(ex), _unit(ey), _unit(ez)
    dot = lambda u, w: (u * w).sum(-1)
    xb0, yb0, xc0, yc0 = dot(ex, b0), dot(ey, b0), dot(ex, c0), dot(ey, c0)
    za1 = dot(ez, a1)
    xb1, yb1, zb1 = dot(ex, b1), dot(ey, b1), dot(ez, b1)
    xc1, yc1, zc1 = dot(ex, c1), dot(ey, c1), dot(ez, c1)
    # the canonical triangle (apex on +y at ra, legs at -rb, +-rc)
    rc = 0.5 * d_base
    rb = np.sqrt(d_leg * d_leg - rc * rc)
    ra = rb * (m[:, 1] + m[:, 2]) * inv_m
    rb = rb - ra
    sinphi = za1 / ra
    cosphi = np.sqrt(1 - sinphi ** 2)
    sinpsi = (zb1 - zc1) / (2 * rc * cosphi)
    cospsi = np.sqrt(1 - sinpsi ** 2)
    ya2 = ra * cosphi
    xb2 = -rc * cospsi
    yb2 = -rb * cosphi - rc * sinpsi * sinphi
    yc2 = -rb * cosphi + rc * sinpsi * sinphi
    hh2 = 4 * xb2 ** 2 + (yb2 - yc2) ** 2 + (zb1 - zc1) ** 2
    xb2 = xb2 - 0.5 * (2 * xb2 + np.sqrt(4 * xb2 ** 2 - hh2 + d_base ** 2))
    # rotation about z that takes it onto the trial orientation
    alpha = xb2 * (xb0 - xc0) + yb0 * yb2 + yc0 * yc2
    beta = xb2 * (yc0 - yb0) + xb0 * yb2 + xc0 * yc2
    gamma = xb0 * yb1 - xb1 * yb0 + xc0 * yc1 - xc1 * yc0
    ab2 = alpha ** 2 + beta ** 2
    sint = (alpha * gamma - beta * np.sqrt(ab2 - gamma ** 2)) / ab2
    cost = np.sqrt(1 - sint ** 2)
    col = lambda s: s[:, None]
    a3 = ex * col(-ya2 * sint) + ey * col(ya2 * cost) + ez * col(za1)
    b3 = ex * col(xb2 * cost - yb2 * sint) + ey * col(xb2 * sint + yb2 * cost) + ez * col(zb1)
    c3 = ex * col(-xb2 * cost - yc2 * sint) + ey * col(-xb2 * sint + yc2 * cost) + ez * col(zc1)
    out[cl[:, 0]] = p[:, 0] + com + a3
    out[cl[:, 1]] = p[:, 0] + com + b3
    out[cl[:, 2]] = p[:, 0] + com + c3
    return out


def settle_velocities(pos, vel, mass, clusters):
    """Velocity stage of SETTLE (ReferenceSETTLEAlgorithm.cpp:197-244): impulses t_AB, t_BC, t_CA along the three edges such that the
    relative velocity along every edge vanishes.  Solved here as the 3 x 3 linear system it is."""
    pos, out = np.asarray(pos, np.float64), np.array(vel, np.float64, copy=True)
    cl = np.asarray(clusters, np.int64).reshape(-1, 3)
    if len(cl) == 0:
        return out
    inv = 1.0 / np.asarray(mass, np.float64)[cl]
    p, v = pos[cl], out[cl]
    e = np.stack([_unit(p[:, 1] - p[:, 0]), _unit(p[:, 2] - p[:, 1]), _unit(p[:, 0] - p[:, 2])], 1)      # AB, BC, CA
    # atom k receives  sum_edges sign[k, edge] * t_edge * e_edge / m_k ;  A: +t_AB e_AB - t_CA e_CA, B: +t_BC e_BC - t_AB e_AB, C: +t_CA e_CA - t_BC e_BC
    sign = np.array([[1.0, 0.0, -1.0], [-1.0, 1.0, 0.0], [0.0, -1.0, 1.0]])
    head, tail = (1, 2, 0), (0, 1, 2)                                       # edge k runs tail -> head
    A = np.zeros((len(cl), 3, 3))
    rhs = np.zeros((len(cl), 3))
    for k in range(3):
        rhs[:, k] = -((v[:, head[k]] - v[:, tail[k]]) * e[:, k]).sum(-1)
        for j in range(3):
            A[:, k, j] = (sign[head[k], j] * inv[:, head[k]] - sign[tail[k], j] * inv[:, tail[k]]) * (e[:, j] * e[:, k]).sum(-1)
    t = np.linalg.solve(A, rhs[:, :, None])[:, :, 0]
    for a in range(3):
        dv = sum(sign[a, j] * t[:, j, None] * e[:, j] for j in range(3)) * inv[:, a, None]
        out[cl[:, a]] = v[:, a] + dv
    return out


def shake(pos, target, inv_mass, clusters, dist, tol, velocities=False, max_iterations=150):
    """SHAKE on clusters (centre, s1, s2, s3; -1 = unused) whose satellites carry no other constraint: the constraints of a cluster are
    visited in turn (Gauss-Seidel), each moving its two atoms along the OLD bond vector, until every one meets the Reference platform's
    convergence test (ReferenceCCMAAlgorithm.cpp:246-275: r'^2 within (1 +- tol)^2 d^2; velocities: |delta| <= tol).
    platforms/common/src/kernels/integrationUtilities.cc (applyShakeToHydrogens) is the reference's GPU counterpart."""
    pos, out = np.asarray(pos, np.float64), np.array(target, np.float64, copy=True)
    inv_mass = np.asarray(inv_mass, np.float64)
    lower, upper = 1 - 2 * tol + tol * tol, 1 + 2 * tol + tol * tol
    for c, d in zip(np.asarray(clusters, np.int64).reshape(-1, 4), np.asarray(dist, np.float64).reshape(-1, 4)[:, :3]):
        centre = c[0]
        sats = [(int(s), float(dk)) for s, dk in zip(c[1:], d) if s >= 0]
        r = {s: pos[centre] - pos[s] for s, _ in sats}
        for _ in range(max_iterations):
            done = True
            for s, dk in sats:
                reduced = 0.5 / (inv_mass[centre] + inv_mass[s])
                rp = out[centre] - out[s]
                if velocities:
                    delta = -2 * reduced * rp.dot(r[s]) / r[s].dot(r[s])
                    if abs(delta) <= tol:
                        continue
                else:
                    rp2 = rp.dot(rp)
                    if lower * dk * dk <= rp2 <= upper * dk * dk:
                        continue
                    delta = reduced * (dk * dk - rp2) / rp.dot(r[s])
                done = False
                out[centre] += r[s] * (delta * inv_mass[centre])
                out[s] -= r[s] * (delta * inv_mass[s])
            if done:
                break
    return out


def ccma_matrix(num_atoms, constraints, distance, mass, angles=(), element_cutoff=0.02):
    """The thresholded inverse of the constraint coupling matrix, as a dense array K with K[j, i] = (A^-1)[j, i] d_i / d_j where that
    exceeds element_cutoff in magnitude, else 0 (ReferenceCCMAAlgorithm.cpp:42-196; ReferenceConstraints.cpp:185 passes 0.02).
    A[j, k] = (1/m_shared) / (1/m_j0 + 1/m_j1) cos(angle between the constraints j and k at the atom they share): the angle from a third
    constraint closing the triangle when there is one, else from a HarmonicAngleForce term `angles` = [(a, b, c, theta)], else the pair
    is left uncoupled.  The reference inverts with a sparse QR; a dense inverse is the same matrix."""
    cons = [tuple(map(int, c)) for c in constraints]
    n = len(cons)
    d = np.asarray(distance, np.float64)
    mass = np.asarray(mass, np.float64)
    of_atom = [set() for _ in range(num_atoms)]
    for j, (a, b) in enumerate(cons):
        of_atom[a].add(j)
        of_atom[b].add(j)
    angle_of = {}
    for a, b, c, theta in angles:
        angle_of.setdefault((int(b), frozenset((int(a), int(c)))), float(theta))      # the first matching term wins, as in the reference's scan
    A = np.zeros((n, n))
    for j, (j0, j1) in enumerate(cons):
        A[j, j] = 1.0
        for k in sorted(of_atom[j0] | of_atom[j1]):
            if k == j:
                continue
            k0, k1 = cons[k]
            if j0 in (k0, k1):
                shared, end_j = j0, j1
            else:
                shared, end_j = j1, j0
            end_k = k1 if k0 == shared else k0
            scale = (1 / mass[shared]) / (1 / mass[j0] + 1 / mass[j1])
            closing = [o for o in sorted(of_atom[end_j]) if end_k in cons[o]]
            if closing:
                d3 = d[closing[0]]
                A[j, k] = scale * (d[j] ** 2 + d[k] ** 2 - d3 ** 2) / (2 * d[j] * d[k])
            elif (shared, frozenset((end_j, end_k))) in angle_of:
                A[j, k] = scale * np.cos(angle_of[(shared, frozenset((end_j, end_k)))])
    K = np.linalg.inv(A) * d[None, :] / d[:, None]
    K[np.abs(K) <= element_cutoff] = 0.0
    return K


def ccma(pos, target, inv_mass, constraints, distance, matrix, tol, velocities=False, max_iterations=150):
    """ReferenceCCMAAlgorithm.cpp:224-311: every iteration computes each constraint's own correction from the current `target`,
    multiplies the vector of corrections by the thresholded inverse coupling matrix and moves the atoms along the OLD bond vectors.
    Returns (corrected target, iterations used)."""
    pos, out = np.asarray(pos, np.float64), np.array(target, np.float64, copy=True)
    cons = np.asarray(constraints, np.int64).reshape(-1, 2)
    d = np.asarray(distance, np.float64)
    inv_mass = np.asarray(inv_mass, np.float64)
    i, j = cons[:, 0], cons[:, 1]
    r = pos[i] - pos[j]
    rr = (r * r).sum(1)
    reduced = 0.5 / (inv_mass[i] + inv_mass[j])
    lower, upper = 1 - 2 * tol + tol * tol, 1 + 2 * tol + tol * tol
    iterations = 0
    while iterations < max_iterations:
        rp = out[i] - out[j]
        if velocities:
            delta = -2 * reduced * (rp * r).sum(1) / rr
            converged = np.abs(delta) <= tol
        else:
            rp2 = (rp * rp).sum(1)
            delta = reduced * (d * d - rp2) / (rp * r).sum(1)
            converged = (rp2 >= lower * d * d) & (rp2 <= upper * d * d)
        if converged.all():
            break
        iterations += 1
        if matrix is not None:
            delta = matrix @ delta
        dr = r * delta[:, None]
        np.add.at(out, i, dr * inv_mass[i, None])
        np.add.at(out, j, -dr * inv_mass[j, None])
    return out, iterations


def verlet_step(pos, vel, force, mass, dt, constrain):
    """ReferenceVerletDynamics.cpp:76-119: v += F dt / m, x' = x + v dt, constrain(x, x') -> x', v = (x' - x) / dt.  `constrain(pos, trial)`
    returns the corrected trial positions (e.g. a composition of ccma / settle_positions in ReferenceConstraints' order).  Atoms of
    mass 0 do not move."""
    pos, vel, force = (np.asarray(a, np.float64) for a in (pos, vel, force))
    mass = np.asarray(mass, np.float64)
    inv = np.where(mass > 0, 1.0 / np.where(mass > 0, mass, 1.0), 0.0)[:, None]
    v = vel + force * inv * dt
    trial = np.where(inv > 0, pos + v * dt, pos)
    trial = constrain(pos, trial)
    v = np.where(inv > 0, (trial - pos) / dt, vel)
    return trial, v
