"""Pins oracle/constraints.py (SETTLE, SHAKE, CCMA, the Verlet step around them) to the real Reference platform
(build/openmm/lib/libOpenMM.so through the harness): Context.applyConstraints / applyVelocityConstraints, force-free and forced
VerletIntegrator steps, and the coupling matrix ReferenceCCMAAlgorithm builds.  SURVEY.md §8 rows a19, a22-a24; no GPU."""
import ctypes as C

import numpy as np

from openmm_amd import harness as H, testsystems as T
from oracle import constraints as OC


def bare_system(w, angles=True):
    """Masses + constraints (+ the HarmonicAngleForce the CCMA matrix reads, with k = 0 so that it exerts nothing)."""
    s = H.System()
    s.addParticles(w.masses)
    s.addConstraints(*w.constraints)
    if angles and w.angles is not None:
        s.addHarmonicAngleForce(w.angles[0], w.angles[1], np.zeros(len(w.angles[1])))
    return s


def reference_partition(system, n_atoms):
    """(settle clusters [n,3], their (leg, base) distances, ccma constraint atoms [m,2], dense thresholded inverse [m,m])"""
    cap = n_atoms
    atoms, dist = np.full(3 * cap, -1, np.int32), np.zeros(2 * cap)
    n = H.lib().omm_reference_settle_clusters(system.h, atoms.ctypes.data_as(C.POINTER(C.c_int)), dist.ctypes.data_as(C.POINTER(C.c_double)), cap)
    rows, cols, vals = np.zeros(1 << 20, np.int32), np.zeros(1 << 20, np.int32), np.zeros(1 << 20)
    ca, nc = np.zeros(2 * cap, np.int32), C.c_int(0)
    ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int))
    nnz = H.lib().omm_reference_ccma_matrix(system.h, ip(rows), ip(cols), vals.ctypes.data_as(C.POINTER(C.c_double)), 1 << 20, ip(ca), cap, C.byref(nc))
    assert nnz >= 0
    K = np.zeros((nc.value, nc.value))
    K[rows[:nnz], cols[:nnz]] = vals[:nnz]
    return atoms[:3 * n].reshape(n, 3), dist[:2 * n].reshape(n, 2), ca[:2 * nc.value].reshape(-1, 2), K


def oracle_constrain(w, settle, sd, cc, cd, K, tol):
    inv = 1.0 / w.masses

    def constrain(pos, trial):          # ReferenceConstraints::apply: CCMA first, then SETTLE (ReferenceConstraints.cpp:194-199)
        out = trial
        if len(cc):
            out, _ = OC.ccma(pos, out, inv, cc, cd, K, tol)
        return OC.settle_positions(pos, out, w.masses, settle, sd[:, 0], sd[:, 1])
    return constrain


def zoo():
    w = T.constraint_zoo()
    system = bare_system(w)
    settle, sd, cc, K_ref = reference_partition(system, w.num_atoms)
    dist_of = {tuple(sorted(map(int, p))): d for p, d in zip(*w.constraints)}
    cd = np.array([dist_of[tuple(sorted(map(int, p)))] for p in cc])
    return w, system, settle, sd, cc, cd, K_ref


def test_ccma_matrix_is_the_reference_platforms():
    w, system, settle, sd, cc, cd, K_ref = zoo()
    assert len(settle) == (w.num_atoms - 150) // 3 and len(cc) == 99
    angles = [(int(a), int(b), int(c), float(t)) for (a, b, c), t in zip(w.angles[0], w.angles[1])]
    K = OC.ccma_matrix(w.num_atoms, cc, cd, w.masses, angles)
    assert np.array_equal(K != 0, K_ref != 0)
    assert np.abs(K - K_ref).max() < 1e-12
    # and without the angle terms: only constraint triangles couple (none here) -> a different matrix, which the reference agrees on
    _, _, cc2, K_ref2 = reference_partition(bare_system(w, angles=False), w.num_atoms)
    K2 = OC.ccma_matrix(w.num_atoms, cc2, cd, w.masses, ())
    assert np.abs(K2 - K_ref2).max() < 1e-12 and np.abs(K2 - K).max() > 1e-2


def test_settle_shake_ccma_positions_match_context_apply_constraints():
    w, system, settle, sd, cc, cd, K = zoo()
    rng = np.random.default_rng(3)
    start = w.positions + rng.normal(0, 0.004, w.positions.shape)          # every constraint violated by ~5 %
    tol = 1e-10
    ctx = H.Context(system, H.Integrator(H.VERLET, 0.001, constraintTolerance=tol), "Reference")
    ctx.setPositions(start)
    ctx.applyConstraints(tol)          # ReferenceKernels.cpp:323-327: before == trial == the current positions
    ref = ctx.getState(getPositions=True).positions
    got = oracle_constrain(w, settle, sd, cc, cd, K, tol)(start, start)
    assert np.abs(got - ref).max() < 1e-12
    # every constraint holds
    p, d = w.constraints
    assert np.abs(np.linalg.norm(got[p[:, 0]] - got[p[:, 1]], axis=1) / d - 1).max() < 1e-7          # SETTLE keeps the float-rounded lengths the reference stores (ReferenceConstraints.cpp:120-130)
    # SHAKE on the X-H clusters alone converges to what the Reference platform's CCMA converges to
    shake_w = T.small_solvated_chain(seed=5)
    sys2 = bare_system(shake_w)
    settle2, sd2, cc2, K2 = reference_partition(sys2, shake_w.num_atoms)
    ctx2 = H.Context(sys2, H.Integrator(H.VERLET, 0.001, constraintTolerance=tol), "Reference")
    ctx2.setPositions(start)
    ctx2.applyConstraints(tol)
    ref2 = ctx2.getState(getPositions=True).positions
    clusters, dist = shake_clusters(cc2, shake_w)
    got2 = OC.shake(start, start, 1.0 / shake_w.masses, clusters, dist, tol)
    got2 = OC.settle_positions(start, got2, shake_w.masses, settle2, sd2[:, 0], sd2[:, 1])
    assert np.abs(got2 - ref2).max() < 2e-10
    ctx.close()
    ctx2.close()


def shake_clusters(cc, w):
    """centre = the atom several constraints share (or the heavier one), satellites = its constraint partners"""
    dist_of = {tuple(sorted(map(int, p))): d for p, d in zip(*w.constraints)}
    by_centre = {}
    for a, b in cc:
        a, b = int(a), int(b)
        centre, sat = (a, b) if w.masses[a] > w.masses[b] else (b, a)
        by_centre.setdefault(centre, []).append(sat)
    clusters, dist = [], []
    for centre, sats in sorted(by_centre.items()):
        assert len(sats) <= 3
        clusters.append([centre] + sats + [-1] * (3 - len(sats)))
        dist.append([dist_of[tuple(sorted((centre, s)))] for s in sats] + [0.0] * (4 - len(sats)))
    return np.array(clusters, np.int32), np.array(dist)


def test_velocity_constraints_match_context_apply_velocity_constraints():
    w, system, settle, sd, cc, cd, K = zoo()
    rng = np.random.default_rng(4)
    tol = 1e-10
    ctx = H.Context(system, H.Integrator(H.VERLET, 0.001, constraintTolerance=tol), "Reference")
    ctx.setPositions(w.positions)
    ctx.applyConstraints(tol)
    pos = ctx.getState(getPositions=True).positions
    vel = rng.normal(0, 1.0, pos.shape) / np.sqrt(w.masses)[:, None]
    ctx.setVelocities(vel)
    ctx.applyVelocityConstraints(tol)
    ref = ctx.getState(getVelocities=True).velocities
    got, _ = OC.ccma(pos, vel, 1.0 / w.masses, cc, cd, K, tol, velocities=True)
    got = OC.settle_velocities(pos, got, w.masses, settle)
    assert np.abs(got - ref).max() < 1e-11
    p = w.constraints[0]
    rel = ((got[p[:, 0]] - got[p[:, 1]]) * (pos[p[:, 0]] - pos[p[:, 1]])).sum(1)
    assert np.abs(rel).max() < 1e-9
    ctx.close()


def test_verlet_steps_with_constraints_match_the_reference_platform():
    """Force-free Verlet steps hand DISTINCT before / trial positions to SETTLE and CCMA (ReferenceVerletDynamics.cpp:76-119)."""
    w, system, settle, sd, cc, cd, K = zoo()
    rng = np.random.default_rng(6)
    tol = 1e-9
    dt = 0.002
    ctx = H.Context(system, H.Integrator(H.VERLET, dt, constraintTolerance=tol), "Reference")
    ctx.setPositions(w.positions)
    ctx.applyConstraints(tol)
    pos = ctx.getState(getPositions=True).positions
    vel = rng.normal(0, 1.6, pos.shape) / np.sqrt(w.masses)[:, None]          # ~300 K
    ctx.setVelocities(vel)
    constrain = oracle_constrain(w, settle, sd, cc, cd, K, tol)
    zero = np.zeros_like(pos)
    for step in range(3):
        ctx.integrator.step(1)
        st = ctx.getState(getPositions=True, getVelocities=True)
        pos, vel = OC.verlet_step(pos, vel, zero, w.masses, dt, constrain)
        assert np.abs(pos - st.positions).max() < 1e-11, step
        assert np.abs(vel - st.velocities).max() < 1e-8, step
    ctx.close()
