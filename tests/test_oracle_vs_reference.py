"""Pins the numpy oracle (oracle/*.py) to the real reference: the Reference platform compiled from the
reference's own sources (oracle/_ref/libOpenMM.so, driven through the harness), the golden values of
tests/TestEwald.h, and analytic known answers of tests/TestNonbondedForce.h."""
import numpy as np
import pytest

from conftest import max_rel_force_error
from openmm_amd import harness as H, testsystems as T
from oracle import nonbonded as ONB, pme as OPME


def reference_state(w, groups=-1, recip_group=False):
    system, nb = w.build()
    if recip_group:
        nb.setReciprocalSpaceForceGroup(1)
    ctx = H.Context(system, H.Integrator(H.VERLET, 0.001), "Reference")
    ctx.setPositions(w.positions)
    st = ctx.getState(getForces=True, getEnergy=True, groups=groups)
    ctx.close()
    return st


def test_coulomb_known_answer():
    # tests/TestNonbondedForce.h:50-72 testCoulomb: charges 0.5 and -1.5 at distance 2 -> force 138.935456*(1.5*0.5)/4
    pos = np.array([[0.0, 0, 0], [2.0, 0, 0]])
    f, e = ONB.direct_space(pos, [0.5, -1.5], [1, 1], [0, 0], ONB.NoCutoff)
    force = ONB.ONE_4PI_EPS0 * (-0.75) / 4.0
    assert np.allclose(f[0], [-force, 0, 0], atol=1e-9) and np.allclose(f[1], [force, 0, 0], atol=1e-9)
    assert abs(e - ONB.ONE_4PI_EPS0 * (-0.75) / 2.0) < 1e-9


def test_lj_known_answer():
    # tests/TestNonbondedForce.h:74-98 testLJ: sigma 1.2/1.4 eps 1/2 at distance 2 -> combined sigma 1.3, eps sqrt(2)
    pos = np.array([[0.0, 0, 0], [2.0, 0, 0]])
    f, e = ONB.direct_space(pos, [0, 0], [1.2, 1.4], [1.0, 2.0], ONB.NoCutoff)
    x = 1.3 / 2.0
    eps = np.sqrt(2.0)
    force = 4.0 * eps * (12 * x ** 12 - 6 * x ** 6) / 2.0
    assert np.allclose(f[0], [-force, 0, 0], atol=1e-9)
    assert abs(e - 4.0 * eps * (x ** 12 - x ** 6)) < 1e-9


@pytest.mark.parametrize("method", [H.NoCutoff, H.CutoffNonPeriodic, H.CutoffPeriodic])
def test_direct_space_matches_reference_platform(method):
    w = T.water_box(5, seed=3, method=method, cutoff=0.7)
    w.dispersion = False
    if method in (H.NoCutoff, H.CutoffNonPeriodic):
        w.box = None if method == H.NoCutoff else w.box
    st = reference_state(w)
    excl = [tuple(p) for p in w.exceptions[0]]
    f, e = ONB.direct_space(w.positions, w.charge, w.sigma, w.epsilon, method, w.cutoff, w.box, excl)
    assert max_rel_force_error(f, st.forces) < 1e-10
    assert abs(e - st.potentialEnergy) < 1e-8 * max(1.0, abs(st.potentialEnergy))


def test_pme_direct_and_reciprocal_match_reference_platform():
    w = T.water_box(6, seed=11)
    alpha = float(np.sqrt(-np.log(2 * w.ewald_tol)) / w.cutoff)
    w.pme_params = (alpha, 18, 20, 21)
    w.dispersion = False
    excl = [tuple(p) for p in w.exceptions[0]]
    direct = reference_state(w, groups=1, recip_group=True)
    recip = reference_state(w, groups=2, recip_group=True)
    f_dir, e_dir = ONB.direct_space(w.positions, w.charge, w.sigma, w.epsilon, ONB.PME, w.cutoff, w.box, excl, alpha)
    f_exc, e_exc = ONB.ewald_exclusion_correction(w.positions, w.charge, excl, alpha)
    assert max_rel_force_error(f_dir + f_exc, direct.forces) < 1e-10
    assert abs(e_dir + e_exc - direct.potentialEnergy) < 1e-8 * abs(direct.potentialEnergy)
    f_rec, e_rec = OPME.pme_exec(w.positions, w.charge, w.box, alpha, (18, 20, 21))
    e_rec += ONB.ewald_self_energy(w.charge, alpha)
    assert max_rel_force_error(f_rec, recip.forces) < 1e-9
    assert abs(e_rec - recip.potentialEnergy) < 1e-9 * abs(recip.potentialEnergy)


def test_pme_triclinic_matches_reference_platform():
    w = T.water_box(6, seed=5)
    L = w.box[0, 0]
    w.box = np.array([[L, 0, 0], [0.2 * L, L, 0], [-0.3 * L, 0.25 * L, L]])
    alpha = float(np.sqrt(-np.log(2 * w.ewald_tol)) / w.cutoff)
    w.pme_params = (alpha, 20, 20, 20)
    w.dispersion = False
    recip = reference_state(w, groups=2, recip_group=True)
    f_rec, e_rec = OPME.pme_exec(w.positions, w.charge, w.box, alpha, (20, 20, 20))
    assert max_rel_force_error(f_rec, recip.forces) < 1e-9


def test_ewald_ksum_and_gromacs_golden_energy(golden):
    # tests/TestEwald.h:98-220: amorphous NaCl, Ewald, tolerance 1e-5; golden energy from Gromacs -3.82047e5 (1e-5)
    g = golden("nacl_amorph.npz")
    pos = g["positions"]
    n = len(pos)
    q = np.concatenate([np.ones(n // 2), -np.ones(n // 2)])
    L, rc, tol = float(g["box"]), float(g["cutoff"]), float(g["ewald_tol"])
    box = np.eye(3) * L
    alpha = np.sqrt(-np.log(2 * tol)) / rc
    # NonbondedForceImpl::calcEwaldParameters (NonbondedForceImpl.cpp:144-158): smallest odd kmax with error < tol
    def find_kmax(width):
        # EwaldErrorFunction (:127-136) and findZero (:186-197)
        value = lambda kk: tol - 0.05 * np.sqrt(width * alpha) * kk * np.exp(-(kk * np.pi / (width * alpha)) ** 2)
        k = 10
        if value(k) > 0.0:
            while value(k) > 0.0 and k > 0:
                k -= 1
            k += 1
        else:
            while value(k) < 0.0:
                k += 1
        return k + 1 if k % 2 == 0 else k
    kmax = (find_kmax(L),) * 3
    f_dir, e_dir = ONB.direct_space(pos, q, np.ones(n), np.zeros(n), ONB.Ewald, rc, box, (), alpha)
    f_rec, e_rec = ONB.ewald_reciprocal(pos, q, box, alpha, kmax)
    e = e_dir + e_rec + ONB.ewald_self_energy(q, alpha)
    assert abs(e - float(g["gromacs_energy"])) < 1e-5 * abs(float(g["gromacs_energy"]))
    # and the same system on the real Reference platform
    w = T.Workload("nacl")
    w.positions, w.box = pos, box
    w.masses = np.concatenate([np.full(n // 2, 22.99), np.full(n // 2, 35.45)])
    w.charge, w.sigma, w.epsilon = q, np.ones(n), np.zeros(n)
    w.method, w.cutoff, w.ewald_tol, w.dispersion = H.Ewald, rc, tol, False
    st = reference_state(w)
    assert abs(e - st.potentialEnergy) < 1e-9 * abs(st.potentialEnergy)
    assert max_rel_force_error(f_dir + f_rec, st.forces) < 1e-9


def test_exceptions_match_reference_platform():
    rng = np.random.default_rng(0)
    w = T.water_box(4, seed=9, method=H.NoCutoff)
    w.box = None
    w.dispersion = False
    pairs = np.array([[0, 5], [3, 10], [7, 20]])
    qq, sig, eps = rng.normal(size=3), 0.2 + 0.1 * rng.random(3), rng.random(3)
    w.exceptions = (np.concatenate([w.exceptions[0], pairs]), np.concatenate([w.exceptions[1], qq]),
                    np.concatenate([w.exceptions[2], sig]), np.concatenate([w.exceptions[3], eps]))
    st = reference_state(w)
    excl = [tuple(p) for p in w.exceptions[0]]
    f, e = ONB.direct_space(w.positions, w.charge, w.sigma, w.epsilon, ONB.NoCutoff, exclusions=excl)
    f14, e14 = ONB.exceptions_14(w.positions, [(int(p[0]), int(p[1]), a, b, c) for p, a, b, c in zip(pairs, qq, sig, eps)])
    assert max_rel_force_error(f + f14, st.forces) < 1e-10
    assert abs(e + e14 - st.potentialEnergy) < 1e-8 * abs(st.potentialEnergy)
