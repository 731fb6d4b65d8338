"""openmm_amd/parity.py: the force-parity statistics used by the GPU tests and by bench.py (CPU-only checks of the helper)."""
import numpy as np

from openmm_amd.parity import force_parity


def _system(seed=3, n=2000, L=3.0):
    rng = np.random.default_rng(seed)
    return rng.random((n, 3)) * L, np.eye(3) * L, rng.normal(size=(n, 3)) * 100.0


def test_statistics_of_a_uniform_perturbation():
    pos, box, f = _system()
    rms = np.sqrt((f ** 2).sum(1).mean())
    p = force_parity(pos, box, 0.9, f + 1e-5 * rms * np.array([1.0, 0.0, 0.0]), f)
    assert abs(p["rms_force"] - rms) < 1e-9 * rms
    assert 0.99e-5 < p["max_rel_err_all_atoms"] <= 1.0001e-5           # |dF| / max(|F|, rms) peaks where |F| <= rms
    assert p["max_rel_err"] <= p["max_rel_err_all_atoms"]
    assert p["median_rel_diff"] > 0


def test_atoms_of_cutoff_edge_pairs_are_reported_separately():
    pos, box, f = _system()
    cutoff = 0.9
    # put atom 1 exactly (to 1e-7 nm) one cutoff away from atom 0, across the periodic boundary
    pos[0] = [0.05, 1.0, 1.0]
    pos[1] = [0.05 - cutoff + 1e-7 + 3.0, 1.0, 1.0]
    g = f.copy()
    g[0] += 0.3          # an error only on the two atoms of the edge pair
    g[1] -= 0.3
    p = force_parity(pos, box, cutoff, g, f)
    assert p["cutoff_edge_pairs"] >= 1 and p["cutoff_edge_atoms"] >= 2
    assert p["max_rel_err"] == 0.0                                   # every other atom is exact
    assert p["max_rel_err_cutoff_edge_atoms"] > 1e-3
    assert p["max_rel_err_all_atoms"] == p["max_rel_err_cutoff_edge_atoms"]
    # the same error on atoms that are not part of an edge pair counts against the target
    h = f.copy()
    h[5] += 0.3
    q = force_parity(pos, box, cutoff, h, f)
    assert q["max_rel_err"] > 1e-3


def test_non_periodic_and_rectangular_boxes():
    pos, box, f = _system()
    assert force_parity(pos, None, 0.9, f, f)["max_rel_err_all_atoms"] == 0.0
    rect = np.diag([3.0, 4.0, 5.0])
    assert force_parity(pos, rect, 0.9, f * (1 + 1e-6), f)["max_rel_err"] < 2e-6


def test_sampled_golden():
    """forces known for a subset of the atoms only (tests/golden/reference_forces_water985527_sample.npz)"""
    pos, box, f = _system()
    idx = np.arange(0, len(pos), 7)
    g = f[idx].copy()
    g[3] += 0.5
    p = force_parity(pos, box, 0.9, g, f[idx], subset=idx, rms=float(np.sqrt((f ** 2).sum(1).mean())))
    assert p["atoms_above_tolerance"] == 1 and p["cutoff_edge_atoms"] == 0 and p["max_rel_err"] > 1e-3
