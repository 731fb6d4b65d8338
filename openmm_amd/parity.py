"""Force-parity statistics of SURVEY.md §8(d), shared by tests/test_gpu_platform.py and bench.py.

max-rel-err = max_i |F_hip,i - F_ref,i| / max(|F_ref,i|, F_rms)   (target 1e-4),
median      = median_i 2 |dF_i| / (|F_ref,i| + |F_hip,i|)           (the statistic of docs-source/usersguide/library/07_testing_validation.rst:121-128).

The truncated direct-space force is discontinuous at r = cutoff (for two TIP3P charges the jump is ~0.2 kJ/mol/nm, 2e-4 of
the RMS force).  A pair whose double-precision distance lies within float32 coordinate resolution of the cutoff
(ulp(6 nm) = 4.8e-7 nm) may legitimately fall on the other side in a float32 pair kernel, exactly as on the reference's
single/mixed precision GPU platforms.  Atoms of pairs within `band` of the cutoff are therefore reported separately and held
to the size of that jump, all other atoms to the target."""
import numpy as np

EDGE_BAND_NM = 1.5e-6


def force_parity(positions, box, cutoff, f_hip, f_ref, band=EDGE_BAND_NM):
    """box: 3x3 rectangular (or None for non-periodic).  -> dict of statistics"""
    from scipy.spatial import cKDTree
    n = len(positions)
    norm_ref = np.linalg.norm(f_ref, axis=1)
    rms = float(np.sqrt((f_ref ** 2).sum(1).mean()))
    diff = np.linalg.norm(f_hip - f_ref, axis=1)
    rel = diff / np.maximum(norm_ref, rms)
    median = float(np.median(2 * diff / (norm_ref + np.linalg.norm(f_hip, axis=1))))
    if box is not None:
        L = np.diag(np.asarray(box, float))
        tree = cKDTree(np.mod(positions, L[None, :]), boxsize=L)
    else:
        L = None
        tree = cKDTree(positions)
    cand = tree.query_pairs(cutoff + band, output_type="ndarray")
    d = positions[cand[:, 0]] - positions[cand[:, 1]]
    if L is not None:
        d -= np.round(d / L[None, :]) * L[None, :]
    r = np.linalg.norm(d, axis=1)
    edge = cand[np.abs(r - cutoff) < band]
    edge_atoms = np.unique(edge)
    interior = np.ones(n, bool)
    interior[edge_atoms] = False
    return {"max_rel_err": float(rel[interior].max()), "max_rel_err_all_atoms": float(rel.max()),
            "max_rel_err_cutoff_edge_atoms": float(rel[~interior].max()) if len(edge_atoms) else 0.0,
            "cutoff_edge_pairs": int(len(edge)), "cutoff_edge_atoms": int(len(edge_atoms)), "edge_band_nm": band,
            "median_rel_diff": median, "rms_force": rms}
