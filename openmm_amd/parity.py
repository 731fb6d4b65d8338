"""Force-parity statistics of SURVEY.md §8(d), shared by tests/test_gpu_platform.py and bench.py.

max-rel-err = max_i |F_hip,i - F_ref,i| / max(|F_ref,i|, F_rms)   (target 1e-4),
median      = median_i 2 |dF_i| / (|F_ref,i| + |F_hip,i|)           (the statistic of docs-source/usersguide/library/07_testing_validation.rst:121-128).

The headline figure is `max_rel_err_all_atoms`: the maximum over EVERY atom.

The truncated direct-space force is discontinuous at r = cutoff (for two TIP3P charges the jump is ~0.2 kJ/mol/nm, 2e-4 of
the RMS force), so a pair whose double-precision distance lies within the rounding of the pair kernel's separation of the
cutoff may fall on the other side.  The pair kernel computes with block-relative coordinates (error ~1e-7 nm whatever the
box size), so that band is 2e-7 nm wide: typically zero or one pair per configuration.  Atoms of pairs inside the band are
also reported on their own (`max_rel_err_cutoff_edge_atoms`) as a diagnostic; they are part of the headline."""
import numpy as np

EDGE_BAND_NM = 2.0e-7


def force_parity(positions, box, cutoff, f_hip, f_ref, band=EDGE_BAND_NM, tol=1e-4, subset=None, rms=None):
    """box: 3x3 rectangular (or None for non-periodic).  `subset`: atom indices the two force arrays refer to (a sampled
    golden); default all atoms.  -> dict of statistics.

    Only atoms above `tol` are examined for cutoff-edge pairs (their neighbours within cutoff + band are searched with a
    k-d tree), so the cost does not grow with the ~150 pairs per atom of the whole system."""
    from scipy.spatial import cKDTree
    idx = np.arange(len(positions)) if subset is None else np.asarray(subset)
    norm_ref = np.linalg.norm(f_ref, axis=1)
    if rms is None:
        rms = float(np.sqrt((f_ref ** 2).sum(1).mean()))
    diff = np.linalg.norm(f_hip - f_ref, axis=1)
    rel = diff / np.maximum(norm_ref, rms)
    median = float(np.median(2 * diff / (norm_ref + np.linalg.norm(f_hip, axis=1))))
    above = np.nonzero(rel > tol)[0]
    edge_local, edge_pairs = [], 0
    if len(above):
        if box is not None:
            L = np.diag(np.asarray(box, float))
            wrapped = np.mod(positions, L[None, :])
            wrapped[wrapped >= L[None, :]] = 0.0
            tree = cKDTree(wrapped, boxsize=L)
        else:
            L, wrapped = None, positions
            tree = cKDTree(positions)
        for a in above:
            i = idx[a]
            nbrs = np.array(tree.query_ball_point(wrapped[i], cutoff + 10 * band), dtype=np.int64)
            d = positions[nbrs] - positions[i]
            if L is not None:
                d -= np.round(d / L[None, :]) * L[None, :]
            r = np.linalg.norm(d, axis=1)
            n_edge = int((np.abs(r - cutoff) < band).sum())
            if n_edge:
                edge_local.append(a)
                edge_pairs += n_edge
    interior = np.ones(len(rel), bool)
    interior[edge_local] = False
    return {"max_rel_err": float(rel[interior].max()), "max_rel_err_all_atoms": float(rel.max()),
            "max_rel_err_cutoff_edge_atoms": float(rel[~interior].max()) if len(edge_local) else 0.0,
            "atoms_above_tolerance": int(len(above)), "cutoff_edge_pairs": int(edge_pairs), "cutoff_edge_atoms": int(len(edge_local)),
            "edge_band_nm": band, "median_rel_diff": median, "rms_force": rms}
