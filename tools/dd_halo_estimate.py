"""Halo size of the planned decomposition: ranks own contiguous ranges of space-filling-curve-sorted
32-atom blocks; the halo of a rank = blocks of other ranks within cutoff + padding of one of its blocks.
Compares curve-range ownership with geometric 2 x 2 x 2 cubes on a water-density lattice.  CPU only (numpy/scipy).

    python tools/dd_halo_estimate.py [n_side] [ranks]      # n_side^3 waters, default 69 (985 527 atoms), 8 ranks
"""
import sys
import numpy as np
from scipy.spatial import cKDTree


def hilbert_key(ix, iy, iz, bits):
    """Skilling's transpose -> Hilbert index for integer coordinates (vectorised)."""
    X = [ix.astype(np.int64).copy(), iy.astype(np.int64).copy(), iz.astype(np.int64).copy()]
    M = 1 << (bits - 1)
    Q = M
    while Q > 1:
        P = Q - 1
        for i in range(3):
            flip = (X[i] & Q) != 0
            X[0] = np.where(flip, X[0] ^ P, X[0])
            t = np.where(~flip, (X[0] ^ X[i]) & P, 0)
            X[0] ^= t
            X[i] ^= t
        Q >>= 1
    for i in range(1, 3):
        X[i] ^= X[i - 1]
    t = np.zeros_like(X[0])
    Q = M
    while Q > 1:
        t = np.where((X[2] & Q) != 0, t ^ (Q - 1), t)
        Q >>= 1
    for i in range(3):
        X[i] ^= t
    key = np.zeros_like(X[0])
    for b in range(bits - 1, -1, -1):
        for i in range(3):
            key = (key << 1) | ((X[i] >> b) & 1)
    return key


def main():
    n_side = int(sys.argv[1]) if len(sys.argv) > 1 else 69
    ranks = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    nw = n_side ** 3
    L = (nw / 33.4) ** (1.0 / 3.0)
    rlist = 0.9 * 1.1
    rng = np.random.default_rng(0)
    g = np.stack(np.meshgrid(*[np.arange(n_side)] * 3, indexing="ij"), -1).reshape(-1, 3)
    centres = (g + 0.5 + 0.15 * (rng.random((nw, 3)) - 0.5)) * (L / n_side)          # molecule centres
    bits = 7
    cell = np.minimum((centres / L * (1 << bits)).astype(np.int64), (1 << bits) - 1)
    order = np.argsort(hilbert_key(cell[:, 0], cell[:, 1], cell[:, 2], bits), kind="stable")
    centres = centres[order]
    # 32-atom blocks = 32/3 molecules; work with blocks of 32 molecules' centres thinned to the same count of blocks
    per_block = 32 / 3.0
    nblocks = int(np.ceil(nw / per_block))
    block_of = np.minimum((np.arange(nw) / per_block).astype(np.int64), nblocks - 1)
    lo = np.full((nblocks, 3), np.inf); hi = np.full((nblocks, 3), -np.inf)
    np.minimum.at(lo, block_of, centres); np.maximum.at(hi, block_of, centres)
    bc, bh = 0.5 * (lo + hi), 0.5 * (hi - lo) + 0.1          # + molecule radius
    print("%d waters (%d atoms), L = %.2f nm, %d blocks, median block half-extent %.2f nm" % (nw, 3 * nw, L, nblocks, float(np.median(bh))))
    tree = cKDTree(np.mod(bc, L), boxsize=L)
    reach = rlist + 2 * float(np.percentile(np.linalg.norm(bh, axis=1), 95))
    pairs = tree.query_pairs(reach, output_type="ndarray")
    d = bc[pairs[:, 0]] - bc[pairs[:, 1]]
    d -= np.round(d / L) * L
    gap = np.maximum(np.abs(d) - bh[pairs[:, 0]] - bh[pairs[:, 1]], 0.0)
    close = (gap ** 2).sum(1) < rlist ** 2
    pairs = pairs[close]
    print("block pairs within %.2f nm: %d (%.0f per block)" % (rlist, len(pairs), 2.0 * len(pairs) / nblocks))

    def halo(owner, label):
        a, b = owner[pairs[:, 0]], owner[pairs[:, 1]]
        cross = a != b
        out = []
        for r in range(ranks):
            mine = int((owner == r).sum())
            h = np.unique(np.concatenate([pairs[cross & (a == r), 1], pairs[cross & (b == r), 0]]))
            out.append((mine, len(h)))
        own = np.mean([o for o, _ in out]); hl = np.mean([h for _, h in out]); hmax = max(h for _, h in out)
        print("%-34s owned blocks/rank %.0f, halo blocks/rank mean %.0f max %d  (halo/owned %.2f; halo atoms mean %.0f)" % (label, own, hl, hmax, hl / own, hl * 32))

    halo(np.minimum(np.arange(nblocks) * ranks // nblocks, ranks - 1), "Hilbert-range ownership:")
    side = round(ranks ** (1 / 3))
    if side ** 3 == ranks:
        c = np.minimum((np.mod(bc, L) / L * side).astype(np.int64), side - 1)
        halo(c[:, 0] * side * side + c[:, 1] * side + c[:, 2], "geometric %dx%dx%d cubes:" % (side, side, side))


if __name__ == "__main__":
    main()
