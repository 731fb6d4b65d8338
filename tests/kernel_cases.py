"""Kernel-level parity cases driven through the C ABI (include/openmm_hip_kernels.h) with ctypes.

The same cases run against the product library on a GPU (tests/test_gpu_kernels.py, `-m gpu`) and against
the CPU-emulated twin (tests/test_emu_host_logic.py) -- the latter only checks indexing/host logic."""
import ctypes as C

import numpy as np

from openmm_amd import capi
from oracle import nonbonded as ONB, pme as OPME


LAST_FIXED_POINT_FORCES = None
LAST_NL = None


def lattice_positions(rng, n, box3, jitter=0.25):
    m = int(np.ceil(n ** (1 / 3)))
    g = np.stack(np.meshgrid(*[np.arange(m)] * 3, indexing="ij"), -1).reshape(-1, 3)[:n]
    frac = (g + 0.5 + jitter * (rng.random((n, 3)) - 0.5)) / m
    return frac @ box3


LAST_SINGLE_FRACTION = 0.0


def box6(box3):
    return (C.c_double * 6)(box3[0, 0], box3[1, 0], box3[1, 1], box3[2, 0], box3[2, 1], box3[2, 2])


def run_direct_space(K, n, method, cutoff, L, excl, triclinic=False, switch=None, seed=0, grid=64, compact=False, cells=False, fused_pme=None, block_range=None, energy=True, lj_free_tail=False, ewald_tol=5e-4, positions=None, charges=None, edge_path=True, prepare=False, prune=True):
    """-> (forces[n,3], energy, oracle forces, oracle energy, nl state)

    compact=False: random slot order, list built by ommhip_nl_update on wrapped coordinates (general image search).
    compact=True: slots sorted along a Morton curve and the per-step entry ommhip_nl_step, which stores image-coherent
    blocks -- with a box wider than 2 (block + cutoff) this drives the pair kernel's single-image path.
    fused_pme=(nx, ny, nz): instead of the list builder and the pair kernel on their own, the single-stream sequence of a
    whole PME evaluation -- ommhip_nl_prepare (with the clears), ommhip_force_front (list build + charge spreading),
    ommhip_pairs_with_fft (pair kernel on the three FFT launches), ommhip_pme_reciprocal(interpolate only) -- against
    direct + reciprocal space of the oracle.
    block_range=(first, count): the list is built for these i-blocks only (force decomposition between ranks); the raw
    fixed-point force buffer of the evaluation is left in LAST_FIXED_POINT_FORCES.
    energy=False: forces only -- plain Ewald/PME in a rectangular box then takes the polynomial form of the real-space force
    (nonbonded.hip, METHOD 9) on the single-image path; the energy returned is 0.
    lj_free_tail=True: epsilon = 0 for the atoms in the slots 12..31 of every block (a water box after the platform's
    in-block ordering): the single-image path leaves the Lennard-Jones arithmetic out for them.
    prepare=True: the per-step entry ommhip_nl_step (double positions in) also for the random slot order.
    prune=False: without the per-step pruned list (the pair kernel walks the rows as built).
    positions / charges: given instead of drawn; edge_path=False: no posq_rel_lo, the float separation decides at the cutoff."""
    rng = np.random.default_rng(seed)
    box3 = np.eye(3) * L
    if triclinic:
        box3 = np.array([[L, 0, 0], [0.2 * L, 0.9 * L, 0], [-0.3 * L, 0.25 * L, 1.1 * L]])
    pos = lattice_positions(rng, n, box3) if positions is None else np.asarray(positions, np.float64)
    q = rng.normal(0, 0.5, n)
    q -= q.mean()
    if charges is not None:
        q = np.asarray(charges, np.float64)
    sig = 0.2 + 0.1 * rng.random(n)
    eps = rng.random(n)
    padded = (n + 31) // 32 * 32
    # a non-trivial slot order exercises atomOfSlot/slotOfAtom
    perm = rng.permutation(n).astype(np.int32)
    if compact:
        cell = np.floor(np.mod(pos, L) / 0.45).astype(np.int64)
        key = np.zeros(n, np.int64)
        for bit in range(6):
            for d in range(3):
                key |= ((cell[:, d] >> bit) & 1) << (3 * bit + d)
        perm = np.argsort(key, kind="stable").astype(np.int32)
    atom_of_slot = np.full(padded, -1, np.int32)
    atom_of_slot[:n] = perm
    if lj_free_tail:
        eps[perm[(np.arange(n) % 32) >= 12]] = 0.0
    slot_of_atom = np.empty(n, np.int32)
    slot_of_atom[perm] = np.arange(n, dtype=np.int32)
    ex = [[] for _ in range(n)]
    for i, j in excl:
        ex[i].append(j)
        ex[j].append(i)
    start = np.zeros(n + 1, np.int32)
    start[1:] = np.cumsum([len(e) for e in ex])
    flat = np.array([a for e in ex for a in e] + [0], np.int32)
    pos4 = np.zeros((n, 4))
    pos4[:, :3] = pos
    wrap = np.zeros((n, 4), np.int32)
    periodic = method in (ONB.CutoffPeriodic, ONB.Ewald, ONB.PME)
    d_pos, d_wrap, d_aos, d_soa = K.upload(pos4), K.upload(wrap), K.upload(atom_of_slot), K.upload(slot_of_atom)
    d_posq, d_se = K.upload(np.zeros((padded, 4), np.float32)), K.upload(np.zeros((padded, 2), np.float32))
    d_q, d_sig, d_eps = K.upload(q), K.upload(sig), K.upload(eps)
    b6 = box6(box3)
    K.set_slot_params(d_q, d_sig, d_eps, d_aos, padded, d_posq, d_se, None)
    K.positions_to_posq(d_pos, d_wrap, d_aos, padded, b6, d_posq, None)
    nl = capi.NeighborList()
    nl.num_atoms, nl.padded_atoms = n, padded
    maxc = 8192
    nl.max_chunks = maxc
    nl.pbc = 0 if not periodic else (2 if triclinic else 1)
    nl.cutoff = cutoff if method != ONB.NoCutoff else 0.0
    nl.padding = 0.1 * cutoff if method != ONB.NoCutoff else 0.0
    for i in range(6):
        nl.box[i] = b6[i]
    nl.posq, nl.posq_ref = d_posq, K.upload(np.zeros((padded, 4), np.float32))
    nl.posq_rel = K.upload(np.zeros((padded, 4), np.float32))
    if edge_path:
        nl.posq_rel_lo = K.upload(np.zeros((padded, 4), np.float32))
    nl.atom_of_slot, nl.slot_of_atom = d_aos, d_soa
    nl.excl_start, nl.excl_atoms = K.upload(start), K.upload(flat)
    st = np.zeros(capi.NL_STATE_INTS, np.int32)
    st[0] = 1
    nl.state = K.upload(st)
    nb = padded // 32
    nl.block_center, nl.block_half = K.upload(np.zeros((nb, 4), np.float32)), K.upload(np.zeros((nb, 4), np.float32))
    nl.chunk_info = K.upload(np.zeros((maxc, 2), np.int32))
    nl.row_j = K.upload(np.zeros(maxc * capi.CHUNK_ROWS * capi.ROW, np.int32))
    nl.row_mask = K.upload(np.zeros(maxc * capi.CHUNK_ROWS * capi.ROW, np.uint32))
    if prune:      # the per-step pruned list (what the platform always gives the builder)
        nl.chunk_info_inner = K.upload(np.zeros((maxc, 2), np.int32))
        nl.row_j_inner = K.upload(np.zeros(maxc * capi.CHUNK_ROWS * capi.ROW, np.int32))
        nl.row_mask_inner = K.upload(np.zeros(maxc * capi.CHUNK_ROWS * capi.ROW, np.uint32))
        nl.block_runs = K.upload(np.zeros(17 * nb, np.int32))
        nl.posq_ref_inner = K.upload(np.zeros((padded, 4), np.float32))
        nl.inner_padding = 0.03 * cutoff
    if block_range is not None:
        nl.first_block, nl.owned_blocks = block_range
    if cells:      # force the cell-binned candidate search of large systems at test size
        nl.max_cells = nb + 64
        nl.cell_start = K.upload(np.zeros(2 * nl.max_cells + 2, np.int32))
        nl.cell_blocks = K.upload(np.zeros(2 * nb, np.int32))
        nl.cell_boxes = K.upload(np.zeros((2 * nb, 4), np.float32))
        nl.cell_meta = K.upload(np.zeros(4, np.float32))
        nl.cell_min_blocks = 1
    pm = None
    if fused_pme is not None:
        assert method == ONB.PME and not triclinic
        ng = fused_pme
        pm = make_pme(K, ng, box3, float(np.sqrt(-np.log(2 * ewald_tol)) / cutoff))
        pm.grid_precleared = 1
        K.pme_build_eterm(C.byref(pm), None)
        d_f, d_e = K.upload(np.full(3 * padded, 12345, np.int64)), K.upload(np.zeros(grid))       # the force buffer starts dirty: nl_prepare clears it
        K.memset(pm.grid_real, 0x55, 4 * ng[0] * ng[1] * ng[2], None)                                # ... and the charge grid
        K.nl_prepare(C.byref(nl), d_pos, d_wrap, d_f, 8 * 3 * padded, pm.grid_real, 4 * ng[0] * ng[1] * ng[2], None)
        K.force_front(C.byref(nl), C.byref(pm), 0, None, d_pos, d_f, d_e, grid, 1, None)
    elif compact or prepare:
        K.nl_step(C.byref(nl), d_pos, d_wrap, None)
    else:
        K.nl_update(C.byref(nl), None)
    state = K.download(nl.state, capi.NL_STATE_INTS, np.int32)
    p = capi.NonbondedParams()
    p.ewald = 1 if method in (ONB.Ewald, ONB.PME) else 0
    alpha = float(np.sqrt(-np.log(2 * ewald_tol)) / cutoff) if p.ewald else 0.0
    p.ewald_alpha = alpha
    if method in (ONB.CutoffPeriodic, ONB.CutoffNonPeriodic):
        p.krf, p.crf = ONB.reaction_field_constants(cutoff, 78.3)
    if switch:
        p.use_switch, p.switch_distance = 1, switch
    p.direct_grid = grid
    if pm is not None:
        K.pairs_with_fft(C.byref(nl), C.byref(p), d_se, C.byref(pm), d_f, d_e, grid, 1 if energy else 0, None)
        pm.phases = capi.PME_INTERPOLATE_ONLY
        K.pme_reciprocal(C.byref(pm), d_posq, padded, d_f, d_e, grid, 1, None)
    else:
        d_f, d_e = K.upload(np.zeros(3 * padded, np.int64)), K.upload(np.zeros(grid))
        K.nb_direct(C.byref(nl), C.byref(p), d_se, d_f, d_e, grid, 1 if energy else 0, None)
    global LAST_FIXED_POINT_FORCES, LAST_NL
    LAST_NL = (nl, slot_of_atom)
    LAST_FIXED_POINT_FORCES = K.download(d_f, (3, padded), np.int64)
    f = LAST_FIXED_POINT_FORCES.astype(np.float64) / 2 ** 32
    e = float(K.download(d_e, grid, np.float64).sum())
    forces = f[:, slot_of_atom].T
    # fraction of i-blocks that qualified for the pair kernel's single-image path (coverage check for compact=True)
    global LAST_SINGLE_FRACTION
    half = K.download(nl.block_half, (nb, 4), np.float32)
    ok = (half[:, 3] != 0) & np.all(half[:, :3] + cutoff < 0.5 * np.diag(box3)[None, :], axis=1)
    LAST_SINGLE_FRACTION = float(ok.mean()) if periodic and not triclinic else 0.0
    f_or, e_or = ONB.direct_space(pos, q, sig, eps, method, cutoff, box3, excl, alpha, switch_distance=switch)
    if pm is not None:
        f_rec, e_rec = OPME.pme_exec(pos, q.astype(np.float32).astype(np.float64), box3, alpha, fused_pme)
        f_or, e_or = f_or + f_rec, e_or + e_rec
    return forces, e, f_or, e_or, state


def run_cutoff_edge(K, n=1200, L=3.4, cutoff=0.7, seed=5, energy=False, edge_path=True, triclinic=False, compact=True):
    """A dense box in which up to 100 pairs of unit charges are moved to distances rc (1 + e), |e| from 1e-10 to 3e-7, on both sides
    of the cutoff: the float32 separation of such a pair (rounding ~1e-7 nm) cannot tell the side, the truncated pair force there
    (`jump`) is of order 1 kJ/mol/nm, so a wrong decision shows as an error of one jump on two atoms.
    -> (forces, oracle forces, planted atoms [2m], jump, energy, oracle energy)"""
    from scipy.special import erfc
    rng = np.random.default_rng(seed)
    box3 = np.eye(3) * L
    if triclinic:
        box3 = np.array([[L, 0, 0], [0.2 * L, 0.9 * L, 0], [-0.3 * L, 0.25 * L, 1.1 * L]])
    pos = lattice_positions(rng, n, box3)                       # what run_direct_space would draw with this seed
    q = np.where(rng.random(n) < 0.5, 1.0, -1.0)
    q[: n // 2 * 2 : 2] = 1.0
    q[1 : n // 2 * 2 : 2] = -1.0
    inv = np.linalg.inv(box3)
    mags = np.array([1e-10, 1e-9, 3e-9, 1e-8, 3e-8, 1e-7, 3e-7])
    used = np.zeros(n, bool)
    planted = []
    for i in range(n):
        if used[i] or len(planted) >= 100:
            continue
        d = pos - pos[i]
        d -= np.round(d @ inv) @ box3                            # nearest image for boxes this much wider than the cutoff
        r = np.linalg.norm(d, axis=1)
        cand = np.where((np.abs(r - cutoff) < 0.02) & ~used)[0]
        cand = cand[cand != i]
        if len(cand) == 0:
            continue
        j = int(cand[0])
        k = len(planted)
        e = mags[k % len(mags)] * (1.0 if (k // len(mags)) % 2 == 0 else -1.0)
        pos[j] = pos[i] + d[j] * (cutoff * (1.0 + e) / r[j])
        used[i] = used[j] = True
        planted.append((i, j))
    assert len(planted) >= 40
    alpha = float(np.sqrt(-np.log(2 * 5e-4)) / cutoff)
    jump = 138.935 * (erfc(alpha * cutoff) / cutoff ** 2 + 2 * alpha / np.sqrt(np.pi) * np.exp(-(alpha * cutoff) ** 2) / cutoff)
    f, en, f_or, e_or, state = run_direct_space(K, n, ONB.PME, cutoff, L, [], triclinic=triclinic, compact=compact, energy=energy, positions=pos, charges=q,
                                                 edge_path=edge_path, seed=seed, prepare=True)
    return f, f_or, np.array(planted).reshape(-1), float(jump), en, e_or


def twiddles(n):
    k = np.arange(n)
    w = np.exp(-2j * np.pi * k / n)
    return np.stack([w.real, w.imag], -1).astype(np.float32)


def make_pme(K, ng, box3, alpha):
    nx, ny, nz = ng
    nzc = nz // 2 + 1
    pm = capi.Pme()
    pm.nx, pm.ny, pm.nz = ng
    pm.alpha = alpha
    b = box6(box3)
    for i in range(6):
        pm.box[i] = b[i]
    pm.moduli_x, pm.moduli_y, pm.moduli_z = (K.upload(OPME.bspline_moduli(n)) for n in ng)
    pm.eterm = K.upload(np.zeros(nx * ny * nzc, np.float32))
    pm.grid_real = K.upload(np.zeros(nx * ny * nz, np.float32))
    pm.grid_complex = K.upload(np.zeros(nx * ny * nzc * 2, np.float32))
    pm.twiddle_x, pm.twiddle_y, pm.twiddle_z = (K.upload(twiddles(n)) for n in ng)
    return pm


def run_fft(K, ng, seed=1, fft_mode=0):
    """-> (forward max error relative to max |ref|, round-trip max abs error)   pattern of TestCudaFFT3D.cpp:52-108"""
    rng = np.random.default_rng(seed)
    nx, ny, nz = ng
    nzc = nz // 2 + 1
    pm = make_pme(K, ng, np.eye(3) * 3.0, 3.0)
    pm.fft_mode = fft_mode
    g = rng.normal(size=ng).astype(np.float32)
    K.memcpy_h2d(pm.grid_real, g.ctypes.data_as(C.c_void_p), g.nbytes, None)
    K.fft3d_r2c_c2r(C.byref(pm), 1, None)
    c = K.download(pm.grid_complex, (nx, ny, nzc, 2), np.float32)
    c = c[..., 0] + 1j * c[..., 1]
    ref = np.fft.rfftn(g.astype(np.float64))
    fwd = float(np.abs(c - ref).max() / np.abs(ref).max())
    K.fft3d_r2c_c2r(C.byref(pm), 0, None)
    back = K.download(pm.grid_real, ng, np.float32)
    return fwd, float(np.abs(back / np.prod(ng) - g).max())


def run_pme(K, n, ng, L, triclinic=False, seed=2, alpha=2.6, tiles=None, sort_cell=None, shift=0.0):
    """tiles=(cap,): spread_mode 2 (one workgroup per 16^3 grid tile gathers the atoms of the blocks that reach it) with per-tile
    block lists of `cap` entries (a small cap forces the scan-everything path); sort_cell: slots sorted along a Morton curve so
    that blocks are compact; shift: all coordinates moved by this many box lengths (blocks outside the primary cell)."""
    rng = np.random.default_rng(seed)
    box3 = np.eye(3) * L
    if triclinic:
        box3 = np.array([[L, 0, 0], [0.2 * L, 0.9 * L, 0], [-0.3 * L, 0.25 * L, 1.1 * L]])
    pos = rng.random((n, 3)) @ box3
    if sort_cell is not None:
        cell = np.floor(pos / sort_cell).astype(np.int64)
        key = np.zeros(n, np.int64)
        for bit in range(8):
            for d in range(3):
                key |= ((cell[:, d] >> bit) & 1) << (3 * bit + d)
        pos = pos[np.argsort(key, kind="stable")]
    pos = pos + shift * L
    q = rng.normal(0, 0.5, n)
    q -= q.mean()
    padded = (n + 31) // 32 * 32
    posq = np.zeros((padded, 4), np.float32)
    posq[:n, :3] = pos
    posq[:n, 3] = q
    d_posq, d_f, d_e = K.upload(posq), K.upload(np.zeros(3 * padded, np.int64)), K.upload(np.zeros(64))
    pm = make_pme(K, ng, box3, alpha)
    if tiles is not None:
        nb = padded // 32
        center, half = np.zeros((nb, 4), np.float32), np.full((nb, 4), -1e30, np.float32)
        for b in range(nb):
            p = posq[b * 32:min((b + 1) * 32, n), :3]
            if len(p):
                center[b, :3] = 0.5 * (p.min(0) + p.max(0))
                half[b, :3] = 0.5 * (p.max(0) - p.min(0)) + 1e-6
        ntiles = int(np.prod([(g + 15) // 16 for g in ng]))
        pm.spread_mode, pm.tile_cap, pm.max_tiles = 2, tiles[0], ntiles
        pm.tile_count, pm.tile_blocks = K.upload(np.zeros(ntiles, np.int32)), K.upload(np.zeros(ntiles * tiles[0], np.int32))
        pm.block_center, pm.block_half = K.upload(center), K.upload(half)
        pm.max_charge = float(np.abs(q).max())
        K.memset(pm.grid_real, 0x55, 4 * ng[0] * ng[1] * ng[2], None)        # tiles must write every cell: start from a dirty grid
    K.pme_build_eterm(C.byref(pm), None)
    K.pme_reciprocal(C.byref(pm), d_posq, padded, d_f, d_e, 64, 1, None)
    f = K.download(d_f, (3, padded), np.int64).astype(np.float64) / 2 ** 32
    e = float(K.download(d_e, 64, np.float64).sum())
    f_or, e_or = OPME.pme_exec(posq[:n, :3].astype(np.float64), q.astype(np.float32).astype(np.float64), box3, alpha, ng)
    return f[:, :n].T, e, f_or, e_or


def run_list_completeness(K, n, cutoff, box_lengths, sort_cell, cells, seed=0, padding=0.1, prune=True):
    """Builds the neighbour list of n random atoms in a rectangular periodic box (slots sorted along a Morton curve through
    `sort_cell` bins, per-step entry ommhip_nl_step) and decodes it.  -> (missing pairs, duplicated pairs, pairs within the cutoff,
    list entries, nl state): every pair within the cutoff must be in the list exactly once (the brute-force side is
    scipy's periodic cKDTree)."""
    from scipy.spatial import cKDTree
    rng = np.random.default_rng(seed)
    Ls = np.asarray(box_lengths, float)
    box3 = np.diag(Ls)
    pos = rng.random((n, 3)) * Ls
    padded = (n + 31) // 32 * 32
    cell = np.floor(pos / sort_cell).astype(np.int64)
    key = np.zeros(n, np.int64)
    for bit in range(8):
        for d in range(3):
            key |= ((cell[:, d] >> bit) & 1) << (3 * bit + d)
    perm = np.argsort(key, kind="stable").astype(np.int32)
    atom_of_slot = np.full(padded, -1, np.int32)
    atom_of_slot[:n] = perm
    slot_of_atom = np.empty(n, np.int32)
    slot_of_atom[perm] = np.arange(n, dtype=np.int32)
    pos4 = np.zeros((n, 4))
    pos4[:, :3] = pos
    d_pos, d_wrap = K.upload(pos4), K.upload(np.zeros((n, 4), np.int32))
    d_aos, d_soa = K.upload(atom_of_slot), K.upload(slot_of_atom)
    d_posq, d_se = K.upload(np.zeros((padded, 4), np.float32)), K.upload(np.zeros((padded, 2), np.float32))
    K.set_slot_params(K.upload(np.zeros(n)), K.upload(np.full(n, 0.3)), K.upload(np.ones(n)), d_aos, padded, d_posq, d_se, None)
    b6 = box6(box3)
    nl = capi.NeighborList()
    nl.num_atoms, nl.padded_atoms = n, padded
    nb = padded // 32
    maxc = 40 * nb
    nl.max_chunks = maxc
    nl.pbc = 1
    nl.cutoff, nl.padding = cutoff, padding * cutoff
    for i in range(6):
        nl.box[i] = b6[i]
    nl.posq, nl.posq_ref = d_posq, K.upload(np.zeros((padded, 4), np.float32))
    nl.posq_rel = K.upload(np.zeros((padded, 4), np.float32))
    nl.atom_of_slot, nl.slot_of_atom = d_aos, d_soa
    nl.excl_start, nl.excl_atoms = K.upload(np.zeros(n + 1, np.int32)), K.upload(np.zeros(1, np.int32))
    st = np.zeros(capi.NL_STATE_INTS, np.int32)
    st[0] = 1
    nl.state = K.upload(st)
    nl.block_center, nl.block_half = K.upload(np.zeros((nb, 4), np.float32)), K.upload(np.zeros((nb, 4), np.float32))
    nl.chunk_info = K.upload(np.zeros((maxc, 2), np.int32))
    nl.row_j = K.upload(np.zeros(maxc * capi.CHUNK_ROWS * capi.ROW, np.int32))
    nl.row_mask = K.upload(np.zeros(maxc * capi.CHUNK_ROWS * capi.ROW, np.uint32))
    if prune:      # the per-step pruned list (what the platform always gives the builder)
        nl.chunk_info_inner = K.upload(np.zeros((maxc, 2), np.int32))
        nl.row_j_inner = K.upload(np.zeros(maxc * capi.CHUNK_ROWS * capi.ROW, np.int32))
        nl.row_mask_inner = K.upload(np.zeros(maxc * capi.CHUNK_ROWS * capi.ROW, np.uint32))
        nl.block_runs = K.upload(np.zeros(17 * nb, np.int32))
        nl.posq_ref_inner = K.upload(np.zeros((padded, 4), np.float32))
        nl.inner_padding = 0.03 * cutoff
    if cells:
        nl.max_cells = 4 * nb + 64
        nl.cell_start = K.upload(np.zeros(2 * nl.max_cells + 2, np.int32))
        nl.cell_blocks = K.upload(np.zeros(2 * nb, np.int32))
        nl.cell_boxes = K.upload(np.zeros((2 * nb, 4), np.float32))
        nl.cell_meta = K.upload(np.zeros(4, np.float32))
        nl.cell_min_blocks = 1
    K.nl_step(C.byref(nl), d_pos, d_wrap, None)
    state = K.download(nl.state, capi.NL_STATE_INTS, np.int32)
    assert state[2] == 0 and 0 < int(state[1]) <= maxc, state

    def decode(chunks, d_info, d_j, d_m):
        info = K.download(d_info, (maxc, 2), np.int32)[:chunks]
        rows_j = K.download(d_j, (maxc, capi.CHUNK_ROWS, capi.ROW), np.int32)[:chunks]
        rows_m = K.download(d_m, (maxc, capi.CHUNK_ROWS, capi.ROW), np.uint32)[:chunks]
        pairs, entries = [], 0
        for c in range(chunks):
            X, nrows = int(info[c, 0]), int(info[c, 1]) & 0xFF
            j = rows_j[c, :nrows].ravel()
            m = rows_m[c, :nrows].ravel()
            keep = m != 0
            j, m = j[keep], m[keep]
            entries += len(j)
            bits = (m[:, None] >> np.arange(32, dtype=np.uint32)[None, :]) & 1
            jj, ii = np.nonzero(bits)
            si, sj = X * 32 + ii, j[jj]
            a, b = atom_of_slot[si], atom_of_slot[sj]
            assert (a >= 0).all() and (b >= 0).all()
            pairs.append(np.stack([np.minimum(a, b), np.maximum(a, b)], 1))
        return pairs, entries

    pairs, entries = decode(int(state[1]), nl.chunk_info, nl.row_j, nl.row_mask)
    global LAST_ENTRIES_AS_BUILT
    LAST_ENTRIES_AS_BUILT = entries
    if prune:
        # what the pair kernel walks: the rows re-packed by the same launch to the j atoms within the cutoff itself of each block's box
        assert state[10] == 0 and 0 < int(state[7]) <= int(state[1]), state
        pairs, entries = decode(int(state[7]), nl.chunk_info_inner, nl.row_j_inner, nl.row_mask_inner)
    listed = np.concatenate(pairs).astype(np.int64)
    listed_key = listed[:, 0] * n + listed[:, 1]
    uniq, counts = np.unique(listed_key, return_counts=True)
    tree = cKDTree(pos, boxsize=Ls)
    true = tree.query_pairs(cutoff, output_type="ndarray").astype(np.int64)
    true_key = np.minimum(true[:, 0], true[:, 1]) * n + np.maximum(true[:, 0], true[:, 1])
    missing = int((~np.isin(true_key, uniq)).sum())
    return missing, int((counts > 1).sum()), len(true_key), entries, state


def run_transpose_reduce(K, waves=7, seed=3):
    """The pair kernel's force reduction on its own (ommhip_test_transpose_reduce): -> (what the kernel returns, plain float64 column sums).
    Lane l of a wave must end with the wave-wide total of partial sum l >> 1."""
    rng = np.random.default_rng(seed)
    v = rng.normal(size=(waves, 64, 32)).astype(np.float32)
    v[0] = (np.arange(64)[:, None] * 32 + np.arange(32)[None, :]).astype(np.float32)          # a pattern in which every misrouted element shows
    d_in = K.upload(v)
    d_out = K.malloc(4 * waves * 64)
    K.test_transpose_reduce(d_in, d_out, waves, None)
    got = K.download(d_out, (waves, 64), np.float32)
    K.free(d_in); K.free(d_out)
    expect = v.astype(np.float64).sum(axis=1)[:, np.arange(64) >> 1]
    return got, expect


# ------------------------------------------------------------------------------------------------ AMOEBA valence terms (kernels/valence.hip)
VALENCE_KINDS = {"poly_bond": (0, 2), "poly_angle": (1, 3), "inplane_angle": (2, 4), "out_of_plane_bend": (3, 4), "stretch_bend": (4, 3), "pi_torsion": (5, 6), "torsion_torsion": (6, 6)}
RAD = 180.0 / np.pi


def _valence_case(kind, rng, n_terms):
    """random molecules of the term's size near a sensible geometry -> (positions, atoms, params, coefficients, grids or None, oracle energy function)"""
    from oracle import valence as OV
    per = VALENCE_KINDS[kind][1]
    na = 5 if kind == "torsion_torsion" else per
    atoms = np.arange(n_terms * na).reshape(n_terms, na)
    # a zig-zag chain of `na` atoms per term, bond length ~0.15 nm, jittered; planar centres get their three partners around them
    base = np.zeros((na, 3))
    for k in range(1, na):
        base[k] = base[k - 1] + 0.15 * np.array([np.cos(0.6 * (k % 2)), np.sin(0.6 * (k % 2)) * (1 if k % 4 < 2 else -1), 0.05 * k])
    if kind in ("inplane_angle", "out_of_plane_bend"):
        base = np.array([[0.14, 0.0, 0.0], [0.0, 0.0, 0.02], [-0.07, 0.12, 0.0], [-0.07, -0.12, 0.0]])        # 1, centre 2 slightly out of the plane, 3, 4
    if kind == "pi_torsion":
        base = np.array([[-0.07, 0.12, 0.0], [-0.07, -0.12, 0.01], [0.0, 0.0, 0.0], [0.14, 0.0, 0.0], [0.21, 0.12, 0.03], [0.21, -0.12, -0.02]])
    if kind == "torsion_torsion":
        # a chain with both dihedrals well away from +-180 degrees (the tabulated polynomial is not periodic: a finite difference across
        # the seam would be meaningless), built from internal coordinates: bond 0.15 nm, angle 110 degrees, dihedrals 65 and -75 degrees
        def place(a, b, c, bond, angle, dihedral):
            bc = (c - b) / np.linalg.norm(c - b)
            nrm = np.cross(b - a, bc); nrm /= np.linalg.norm(nrm)
            m = np.cross(nrm, bc)
            d2 = np.array([-bond * np.cos(angle), bond * np.sin(angle) * np.cos(dihedral), bond * np.sin(angle) * np.sin(dihedral)])
            return c + d2[0] * bc + d2[1] * m + d2[2] * nrm
        base = np.zeros((5, 3)); base[1] = (0.15, 0, 0); base[2] = base[1] + 0.15 * np.array([np.cos(np.radians(70)), np.sin(np.radians(70)), 0])
        base[3] = place(base[0], base[1], base[2], 0.15, np.radians(110), np.radians(65))
        base[4] = place(base[1], base[2], base[3], 0.15, np.radians(110), np.radians(-75))
    pos = (base[None] + (0.01 if kind == "torsion_torsion" else 0.02) * rng.normal(size=(n_terms, na, 3)) + rng.uniform(0, 3, size=(n_terms, 1, 3))).reshape(-1, 3)
    c = np.zeros(6)
    grids = None
    if kind == "poly_bond":
        params = np.stack([0.14 + 0.02 * rng.random(n_terms), 2e5 * (0.5 + rng.random(n_terms))], -1); c[:2] = (-25.5, 379.3125)
        energy = lambda p: OV.poly_bond(p, atoms, params, c)
    elif kind in ("poly_angle", "inplane_angle"):
        params = np.stack([100.0 + 20 * rng.random(n_terms), 0.02 + 0.05 * rng.random(n_terms)], -1); c[:5] = (-0.014, 5.6e-5, -7e-7, 2.2e-8, RAD)
        energy = (lambda p: OV.poly_angle(p, atoms, params, c)) if kind == "poly_angle" else (lambda p: OV.inplane_angle(p, atoms, params, c))
    elif kind == "out_of_plane_bend":
        params = (0.005 + 0.02 * rng.random(n_terms))[:, None]; c[:5] = (-0.014, 5.6e-5, -7e-7, 2.2e-8, RAD)
        energy = lambda p: OV.out_of_plane_bend(p, atoms, params, c)
    elif kind == "stretch_bend":
        params = np.stack([0.14 + 0.01 * rng.random(n_terms), 0.15 + 0.01 * rng.random(n_terms), 1.8 + 0.3 * rng.random(n_terms), 50 * rng.random(n_terms), 80 * rng.random(n_terms)], -1); c[0] = RAD
        energy = lambda p: OV.stretch_bend(p, atoms, params, c)
    elif kind == "pi_torsion":
        params = (20 + 30 * rng.random(n_terms))[:, None]
        energy = lambda p: OV.pi_torsion(p, atoms, params, c)
    else:
        # two maps, each a bicubic polynomial of the two angles tabulated with its exact derivatives: the kernel's bicubic patch must
        # reproduce it; the sixth atom of a term: the chirality marker (another atom of the system bonded nowhere in particular) or -1
        n_grid = 25
        coef = rng.normal(size=(2, 4, 4)) * np.array([1.0, 1e-2, 1e-4, 1e-6])[None, :, None] * np.array([1.0, 1e-2, 1e-4, 1e-6])[None, None, :]
        ang = np.linspace(-180.0, 180.0, n_grid)
        def surface_of(m):
            def f(x, y, dx=0, dy=0):
                out = np.zeros(np.broadcast(x, y).shape)
                for i in range(4):
                    for j in range(4):
                        if i < dx or j < dy: continue
                        ci = np.prod(np.arange(i, i - dx, -1)) if dx else 1.0
                        cj = np.prod(np.arange(j, j - dy, -1)) if dy else 1.0
                        out = out + coef[m, i, j] * ci * cj * x ** (i - dx) * y ** (j - dy)
                return out
            return f
        tables = []
        for m in range(2):
            f = surface_of(m)
            X, Y = np.meshgrid(ang, ang, indexing="ij")
            tables.append(np.stack([X, Y, f(X, Y), f(X, Y, 1, 0), f(X, Y, 0, 1), f(X, Y, 1, 1)], -1))
        grids = np.ascontiguousarray(np.stack(tables))
        which = rng.integers(0, 2, n_terms)
        marker = np.where(rng.random(n_terms) < 0.6, (atoms[:, 2] + 7) % (n_terms * na), -1)          # some atom of another term
        atoms = np.concatenate([atoms, marker[:, None]], 1)
        params = np.stack([which * grids[0].size, np.full(n_terms, n_grid)], -1).astype(np.float64)
        def energy(p):
            e = np.zeros(n_terms)
            for m in range(2):
                sel = which == m
                if sel.any(): e[sel] = OV.torsion_torsion(p, atoms[sel], lambda x, y: surface_of(m)(x, y))
            return e
    return pos, atoms, params, c, grids, energy


def run_valence(K, kind, n_terms=40, seed=1):
    """-> (forces from ommhip_valence_forces, energy, oracle forces (central differences of the numpy energy), oracle energy)"""
    from oracle import valence as OV
    rng = np.random.default_rng(seed)
    pos, atoms, params, c, grids, energy = _valence_case(kind, rng, n_terms)
    n = len(pos)
    padded = (n + 31) // 32 * 32
    perm = rng.permutation(padded)[:n]                   # atom -> slot: the kernels index forces by slot
    pos4 = np.zeros((n, 4)); pos4[:, :3] = pos
    lst = capi.ValenceList()
    lst.kind, lst.num_terms = VALENCE_KINDS[kind][0], n_terms
    lst.atoms = K.upload(atoms.astype(np.int32)); lst.params = K.upload(params.astype(np.float64))
    for i in range(6): lst.coefficients[i] = c[i]
    lst.grids = K.upload(grids) if grids is not None else None
    pos_d, slot_d = K.upload(pos4), K.upload(perm.astype(np.int32))
    force_d = K.upload(np.zeros(3 * padded, dtype=np.int64)); e_d = K.upload(np.zeros(64)); out_d = K.upload(np.zeros(3 * n))
    K.valence_forces(1, C.byref(lst), pos_d, slot_d, padded, force_d, e_d, 64, 1, None)
    K.forces_to_atom_order(force_d, slot_d, n, padded, out_d, None)
    K.stream_sync(None)
    f = K.download(out_d, (n, 3), np.float64)
    e = K.download(e_d, (64,), np.float64).sum()
    return f, e, OV.forces(energy, pos), energy(pos).sum()


# ------------------------------------------------------------------------------------------------ CustomIntegrator interpreter (kernels/custom_integrator.hip)
def run_vm(K, n=300, seed=2):
    """Three computations in one launch of ommhip_vm_per_dof -- r0 = x + 2 v - f / m * g0 ; v = v * exp(-g1) + sqrt(abs(r0)) ; x = select(step(x), x, -x) + v --
    and a sum of m v^2 / 2 in a launch of its own, programs written by hand in the postfix form of include/openmm_hip_kernels.h.
    -> dict of (kernel result, numpy result) pairs; one massless particle must be left alone"""
    rng = np.random.default_rng(seed)
    pos = np.zeros((n, 4)); pos[:, :3] = rng.normal(size=(n, 3))
    mass = 1.0 + 10 * rng.random(n); mass[5] = 0.0
    vel = np.zeros((n, 4)); vel[:, :3] = rng.normal(size=(n, 3)); vel[:, 3] = np.where(mass > 0, 1.0 / np.where(mass > 0, mass, 1.0), 0.0)
    force = rng.normal(size=(n, 3)) * 100
    g = np.array([0.37, 0.05])
    CONST, VAR, GLOBAL, ADD, SUB, MUL, DIV = 0, 1, 2, 3, 4, 5, 6
    NEG, SQRT, EXP, STEP, MULC, ABS, SELECT = 8, 9, 10, 27, 33, 37, 40
    prog = [
        # r0 = x + 2 v - f / m * g0
        (VAR, 0, 0), (VAR, 1, 0), (MULC, 0, 2.0), (ADD, 0, 0), (VAR, 2, 0), (VAR, 3, 0), (DIV, 0, 0), (GLOBAL, 0, 0), (MUL, 0, 0), (SUB, 0, 0),
        # v = v exp(-g1) + sqrt(abs(r0))
        (VAR, 1, 0), (GLOBAL, 1, 0), (NEG, 0, 0), (EXP, 0, 0), (MUL, 0, 0), (VAR, 6, 0), (ABS, 0, 0), (SQRT, 0, 0), (ADD, 0, 0),
        # x = select(step(x), x, -x) + v
        (VAR, 0, 0), (STEP, 0, 0), (VAR, 0, 0), (VAR, 0, 0), (NEG, 0, 0), (SELECT, 0, 0), (VAR, 1, 0), (ADD, 0, 0),
        # sum: m v v / 2
        (VAR, 3, 0), (VAR, 1, 0), (MUL, 0, 0), (VAR, 1, 0), (MUL, 0, 0), (MULC, 0, 0.5)]
    instr = (capi.VmInstruction * len(prog))(*[capi.VmInstruction(op, arg, val) for op, arg, val in prog])
    prog_d = K.malloc(C.sizeof(instr)); K.memcpy_h2d(prog_d, C.cast(instr, C.c_void_p), C.c_size_t(C.sizeof(instr)), None); K.stream_sync(None)
    st = capi.VmState()
    st.num_atoms, st.num_per_dof = n, 1
    st.pos, st.vel = K.upload(pos), K.upload(vel)
    st.per_dof = K.upload(np.zeros(3 * n)); st.globals = K.upload(g); st.program = prog_d; st.seed = 1
    st.sum_scratch = K.upload(np.zeros(4096)); st.sum_result = K.upload(np.zeros(2))
    force_d = K.upload(force.reshape(-1))
    steps = (capi.VmStep * 3)(capi.VmStep(0, 10, 2, 0, force_d, 0), capi.VmStep(10, 9, 1, 0, None, 0), capi.VmStep(19, 8, 0, 0, None, 0))
    K.vm_per_dof(C.byref(st), 3, steps, None)
    total = (capi.VmStep * 1)(capi.VmStep(27, 6, -1, 0, None, 0))
    K.vm_per_dof(C.byref(st), 1, total, None)
    K.stream_sync(None)
    out_pos, out_vel = K.download(st.pos, (n, 4), np.float64), K.download(st.vel, (n, 4), np.float64)
    r0 = K.download(st.per_dof, (n, 3), np.float64)
    s = K.download(st.sum_result, (2,), np.float64)[0]
    live = mass > 0
    m3 = mass[:, None]
    e_r0 = np.where(live[:, None], pos[:, :3] + 2 * vel[:, :3] - force / np.where(live, mass, 1.0)[:, None] * g[0], 0.0)
    e_v = np.where(live[:, None], vel[:, :3] * np.exp(-g[1]) + np.sqrt(np.abs(e_r0)), vel[:, :3])
    e_x = np.where(live[:, None], np.where(pos[:, :3] >= 0, pos[:, :3], -pos[:, :3]) + e_v, pos[:, :3])
    e_s = (0.5 * m3 * e_v ** 2)[live].sum()
    return {"r0": (r0, e_r0), "v": (out_vel[:, :3], e_v), "x": (out_pos[:, :3], e_x), "sum": (np.array([s]), np.array([e_s])),
            "inverse_mass_kept": (out_vel[:, 3], vel[:, 3])}


def run_vm_bonds(K, n=200, n_bonds=500, seed=4):
    """ommhip_vm_bond_forces: a Morse bond with a global scale, E = s D (1 - exp(-a (r - r0)))^2, energy and dE/dr programs written by hand in
    the postfix form of include/openmm_hip_kernels.h; periodic in a triclinic box whose minimum image the bonds need (atoms scattered over
    several cells).  -> (forces, energy, oracle forces (central differences of the numpy energy), oracle energy)"""
    from oracle import valence as OV
    rng = np.random.default_rng(seed)
    box = np.array([2.0, 0.3, 2.2, -0.4, 0.5, 2.4])          # ax, bx, by, cx, cy, cz
    a_, b_, c_ = np.array([box[0], 0, 0]), np.array([box[1], box[2], 0]), np.array([box[3], box[4], box[5]])
    base = rng.uniform(0, 1, size=(n, 3)) @ np.stack([a_, b_, c_])
    atoms = np.stack([rng.integers(0, n, n_bonds), rng.integers(0, n, n_bonds)], -1).astype(np.int32)
    atoms = atoms[atoms[:, 0] != atoms[:, 1]]
    n_bonds = len(atoms)
    pos = base + rng.integers(-2, 3, size=(n, 3)) @ np.stack([a_, b_, c_])          # every atom in some other cell
    D, a, r0 = 50 + 50 * rng.random(n_bonds), 1 + rng.random(n_bonds), 0.5 + rng.random(n_bonds)
    scale = 0.7

    def energy(p):
        d = p[atoms[:, 1]] - p[atoms[:, 0]]
        d -= np.floor(d[:, 2] / c_[2] + 0.5)[:, None] * c_
        d -= np.floor(d[:, 1] / b_[1] + 0.5)[:, None] * b_
        d -= np.floor(d[:, 0] / a_[0] + 0.5)[:, None] * a_
        r = np.linalg.norm(d, axis=1)
        return scale * D * (1 - np.exp(-a * (r - r0))) ** 2

    CONST, VAR, GLOBAL, ADD, SUB, MUL = 0, 1, 2, 3, 4, 5
    NEG, EXP, SQUARE, MULC = 8, 10, 29, 33
    u = [(VAR, 7, 0), (NEG, 0, 0), (VAR, 0, 0), (VAR, 8, 0), (SUB, 0, 0), (MUL, 0, 0), (EXP, 0, 0)]          # exp(-a (r - r0)); parameters: 6 = D, 7 = a, 8 = r0
    e_prog = [(GLOBAL, 0, 0), (VAR, 6, 0), (MUL, 0, 0), (CONST, 0, 1.0)] + u + [(SUB, 0, 0), (SQUARE, 0, 0), (MUL, 0, 0)]
    d_prog = [(GLOBAL, 0, 0), (VAR, 6, 0), (MUL, 0, 0), (VAR, 7, 0), (MUL, 0, 0), (MULC, 0, 2.0), (CONST, 0, 1.0)] + u + [(SUB, 0, 0), (MUL, 0, 0)] + u + [(MUL, 0, 0)]
    prog = e_prog + d_prog
    instr = (capi.VmInstruction * len(prog))(*[capi.VmInstruction(op, arg, val) for op, arg, val in prog])
    prog_d = K.malloc(C.sizeof(instr)); K.memcpy_h2d(prog_d, C.cast(instr, C.c_void_p), C.c_size_t(C.sizeof(instr)), None); K.stream_sync(None)
    stride = (n_bonds + 2) // 3 * 3
    params = np.zeros((3, stride)); params[0, :n_bonds], params[1, :n_bonds], params[2, :n_bonds] = D, a, r0
    b = capi.VmBonds()
    b.num_bonds, b.num_params, b.param_stride, b.periodic = n_bonds, 3, stride, 1
    b.atoms, b.params, b.program = K.upload(atoms.reshape(-1)), K.upload(params.reshape(-1)), prog_d
    b.energy_first, b.energy_count, b.deriv_first, b.deriv_count = 0, len(e_prog), len(e_prog), len(d_prog)
    b.globals = K.upload(np.array([scale]))
    for k in range(6): b.box[k] = box[k]
    padded = (n + 31) // 32 * 32 + 32
    perm = rng.permutation(padded)[:n]
    pos4 = np.zeros((n, 4)); pos4[:, :3] = pos
    pos_d, slot_d = K.upload(pos4), K.upload(perm.astype(np.int32))
    force_d = K.upload(np.zeros(3 * padded, dtype=np.int64)); e_d = K.upload(np.zeros(64)); out_d = K.upload(np.zeros(3 * n))
    K.vm_bond_forces(C.byref(b), pos_d, slot_d, padded, force_d, e_d, 64, 1, None)
    K.forces_to_atom_order(force_d, slot_d, n, padded, out_d, None)
    K.stream_sync(None)
    return K.download(out_d, (n, 3), np.float64), K.download(e_d, (64,), np.float64).sum(), OV.forces(energy, pos, h=1e-6), energy(pos).sum()


def run_vm_angles(K, n=150, n_angles=400, seed=6):
    """ommhip_vm_angle_forces: E = k (theta - t0)^2 / 2 + g cos(theta) with a global g, programs for E and dE/dtheta written by hand; periodic in a
    triclinic box with the atoms scattered over several cells.  -> (forces, energy, oracle forces (central differences), oracle energy)"""
    from oracle import valence as OV
    rng = np.random.default_rng(seed)
    box = np.array([2.0, 0.3, 2.2, -0.4, 0.5, 2.4])
    cell = np.array([[box[0], 0, 0], [box[1], box[2], 0], [box[3], box[4], box[5]]])
    base = rng.uniform(0, 1, size=(n, 3)) @ cell
    atoms = np.stack([rng.permutation(n)[:3] for _ in range(n_angles)]).astype(np.int32)
    pos = base + rng.integers(-2, 3, size=(n, 3)) @ cell
    k, t0, g = 50 + 50 * rng.random(n_angles), 1.0 + rng.random(n_angles), 3.5

    def image(d):
        d = d - np.floor(d[:, 2] / cell[2, 2] + 0.5)[:, None] * cell[2]
        d = d - np.floor(d[:, 1] / cell[1, 1] + 0.5)[:, None] * cell[1]
        return d - np.floor(d[:, 0] / cell[0, 0] + 0.5)[:, None] * cell[0]

    def energy(p):
        u, w = image(p[atoms[:, 0]] - p[atoms[:, 1]]), image(p[atoms[:, 2]] - p[atoms[:, 1]])
        theta = np.arccos(np.clip((u * w).sum(1) / np.sqrt((u * u).sum(1) * (w * w).sum(1)), -1, 1))
        return 0.5 * k * (theta - t0) ** 2 + g * np.cos(theta)

    VAR, GLOBAL, ADD, SUB, MUL = 1, 2, 3, 4, 5
    SIN, COS, SQUARE, MULC = 12, 13, 29, 33
    d = [(VAR, 0, 0), (VAR, 7, 0), (SUB, 0, 0)]                          # theta - t0; parameters: 6 = k, 7 = t0
    e_prog = [(VAR, 6, 0)] + d + [(SQUARE, 0, 0), (MUL, 0, 0), (MULC, 0, 0.5), (GLOBAL, 0, 0), (VAR, 0, 0), (COS, 0, 0), (MUL, 0, 0), (ADD, 0, 0)]
    d_prog = [(VAR, 6, 0)] + d + [(MUL, 0, 0), (GLOBAL, 0, 0), (VAR, 0, 0), (SIN, 0, 0), (MUL, 0, 0), (SUB, 0, 0)]
    prog = e_prog + d_prog
    instr = (capi.VmInstruction * len(prog))(*[capi.VmInstruction(op, arg, val) for op, arg, val in prog])
    prog_d = K.malloc(C.sizeof(instr)); K.memcpy_h2d(prog_d, C.cast(instr, C.c_void_p), C.c_size_t(C.sizeof(instr)), None); K.stream_sync(None)
    stride = (n_angles + 2) // 3 * 3
    params = np.zeros((2, stride)); params[0, :n_angles], params[1, :n_angles] = k, t0
    b = capi.VmBonds()
    b.num_bonds, b.num_params, b.param_stride, b.periodic = n_angles, 2, stride, 1
    b.atoms, b.params, b.program = K.upload(atoms.reshape(-1)), K.upload(params.reshape(-1)), prog_d
    b.energy_first, b.energy_count, b.deriv_first, b.deriv_count = 0, len(e_prog), len(e_prog), len(d_prog)
    b.globals = K.upload(np.array([g]))
    for i in range(6): b.box[i] = box[i]
    padded = (n + 31) // 32 * 32 + 32
    perm = rng.permutation(padded)[:n]
    pos4 = np.zeros((n, 4)); pos4[:, :3] = pos
    pos_d, slot_d = K.upload(pos4), K.upload(perm.astype(np.int32))
    force_d = K.upload(np.zeros(3 * padded, dtype=np.int64)); e_d = K.upload(np.zeros(64)); out_d = K.upload(np.zeros(3 * n))
    K.vm_angle_forces(C.byref(b), pos_d, slot_d, padded, force_d, e_d, 64, 1, None)
    K.forces_to_atom_order(force_d, slot_d, n, padded, out_d, None)
    K.stream_sync(None)
    return K.download(out_d, (n, 3), np.float64), K.download(e_d, (64,), np.float64).sum(), OV.forces(energy, pos, h=1e-6), energy(pos).sum()


def _zoo_partition():
    """The constraint zoo (openmm_amd/testsystems.py::constraint_zoo) cut up as the platform cuts it: SETTLE waters, SHAKE clusters
    (a centre whose 1-3 satellites carry no other constraint), everything else to CCMA -- here computed in numpy from the constraint
    graph, independently of both the plugin and the Reference platform."""
    from openmm_amd import testsystems as T
    w = T.constraint_zoo()
    pairs, dist = np.asarray(w.constraints[0], np.int64), np.asarray(w.constraints[1], np.float64)
    n = w.num_atoms
    degree = np.bincount(pairs.reshape(-1), minlength=n)
    first_water = 150
    water = (pairs[:, 0] >= first_water)
    o = np.arange(first_water, n, 3)
    settle = np.stack([o, o + 1, o + 2], 1).astype(np.int32)
    # solute: satellites = atoms of degree 1 whose partner... a cluster is SHAKE-able when the centre's constraints all end in degree-1 atoms
    nbr = [[] for _ in range(n)]
    for (a, b), d in zip(pairs[~water], dist[~water]):
        nbr[a].append((int(b), d)); nbr[b].append((int(a), d))
    shake_atoms, shake_dist, in_shake = [], [], np.zeros(n, bool)
    for c in range(first_water):
        if degree[c] == 0 or in_shake[c]:
            continue
        if all(degree[s] == 1 for s, _ in nbr[c]) and len(nbr[c]) <= 3 and (degree[c] > 1 or w.masses[c] > w.masses[nbr[c][0][0]]):
            sats = nbr[c]
            shake_atoms.append([c] + [s for s, _ in sats] + [-1] * (3 - len(sats)))
            shake_dist.append([d for _, d in sats] + [0.0] * (4 - len(sats)))
            in_shake[[c] + [s for s, _ in sats]] = True
    ccma_rows = [k for k, (a, b) in enumerate(pairs) if not water[k] and not in_shake[a] and not in_shake[b]]
    assert all(in_shake[a] == in_shake[b] for a, b in pairs[~water])
    return w, settle, np.array(shake_atoms, np.int32), np.array(shake_dist), pairs[ccma_rows].astype(np.int32), dist[ccma_rows]


def run_constraints(K, velocities, seed=8, tol=1e-9):
    """ommhip_settle, ommhip_shake and ommhip_ccma_iterations through the C ABI on the constraint zoo against oracle/constraints.py
    (pinned to the Reference platform by tests/test_oracle_constraints.py).  Positions: a Verlet drift of ~300 K velocities from
    constrained positions (distinct before / trial arrays); velocities: random velocities projected.
    -> dict of (got, oracle) per algorithm + the CCMA iteration counts (device, oracle)."""
    from oracle import constraints as OC
    w, settle, shake_atoms, shake_dist, cc, cd = _zoo_partition()
    rng = np.random.default_rng(seed)
    n = w.num_atoms
    inv = 1.0 / w.masses
    angles = [(int(a), int(b), int(c), float(t)) for (a, b, c), t in zip(w.angles[0], w.angles[1])]
    Kmat = OC.ccma_matrix(n, cc, cd, w.masses, angles)
    # constrained start (the oracle's own solution of the builder's coordinates)
    t3 = T3 = None
    d_leg, d_base = w.constraints[1][-3], w.constraints[1][-1]
    pos, _ = OC.ccma(w.positions, w.positions, inv, cc, cd, Kmat, 1e-12)
    pos = OC.shake(pos, pos, inv, shake_atoms, shake_dist, 1e-12)
    pos = OC.settle_positions(pos, pos, w.masses, settle, d_leg, d_base)
    vel = rng.normal(0, 1.6, (n, 3)) * np.sqrt(inv)[:, None]
    target = vel if velocities else pos + 0.002 * vel
    pos4 = np.zeros((n, 4)); pos4[:, :3] = pos
    vm4 = np.zeros((n, 4)); vm4[:, 3] = inv
    t4 = np.zeros((n, 4)); t4[:, :3] = target; t4[:, 3] = inv if velocities else 0.0
    d_pos, d_vm = K.upload(pos4), K.upload(vm4)
    out = {}
    # SETTLE
    sa = np.full((len(settle), 4), -1, np.int32); sa[:, :3] = settle
    sd = np.tile([d_leg, d_base], (len(settle), 1))
    d_t = K.upload(t4)
    K.settle(len(settle), K.upload(sa), K.upload(sd), d_pos, d_t, d_vm, int(velocities), None)
    K.stream_sync(None)
    got = K.download(d_t, (n, 4), np.float64)
    want = OC.settle_velocities(pos, target, w.masses, settle) if velocities else OC.settle_positions(pos, target, w.masses, settle, d_leg, d_base)
    assert np.array_equal(got[:, 3], t4[:, 3])
    out["settle"] = (got[:, :3], want)
    # SHAKE
    d_t = K.upload(t4)
    K.shake(len(shake_atoms), K.upload(shake_atoms), K.upload(shake_dist), d_pos, d_t, d_vm, int(velocities), tol, 150, None)
    K.stream_sync(None)
    out["shake"] = (K.download(d_t, (n, 4), np.float64)[:, :3], OC.shake(pos, target, inv, shake_atoms, shake_dist, tol, velocities=velocities))
    # CCMA, device-resident batches as HipConstraints::runCcma drives them
    rows = [np.nonzero(Kmat[r])[0] for r in range(len(cc))]
    row_start = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int32)
    col = np.concatenate(rows).astype(np.int32)
    val = np.concatenate([Kmat[r, rows[r]] for r in range(len(cc))])
    c = capi.Ccma()
    c.num_constraints = len(cc)
    c.atoms, c.distance = K.upload(cc), K.upload(cd)
    c.delta, c.delta2 = K.upload(np.zeros(len(cc))), K.upload(np.zeros(len(cc)))
    c.row_start, c.col, c.value = K.upload(row_start), K.upload(col), K.upload(val)
    c.converged = K.upload(np.zeros(4, np.int32))
    d_t = K.upload(t4)
    batches = 0
    while batches < 40:
        K.ccma_iterations(C.byref(c), d_pos, d_t, d_vm, int(velocities), tol, 4, None)
        K.stream_sync(None)
        batches += 1
        state = K.download(c.converged, (4,), np.int32)
        if state[2]:
            break
    want, iterations = OC.ccma(pos, target, inv, cc, cd, Kmat, tol, velocities=velocities)
    out["ccma"] = (K.download(d_t, (n, 4), np.float64)[:, :3], want)
    out["ccma_iterations"] = (int(state[3]), iterations, int(state[2]))
    out["scale"] = float(np.abs(target).max())
    return out


def run_ewald_reciprocal(K, n=300, L=2.4, seed=12, kmax=(9, 11, 7)):
    """ommhip_ewald_reciprocal (classic Ewald k-sum, rectangular box) against oracle.nonbonded.ewald_reciprocal (pinned to the Reference
    platform and the Gromacs golden of TestEwald.h).  -> (forces, energy, oracle forces, oracle energy)"""
    rng = np.random.default_rng(seed)
    box3 = np.diag([L, 1.1 * L, 0.9 * L])
    pos = rng.random((n, 3)) @ box3 + rng.integers(-1, 2, (n, 3)) @ box3       # some atoms outside the first cell
    q = rng.normal(0, 0.6, n); q -= q.mean()
    alpha = 3.1
    padded = (n + 31) // 32 * 32 + 32
    perm = rng.permutation(padded)[:n].astype(np.int32)
    pos4 = np.zeros((n, 4)); pos4[:, :3] = pos
    numk = kmax[0] * (2 * kmax[1] - 1) * (2 * kmax[2] - 1)
    d_f, d_e, d_out = K.upload(np.zeros(3 * padded, np.int64)), K.upload(np.zeros(64)), K.upload(np.zeros(3 * n))
    d_slot = K.upload(perm)
    K.ewald_reciprocal(K.upload(pos4), K.upload(q), d_slot, n, padded, box6(box3), alpha, kmax[0], kmax[1], kmax[2], K.upload(np.zeros((numk, 2))),
                       d_f, d_e, 64, 1, None)
    K.forces_to_atom_order(d_f, d_slot, n, padded, d_out, None)
    K.stream_sync(None)
    f_or, e_or = ONB.ewald_reciprocal(pos, q, box3, alpha, kmax)
    return K.download(d_out, (n, 3), np.float64), K.download(d_e, (64,), np.float64).sum(), f_or, e_or
