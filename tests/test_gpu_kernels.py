"""`-m gpu`: kernel-level parity on a real MI355X through the C ABI of include/openmm_hip_kernels.h.
Tolerances (single-precision forces, fixed-point accumulation): force max-rel-err (|dF|_max / RMS|F|) <= 2e-5 per
kernel against the float64 numpy oracle, energies 1e-5 relative; FFT 1e-5 as in TestCudaFFT3D.cpp."""
import numpy as np
import pytest

from conftest import max_rel_force_error
import kernel_cases as KC
from openmm_amd import capi
from oracle import nonbonded as ONB

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def K():
    return capi.load()          # product library; raises if it is missing


EXCL = [(i, i + 1) for i in range(0, 900, 3)] + [(i, i + 2) for i in range(0, 900, 3)]


@pytest.mark.parametrize("n,method,tric,switch", [
    (33, ONB.NoCutoff, False, None), (300, ONB.NoCutoff, False, None), (1500, ONB.CutoffNonPeriodic, False, None),
    (2000, ONB.CutoffPeriodic, False, None), (2000, ONB.PME, False, None), (2000, ONB.PME, True, None),
    (2000, ONB.CutoffPeriodic, False, 0.8), (2000, ONB.Ewald, True, 0.85), (3000, ONB.PME, False, None)])
def test_direct_space_kernel(K, n, method, tric, switch):
    excl = [p for p in EXCL if p[1] < n]
    f, e, f_or, e_or, state = KC.run_direct_space(K, n, method, 1.0, 3.2 if n <= 2000 else 3.7, excl, tric, switch, grid=256)
    assert state[0] == 0 and state[2] == 0 and state[1] > 0
    # jittered-lattice inputs contain a few close pairs with forces ~10x the RMS; single-precision pair arithmetic
    # is ~1e-6 of each pair force, so the bound relative to the RMS is the north-star 1e-4 (typical value: 5e-6..5e-5)
    assert max_rel_force_error(f, f_or) < 1e-4
    assert abs(e - e_or) < 1e-5 * max(abs(e_or), 100.0)


@pytest.mark.parametrize("n,method,L,cutoff,switch", [(3000, ONB.PME, 4.6, 0.9, None), (3000, ONB.CutoffPeriodic, 4.6, 0.9, 0.8), (2500, ONB.Ewald, 4.4, 0.8, None)])
def test_direct_space_single_image_path(K, n, method, L, cutoff, switch):
    """Spatially sorted slots and the per-step entry ommhip_nl_step (image-coherent blocks): the pair kernel searches the
    periodic image once per j atom instead of once per pair; same bar against the oracle as the general path."""
    excl = [p for p in EXCL if p[1] < n]
    f, e, f_or, e_or, state = KC.run_direct_space(K, n, method, cutoff, L, excl, False, switch, grid=256, compact=True)
    assert state[0] == 0 and state[2] == 0 and state[1] > 0
    assert KC.LAST_SINGLE_FRACTION > 0.8
    assert max_rel_force_error(f, f_or) < 1e-4
    assert abs(e - e_or) < 1e-5 * max(abs(e_or), 100.0)


@pytest.mark.parametrize("energy,lj_free_tail,fused", [(False, False, None), (False, True, None), (True, True, None), (False, True, (32, 32, 32))])
def test_direct_space_force_only_and_lj_free_variants(K, energy, lj_free_tail, fused):
    """The loops a production step runs: forces only (polynomial form of the real-space Ewald force, no exp / rcp) and blocks
    whose atoms from slot 12 on have no Lennard-Jones parameters (LJ arithmetic left out) -- same bar against the oracle."""
    f, e, f_or, e_or, state = KC.run_direct_space(K, 3000, ONB.PME, 0.9, 4.6, EXCL, grid=256, compact=True, energy=energy, lj_free_tail=lj_free_tail, fused_pme=fused)
    assert state[2] == 0 and state[1] > 0
    assert KC.LAST_SINGLE_FRACTION > 0.8
    assert max_rel_force_error(f, f_or) < 1e-4
    if energy:
        assert abs(e - e_or) < 1e-5 * max(abs(e_or), 100.0)


@pytest.mark.parametrize("kw", [dict(), dict(energy=True), dict(triclinic=True, compact=False), dict(compact=False)],
                         ids=["single_image_forces", "single_image_energy", "triclinic", "per_pair_image"])
def test_cutoff_edge_pairs_are_decided_in_double(K, kw):
    """~100 pairs planted at rc (1 +- 1e-10 ... 3e-7) in a dense 3 000-atom box.  Float32 separations cannot tell
    which side of the cutoff they are on; the pair kernel's rare double-precision path must put every one where the oracle (and
    ReferenceNeighborList.cpp:195-197) puts it -- a wrong decision is an error of one whole truncation jump on two atoms."""
    f, f_or, planted, jump, en, e_or = KC.run_cutoff_edge(K, n=3000, L=4.6, cutoff=0.9, **kw)
    err = np.linalg.norm(f - f_or, axis=1)
    rms = np.sqrt((f_or ** 2).sum(1).mean())
    assert err.max() < (5e-5 if kw.get("compact", True) else 1e-4) * rms and err[planted].max() < (0.02 if kw.get("compact", True) else 0.1) * jump
    if kw.get("energy"):
        assert abs(en - e_or) < 1e-5 * abs(e_or)


def test_cutoff_edge_case_is_not_vacuous(K):
    f, f_or, planted, jump, en, e_or = KC.run_cutoff_edge(K, n=3000, L=4.6, cutoff=0.9, edge_path=False)
    assert np.linalg.norm(f - f_or, axis=1)[planted].max() > 0.5 * jump


@pytest.mark.parametrize("ewald_tol", [1e-4, 1e-6])
def test_direct_space_force_only_at_other_ewald_tolerances(K, ewald_tol):
    """alpha * cutoff = 2.92 (polynomial form of the real-space Ewald force) and 3.62 (beyond its fit: erfc form) -- same bar."""
    f, e, f_or, e_or, state = KC.run_direct_space(K, 3000, ONB.PME, 0.9, 4.6, EXCL, grid=256, compact=True, energy=False, ewald_tol=ewald_tol)
    assert state[2] == 0 and state[1] > 0
    assert max_rel_force_error(f, f_or) < 1e-4


@pytest.mark.parametrize("n,L,cutoff,compact", [(3000, 4.6, 0.9, True), (3000, 4.6, 0.9, False), (2500, 6.5, 0.8, True)])
def test_direct_space_cell_binned_builder(K, n, L, cutoff, compact):
    """The candidate search used from 65 k atoms up (blocks bucketed by the grid cell of their centre, only nearby cells
    scanned) forced at a size the dense oracle can check, for compact and for scattered blocks and a sparse box."""
    excl = [p for p in EXCL if p[1] < n]
    f, e, f_or, e_or, state = KC.run_direct_space(K, n, ONB.PME, cutoff, L, excl, grid=256, compact=compact, cells=True)
    assert state[0] == 0 and state[2] == 0 and state[1] > 0
    assert max_rel_force_error(f, f_or) < 1e-4
    assert abs(e - e_or) < 1e-5 * max(abs(e_or), 100.0)


@pytest.mark.parametrize("n,L,cutoff,ng,switch", [(3000, 4.6, 0.9, (40, 40, 40), None), (6000, 6.2, 0.9, (56, 56, 56), None),
                                                  (2500, 4.4, 0.8, (36, 40, 32), None), (3000, 4.6, 0.9, (40, 36, 48), 0.8)])
def test_fused_single_stream_evaluation(K, n, L, cutoff, ng, switch):
    """The launch sequence of the single-stream default through the C ABI: ommhip_nl_prepare (+clears of a dirty force
    buffer and charge grid), ommhip_force_front (list build + charge spreading in one launch), ommhip_pairs_with_fft
    (pair kernel riding on the three FFT launches), ommhip_pme_reciprocal(interpolate only) -- against the oracle's
    direct + reciprocal space."""
    f, e, f_or, e_or, state = KC.run_direct_space(K, n, ONB.PME, cutoff, L, EXCL, compact=True, fused_pme=ng, switch=switch)
    assert state[2] == 0 and state[1] > 0
    assert max_rel_force_error(f, f_or) < 1e-4
    assert abs(e - e_or) < 2e-5 * abs(e_or) + 1e-2


@pytest.mark.parametrize("compact", [False, True])
def test_block_range_lists_partition_the_pairs(K, compact):
    """Force decomposition between ranks: lists built for disjoint ranges of i-blocks
    (ommhip_neighbor_list.first_block / owned_blocks) partition the pairs -- the fixed-point force buffers of the parts
    add up to the buffer of the whole evaluation.  Every list is built anew here, and on the GPU the composition of the
    rows depends on the order in which wavefronts append candidates, so the float partial sums inside a chunk differ in
    rounding: agreement is to summation noise.  (With identical row composition -- the deterministic CPU emulator,
    tests/test_multirank_cpu.py with two gloo ranks -- the sum is bit for bit the single-rank buffer.)"""
    n, blocks = 3000, (3000 + 31) // 32
    whole = KC.run_direct_space(K, n, ONB.PME, 0.9, 4.6, EXCL, compact=compact)
    total = KC.LAST_FIXED_POINT_FORCES.copy()
    acc = np.zeros_like(total)
    chunks = 0
    for first, count in ((0, 30), (30, 1), (31, blocks - 31)):
        part = KC.run_direct_space(K, n, ONB.PME, 0.9, 4.6, EXCL, compact=compact, block_range=(first, count))
        acc += KC.LAST_FIXED_POINT_FORCES
        chunks += int(part[4][1])
    scale = np.abs(total).max()
    assert np.abs(acc - total).max() < 1e-5 * scale          # float noise of the per-chunk partial sums (measured ~1e-6)
    assert chunks >= int(whole[4][1]) - 2      # the same pairs, possibly packed into a few more chunks


def test_direct_space_kernel_launch_shape_independent(K):
    """Different launch shapes (and a rebuilt list, whose row composition depends on the order in which wavefronts
    append to it) must agree to float-summation noise; the integer force accumulation itself is order independent."""
    a = KC.run_direct_space(K, 2000, ONB.PME, 1.0, 3.2, EXCL, grid=256)[0]
    b = KC.run_direct_space(K, 2000, ONB.PME, 1.0, 3.2, EXCL, grid=64)[0]
    assert max_rel_force_error(a, b) < 2e-6


@pytest.mark.parametrize("ng", [(8, 6, 10), (28, 25, 30), (25, 28, 25), (21, 20, 18), (56, 56, 56), (64, 60, 72), (98, 98, 70), (96, 96, 96)])
@pytest.mark.parametrize("fft_mode", [0, 1])
def test_fft3d_against_numpy(K, ng, fft_mode):
    # fft_mode 0 uses the fused LDS plane kernel where a plane fits; 1 forces the three separate line passes
    fwd, back = KC.run_fft(K, ng, fft_mode=fft_mode)
    assert fwd < 1e-5 and back < 1e-5


@pytest.mark.parametrize("ng", [(12, 105, 140), (128, 160, 144), (192, 192, 192)])
def test_fft3d_large_planes_against_numpy(K, ng):
    # planes beyond the small plane kernel's LDS: fft_bigplane_kernel (one 1024-thread workgroup per x plane, 156 KB of LDS, in place)
    fwd, back = KC.run_fft(K, ng, fft_mode=2)          # 2: the large plane kernel whatever the number of planes (by default only from 64 planes up)
    assert fwd < 1e-5 and back < 1e-5


@pytest.mark.parametrize("n,ng,tric", [(500, (20, 24, 28), False), (500, (20, 24, 28), True), (3000, (32, 32, 32), False), (6000, (56, 56, 56), False)])
def test_pme_reciprocal(K, n, ng, tric):
    f, e, f_or, e_or = KC.run_pme(K, n, ng, 3.0 if n <= 3000 else 6.2, tric, alpha=2.6 if n <= 3000 else 2.92)
    assert max_rel_force_error(f, f_or) < 2e-5
    assert abs(e - e_or) < 1e-5 * abs(e_or)


@pytest.mark.parametrize("n,cutoff,box,sort_cell", [(21000, 0.5, (4.2, 5.0, 6.1), 0.12), (9000, 0.5, (2.3, 5.0, 8.1), 0.12),
                                                    (9000, 0.7, (3.3, 3.0, 3.1), 0.3), (150000, 0.9, (11.0, 11.5, 12.0), 0.3)])
def test_cell_binned_list_is_complete(K, n, cutoff, box, sort_cell):
    """Every pair within the cutoff (scipy's periodic cKDTree) is in the list exactly once when the candidate blocks come from
    the cell-sorted block list (the search of 0.5M+ atom systems, forced here); same entries as the all-blocks scan."""
    missing, dup, true_pairs, entries, state = KC.run_list_completeness(K, n, cutoff, box, sort_cell, cells=True, seed=n % 7)
    assert missing == 0 and dup == 0 and true_pairs > 100000
    # (`entries` counts the rows the pair kernel walks: re-packed by the same launch to the j atoms within the cutoff itself of the block's box)
    assert entries < KC.LAST_ENTRIES_AS_BUILT
    missing0, dup0, _, entries0, _ = KC.run_list_completeness(K, n, cutoff, box, sort_cell, cells=False, seed=n % 7)
    assert missing0 == 0 and dup0 == 0 and entries0 == entries


@pytest.mark.parametrize("kw", [dict(tiles=(128,)), dict(tiles=(4,)), dict(tiles=(128,), shift=1.0)])
def test_pme_spreading_by_grid_tiles(K, kw):
    """spread_mode 2 (one workgroup per 16^3 grid tile, written once, no global atomics) against the float64 oracle: 20 000 atoms
    on a 64 x 60 x 72 grid, lists that hold everything, lists of 4 entries (scan-everything path), atoms one box length away."""
    f, e, f_or, e_or = KC.run_pme(K, 20000, (64, 60, 72), 6.4, sort_cell=0.4, **kw)
    assert np.abs(f - f_or).max() / np.sqrt((f_or ** 2).sum(1).mean()) < 2e-5
    assert abs(e - e_or) < 5e-6 * abs(e_or)


def test_permlane_swap_transpose_reduce_against_plain_sums(K):
    """The force reduction the pair kernel runs on gfx950 -- v_permlane32_swap / v_permlane16_swap halving two partial sums per instruction
    (kernels/nonbonded.hip: swap_add32, swap_add16, transpose_reduce32) -- against float64 column sums.  The CPU emulator compiles a shuffle
    twin of the two swap helpers (tests/test_emu_host_logic.py runs the same case on it), so only this test pins the instructions themselves."""
    got, expect = KC.run_transpose_reduce(K)
    assert np.allclose(got, expect, rtol=2e-6, atol=2e-5), np.abs(got - expect).max()
    assert np.array_equal(got[0], expect[0].astype(np.float32))           # integers below 2^24: exact whatever the order of the additions


@pytest.mark.parametrize("kind", sorted(KC.VALENCE_KINDS))
def test_valence_kernels_against_numpy_energies(K, kind):
    """ommhip_valence_forces (kernels/valence.hip) through the C ABI: every AMOEBA valence term kind against the numpy restatement of its
    energy (oracle/valence.py) and central differences of it."""
    f, e, f_or, e_or = KC.run_valence(K, kind, n_terms=500)
    scale = np.abs(f_or).max()
    assert abs(e - e_or) < 1e-9 * max(1.0, abs(e_or)), (e, e_or)
    assert np.abs(f - f_or).max() < 2e-6 * scale, (np.abs(f - f_or).max(), scale)


def test_custom_integrator_interpreter_through_the_c_abi(K):
    """ommhip_vm_per_dof with hand-written postfix programs: three computations in one launch (a per-DOF variable, v, x -- each reading what
    the one before wrote), a sum over the degrees of freedom, a massless particle left alone -- against numpy."""
    for name, (got, expected) in KC.run_vm(K).items():
        assert np.allclose(got, expected, rtol=1e-13, atol=1e-13), name


def test_interpreted_custom_bond_force_through_the_c_abi(K):
    """ommhip_vm_bond_forces: a Morse bond with a global parameter, programs for E and dE/dr written by hand, periodic in a triclinic box with
    the atoms scattered over several cells -- energy and forces against numpy (forces: central differences of the numpy energy)."""
    f, e, f_or, e_or = KC.run_vm_bonds(K)
    assert abs(e - e_or) < 1e-11 * abs(e_or), (e, e_or)
    assert np.abs(f - f_or).max() < 1e-6 * np.abs(f_or).max(), (np.abs(f - f_or).max(), np.abs(f_or).max())
    # ommhip_vm_angle_forces: k (theta - t0)^2 / 2 + g cos(theta), the same box
    f, e, f_or, e_or = KC.run_vm_angles(K)
    assert abs(e - e_or) < 1e-11 * abs(e_or), (e, e_or)
    assert np.abs(f - f_or).max() < 1e-6 * np.abs(f_or).max(), (np.abs(f - f_or).max(), np.abs(f_or).max())



@pytest.mark.parametrize("velocities", [False, True])
def test_settle_shake_and_ccma_through_the_c_abi(K, velocities):
    """ommhip_settle / ommhip_shake / ommhip_ccma_iterations against oracle/constraints.py (itself pinned to
    ReferenceSETTLEAlgorithm / ReferenceCCMAAlgorithm through the Reference platform, tests/test_oracle_constraints.py) on the constraint
    zoo, positions (distinct before / trial arrays, as inside a step) and velocities.  SETTLE is analytic: 1e-12 of the coordinates;
    the iterative ones stop inside the same tolerance band as the oracle, CCMA after the same number of iterations."""
    out = KC.run_constraints(K, velocities)
    scale = out["scale"]
    got, want = out["settle"]
    assert np.abs(got - want).max() < 1e-12 * scale
    got, want = out["shake"]
    assert np.abs(got - want).max() < 1e-12 * scale         # same Gauss-Seidel order, same stopping rule: the same numbers
    got, want = out["ccma"]
    device_iterations, oracle_iterations, converged = out["ccma_iterations"]
    assert converged == 1, "the device never announced convergence"
    assert np.abs(got - want).max() < 1e-11 * scale          # float atomics add the corrections of one atom in another order
    # the device counts every delta kernel it ran up to and including the one that found everything converged
    assert device_iterations == oracle_iterations + 1, (device_iterations, oracle_iterations)


def test_ewald_reciprocal_sum_through_the_c_abi(K):
    """ommhip_ewald_reciprocal against the numpy k-sum (pinned to the Reference platform and TestEwald.h's Gromacs
    golden): forces to 1e-9 of the largest (fixed-point quantum 2^-32), energy 1e-12."""
    f, e, f_or, e_or = KC.run_ewald_reciprocal(K)
    assert abs(e - e_or) < 1e-11 * abs(e_or), (e, e_or)
    assert np.abs(f - f_or).max() < 1e-9 * np.abs(f_or).max(), (np.abs(f - f_or).max(), np.abs(f_or).max())
