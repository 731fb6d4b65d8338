"""Host-logic checks without a GPU: the same kernel sources and plugin, compiled with g++ against the SIMT
emulator of tests/emu (test infrastructure only), run (a) kernel-level cases through the C ABI, (b) the
reference's own test bodies (tests/Test*.h of the OpenMM tree, built by tests/hip/Makefile) and (c) a
Context("HIP") round trip through OpenMM's plugin loader.  Numerics on real hardware are covered by `-m gpu`."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import EMU_BUILD, ROOT, max_rel_force_error
import kernel_cases as KC
from openmm_amd import capi
from oracle import nonbonded as ONB

EMU_LIB = os.path.join(EMU_BUILD, "libopenmm_hip_kernels.so")
needs_emu = pytest.mark.skipif(not os.path.exists(EMU_LIB), reason="emulated build missing (run __graft_entry__.build())")


@pytest.fixture(scope="module")
def K():
    return capi.load(EMU_LIB)


EXCL = [(i, i + 1) for i in range(0, 200, 3)] + [(i, i + 2) for i in range(0, 200, 3)]


@needs_emu
@pytest.mark.parametrize("n,method,tric,switch", [
    (300, ONB.NoCutoff, False, None), (500, ONB.CutoffNonPeriodic, False, None), (700, ONB.CutoffPeriodic, False, None),
    (700, ONB.PME, False, None), (700, ONB.PME, True, None), (700, ONB.CutoffPeriodic, False, 0.8)])
def test_direct_space_kernel_logic(K, n, method, tric, switch):
    f, e, f_or, e_or, state = KC.run_direct_space(K, n, method, 1.0, 3.0, EXCL, tric, switch)
    assert state[0] == 0 and state[2] == 0 and state[1] > 0
    assert max_rel_force_error(f, f_or) < 5e-5
    assert abs(e - e_or) < 5e-5 * max(abs(e_or), 100.0)


@needs_emu
def test_transpose_reduce_of_the_pair_kernel_on_the_emulator(K):
    """transpose_reduce32 with the emulator's shuffle form of swap_add32 / swap_add16 (the GPU suite pins the permlane-swap instructions)."""
    got, expect = KC.run_transpose_reduce(K)
    assert np.allclose(got, expect, rtol=2e-6, atol=2e-5)
    assert np.array_equal(got[0], expect[0].astype(np.float32))


@needs_emu
def test_direct_space_single_image_path(K):
    """Morton-sorted slots + the per-step entry (image-coherent blocks): most chunks take the single-image path."""
    f, e, f_or, e_or, state = KC.run_direct_space(K, 1200, ONB.PME, 0.7, 3.4, EXCL, compact=True)
    assert state[0] == 0 and state[2] == 0 and state[1] > 0
    assert KC.LAST_SINGLE_FRACTION > 0.5
    assert max_rel_force_error(f, f_or) < 5e-5
    assert abs(e - e_or) < 5e-5 * max(abs(e_or), 100.0)


@needs_emu
@pytest.mark.parametrize("energy,lj_free_tail,fused", [(False, False, None), (False, True, None), (True, True, None), (False, True, (24, 24, 24))])
def test_direct_space_force_only_and_lj_free_variants(K, energy, lj_free_tail, fused):
    """The loops a production step runs: forces only (polynomial form of the real-space Ewald force, no exp / rcp) and blocks
    whose atoms from slot 12 on have no Lennard-Jones parameters (LJ arithmetic left out) -- same bar against the oracle."""
    f, e, f_or, e_or, state = KC.run_direct_space(K, 1200, ONB.PME, 0.7, 3.4, EXCL, compact=True, energy=energy, lj_free_tail=lj_free_tail, fused_pme=fused)
    assert state[2] == 0 and state[1] > 0
    assert KC.LAST_SINGLE_FRACTION > 0.5
    assert max_rel_force_error(f, f_or) < (1e-4 if fused else 5e-5)
    if energy:
        assert abs(e - e_or) < 5e-5 * max(abs(e_or), 100.0)


@needs_emu
@pytest.mark.parametrize("ewald_tol", [1e-4, 1e-6])
def test_direct_space_force_only_at_other_ewald_tolerances(K, ewald_tol):
    """alpha * cutoff = 2.92 (the degree-11 fit of the real-space Ewald force still holds: polynomial form) and 3.62 (it does not:
    the kernel must fall back to the erfc form) -- same accuracy against the oracle either way."""
    f, e, f_or, e_or, state = KC.run_direct_space(K, 1200, ONB.PME, 0.7, 3.4, EXCL, compact=True, energy=False, ewald_tol=ewald_tol)
    assert state[2] == 0 and state[1] > 0
    assert max_rel_force_error(f, f_or) < 5e-5


@needs_emu
@pytest.mark.parametrize("switch,ng", [(None, (24, 24, 24)), (0.6, (24, 20, 28))])
def test_fused_single_stream_evaluation(K, switch, ng):
    """nl_prepare (+clears) -> force_front (list build + charge spreading) -> pairs_with_fft -> interpolate, through the C ABI"""
    f, e, f_or, e_or, state = KC.run_direct_space(K, 1200, ONB.PME, 0.7, 3.4, EXCL, compact=True, fused_pme=ng, switch=switch)
    assert state[2] == 0 and state[1] > 0
    assert max_rel_force_error(f, f_or) < 1e-4
    assert abs(e - e_or) < 5e-5 * max(abs(e_or), 100.0)


@needs_emu
@pytest.mark.parametrize("kw", [dict(), dict(energy=True), dict(triclinic=True, compact=False), dict(compact=False)],
                         ids=["single_image_forces", "single_image_energy", "triclinic", "per_pair_image"])
def test_cutoff_edge_pairs_are_decided_in_double(K, kw):
    # a pair within float rounding of the cutoff must land on the side the Reference platform puts it
    f, f_or, planted, jump, en, e_or = KC.run_cutoff_edge(K, **kw)
    if not kw or kw.get("energy"):
        assert KC.LAST_SINGLE_FRACTION > 0.8
    err = np.linalg.norm(f - f_or, axis=1)
    rms = np.sqrt((f_or ** 2).sum(1).mean())
    # (a wrong decision is an error of one whole jump; the float noise of a random slot order, whose blocks span the box, is 3 % of it)
    assert err.max() < (5e-5 if kw.get("compact", True) else 1e-4) * rms and err[planted].max() < (0.02 if kw.get("compact", True) else 0.1) * jump, "a pair was counted on the wrong side of the cutoff: %g of the force jump there" % (err[planted].max() / jump)
    if kw.get("energy"):
        assert abs(en - e_or) < 1e-5 * abs(e_or)


def test_cutoff_edge_case_is_not_vacuous(K):
    # the same pairs without the low parts of the coordinates: the float separation decides, and some pairs land on the wrong side
    f, f_or, planted, jump, en, e_or = KC.run_cutoff_edge(K, edge_path=False)
    assert np.linalg.norm(f - f_or, axis=1)[planted].max() > 0.5 * jump


def test_pairs_with_fft_declines_configurations_it_does_not_cover(K):
    """ommhip_pairs_with_fft returns -1 and launches nothing for triclinic boxes, non-Ewald methods and planes beyond its
    LDS budget; the caller then uses the separate entry points."""
    import ctypes as C
    from openmm_amd import capi
    fn = K.lib.ommhip_pairs_with_fft
    fn.restype = C.c_int
    nl, p, pm = capi.NeighborList(), capi.NonbondedParams(), capi.Pme()
    nl.pbc, p.ewald = 1, 1
    pm.nx, pm.ny, pm.nz = 96, 96, 96                      # plane 96 x 97 > 4160 elements
    assert fn(C.byref(nl), C.byref(p), None, C.byref(pm), None, None, 1, 0, None) == -1
    pm.nx, pm.ny, pm.nz = 32, 32, 32
    nl.pbc = 2                                            # triclinic
    assert fn(C.byref(nl), C.byref(p), None, C.byref(pm), None, None, 1, 0, None) == -1
    nl.pbc, p.ewald = 1, 0                                # cutoff without Ewald: no reciprocal space to ride on
    assert fn(C.byref(nl), C.byref(p), None, C.byref(pm), None, None, 1, 0, None) == -1


@needs_emu
@pytest.mark.parametrize("compact", [False, True])
def test_direct_space_cell_binned_builder(K, compact):
    """The candidate search of large systems (blocks bucketed by grid cell) forced at test size: same list, same forces."""
    f, e, f_or, e_or, state = KC.run_direct_space(K, 1200, ONB.PME, 0.7, 3.4, EXCL, compact=compact, cells=True)
    assert state[0] == 0 and state[2] == 0 and state[1] > 0
    assert max_rel_force_error(f, f_or) < 5e-5
    assert abs(e - e_or) < 5e-5 * max(abs(e_or), 100.0)


@needs_emu
@pytest.mark.parametrize("n,cutoff,box,sort_cell", [(21000, 0.5, (4.2, 5.0, 6.1), 0.12),      # 7 x 9 x 11 cells: sub-period spans on every axis, wrapped columns
                                                    (9000, 0.5, (2.3, 5.0, 8.1), 0.12),       # whole period along x, sub-period along z
                                                    (9000, 0.7, (3.3, 3.0, 3.1), 0.3)])       # fat blocks: most go to the oversized list
def test_cell_binned_list_is_complete(K, n, cutoff, box, sort_cell):
    """Every pair within the cutoff (scipy's periodic cKDTree) is in the list exactly once, with the candidate search through
    the cell-sorted block list -- columns of cells cut to what the x/y gap leaves of the list cutoff -- and the list has the
    same number of entries as the one built by scanning all blocks."""
    missing, dup, true_pairs, entries, state = KC.run_list_completeness(K, n, cutoff, box, sort_cell, cells=True, seed=n % 7)
    assert missing == 0 and dup == 0 and true_pairs > 100000
    # (`entries` counts the rows the pair kernel walks: re-packed by the same launch to the j atoms within the cutoff itself of the block's box)
    assert entries < KC.LAST_ENTRIES_AS_BUILT
    missing0, dup0, _, entries0, _ = KC.run_list_completeness(K, n, cutoff, box, sort_cell, cells=False, seed=n % 7)
    assert missing0 == 0 and dup0 == 0 and entries0 == entries


@needs_emu
@pytest.mark.parametrize("ng", [(8, 6, 10), (28, 25, 30), (21, 20, 18)])
@pytest.mark.parametrize("fft_mode", [0, 1])
def test_fft_logic(K, ng, fft_mode):
    fwd, back = KC.run_fft(K, ng, fft_mode=fft_mode)
    assert fwd < 1e-5 and back < 1e-5


@needs_emu
@pytest.mark.parametrize("ng", [(6, 100, 96), (6, 105, 140), (6, 192, 192)])
def test_fft_large_plane_kernel(K, ng):
    """Planes beyond the two-buffer plane kernel's LDS (fft_bigplane_kernel: one 1024-thread workgroup per x plane, in-place passes,
    the real z transform as a half-length complex one + split / recombination): radices 8 4 3 | 5 7 | 8 8 3, self-paired and
    odd half lengths, against numpy's rfftn and a round trip."""
    assert ng[2] * (ng[1] + 1) > 9472          # beyond PLANE_MAX: the small plane kernel does not take these
    fwd, back = KC.run_fft(K, ng, fft_mode=2)          # 2: the large plane kernel whatever the number of planes (by default only from 64 planes up)
    assert fwd < 1e-5 and back < 1e-5


@needs_emu
@pytest.mark.parametrize("kw", [dict(tiles=(64,)), dict(tiles=(2,)), dict(tiles=(64,), shift=-1.0), dict(tiles=(64,), sort_cell=None, n=1000, ng=(32, 32, 32), L=3.0)])
def test_pme_spreading_by_grid_tiles(K, kw):
    """spread_mode 2: one workgroup per 16^3 grid tile gathers the stencil points of the atoms of the blocks that reach it
    (per-tile block lists from the blocks' bounding boxes) and writes the tile once -- from a dirty grid, with lists that hold
    everything, with lists of 2 entries (the scan-everything path), with all atoms one box length away from the primary cell, and
    with unsorted atoms (every block reaches every tile).  Forces / energy against the float64 oracle."""
    args = dict(n=3000, ng=(40, 36, 50), L=4.0, sort_cell=0.4)
    args.update(kw)
    f, e, f_or, e_or = KC.run_pme(K, args.pop("n"), args.pop("ng"), args.pop("L"), **args)
    assert np.abs(f - f_or).max() / np.sqrt((f_or ** 2).sum(1).mean()) < 2e-5
    assert abs(e - e_or) < 5e-6 * abs(e_or)


@needs_emu
def test_tile_spreading_through_the_platform():
    """The platform's own wiring of the tile spreading (block boxes from the neighbour list, no pre-cleared grid), forced at test
    size: same forces as the brick kernel to float32 noise."""
    code = (
        "import os, sys, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "from openmm_amd import harness as H, testsystems as T\n"
        "H.load_hip_platform(emulated=True)\n"
        "w = T.water_box(8, seed=5)\n"
        "w.pme_params = (float(np.sqrt(-np.log(2 * w.ewald_tol)) / w.cutoff), 32, 32, 36)\n"
        "def forces(env):\n"
        "    os.environ.update(env)\n"
        "    s, nb = w.build()\n"
        "    c = H.Context(s, H.Integrator(H.VERLET, 0.001), 'HIP')\n"
        "    c.setPositions(w.positions)\n"
        "    st = c.getState(getForces=True, getEnergy=True)\n"
        "    c.close()\n"
        "    for k in env: os.environ.pop(k)\n"
        "    return st\n"
        "a, b = forces({}), forces({'OPENMM_HIP_TILE_SPREAD_MIN_ATOMS': '1'})\n"
        "rms = np.sqrt((a.forces ** 2).sum(1).mean())\n"
        "d = np.abs(a.forces - b.forces).max() / rms\n"
        "assert 0 < d < 1e-5, d          # not zero: the two kernels round differently\n"
        "assert abs(a.potentialEnergy - b.potentialEnergy) < 0.01\n"
        "print('OK')\n" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert "OK" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


@needs_emu
def test_fft_line_passes_with_several_tiles_per_workgroup():
    """Large grids run the line-pass kernel with fewer workgroups than tiles: each workgroup walks through its tiles and requests
    the next one while it transforms the current one.  Forced at test size (3 workgroups per launch) in a process of its own,
    because the launch geometry is read once."""
    code = (
        "import os, sys\n"
        "sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, 'tests'))\n"
        "import kernel_cases as KC\n"
        "from openmm_amd import capi\n"
        "K = capi.load(%r)\n"
        "for ng in ((28, 25, 30), (21, 20, 18), (8, 6, 10)):\n"
        "    fwd, back = KC.run_fft(K, ng, fft_mode=1)\n"
        "    assert fwd < 1e-5 and back < 1e-5, (ng, fwd, back)\n"
        "f, e, f_or, e_or = KC.run_pme(K, 300, (20, 24, 28), 3.0, False)\n"
        "import numpy as np\n"
        "assert np.abs(f - f_or).max() / np.sqrt((f_or ** 2).sum(1).mean()) < 5e-5 and abs(e - e_or) < 1e-5 * abs(e_or) + 1e-4, (e, e_or)\n"
        "print('OK')\n" % (ROOT, ROOT, EMU_LIB))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, OMMHIP_FFT_RESIDENT_WORKGROUPS="3"))
    assert "OK" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


@needs_emu
@pytest.mark.parametrize("tric", [False, True])
def test_pme_logic(K, tric):
    f, e, f_or, e_or = KC.run_pme(K, 300, (20, 24, 28), 3.0, tric)
    assert max_rel_force_error(f, f_or) < 5e-5
    assert abs(e - e_or) < 1e-5 * abs(e_or)


FAST_REFERENCE_TESTS = ["HarmonicBondForce", "HarmonicAngleForce", "PeriodicTorsionForce", "CMMotionRemover", "Checkpoints",
                        "CustomBondForce", "RBTorsionForce", "Settle", "NonbondedForce", "CustomExternalForce", "VirtualSites",
                        "AmoebaVdwForce", "AmoebaMultipoleForce", "AmoebaTorsionTorsionForce", "AmoebaExtrapolatedPolarization",
                        "CustomAngleForce", "CustomCompoundBondForce",
                        # tests/hip/TestHipPmeKernel.cpp: CalcPmeReciprocalForceKernel + ::IO, the HIP twin of plugins/cpupme/tests/TestCpuPme.cpp's testPME
                        "PmeKernel",
                        # tests/hip/TestHipParallel.cpp: ONE Context over a device list ("d,d") -- the HIP twin of testParallelComputation
                        # (platforms/cuda/tests/TestCudaNonbondedForce.cpp:37-96); the dynamics part of it runs on the GPU only
                        "Parallel"]
EMU_TEST_ARGS = {"Parallel": ["quick"]}


@needs_emu
@pytest.mark.parametrize("name", FAST_REFERENCE_TESTS)
def test_reference_test_bodies_on_emulated_platform(name):
    exe = os.path.join(EMU_BUILD, "tests", "TestHip" + name)
    if not os.path.exists(exe):
        pytest.skip("not built")
    out = subprocess.run([exe] + EMU_TEST_ARGS.get(name, []), capture_output=True, text=True, timeout=900)
    # Bodies that draw their seed from the clock check statistics with ASSERT_USUALLY_*: the reference's own message says such a
    # failure "may occasionally" happen (openmmapi/include/openmm/internal/AssertionUtilities.h:59-61), so those -- and only those -- get two more draws.
    for attempt in range(2):
        if out.returncode == 0 or "This test is stochastic and may occasionally fail" not in out.stdout:
            break
        out = subprocess.run([exe], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "Done" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
    if name in NATIVE_AMOEBA:
        import re
        m = re.search(r"native AMOEBA kernel evaluations: vdw (\d+) multipole (\d+)", out.stdout)
        assert m is not None and int(m.group(1 if NATIVE_AMOEBA[name] == "vdw" else 2)) > 0, out.stdout[-500:]


NATIVE_AMOEBA = {"AmoebaVdwForce": "vdw", "AmoebaMultipoleForce": "multipole", "AmoebaExtrapolatedPolarization": "multipole"}         # bodies whose forces must have gone through the native kernels of libOpenMMAmoebaHIP.so


@needs_emu
def test_native_amoeba_multipole_kernel_matches_the_plugins_reference_kernel():
    """tests/hip/AmoebaParity.cpp: PME + direct polarization on the systems of the reference's own test body (4 waters, 2 ions + 2 waters,
    216 waters: local frames, covalent scale factors, Thole damping, torques) -- the native kernel against the AMOEBA plugin's Reference
    kernel run on the same HIP Context as a fallback force; the program fails above 1e-4 (measured: 1e-7 ... 3e-6)."""
    exe = os.path.join(EMU_BUILD, "tests", "AmoebaParity")
    if not os.path.exists(exe):
        pytest.skip("not built")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "Done" in out.stdout and "multipole 9" in out.stdout, out.stdout[-2000:] + out.stderr[-1000:]


@needs_emu
@pytest.mark.parametrize("kind", sorted(KC.VALENCE_KINDS))
def test_valence_kernels_against_numpy_energies(K, kind):
    """ommhip_valence_forces through the C ABI on the emulator: every AMOEBA valence term kind against the numpy restatement of its energy
    (oracle/valence.py) and central differences of it -- the kernels differentiate the same expressions with dual numbers."""
    f, e, f_or, e_or = KC.run_valence(K, kind)
    scale = np.abs(f_or).max()
    assert abs(e - e_or) < 1e-9 * max(1.0, abs(e_or)), (e, e_or)
    assert np.abs(f - f_or).max() < 2e-6 * scale, (np.abs(f - f_or).max(), scale)


@needs_emu
def test_custom_forces_native_when_recognised_reference_kernel_otherwise():
    """Four CustomBondForces and two CustomAngleForces on one System: the AMOEBA bond expression with its per-bond parameters declared in the
    other order (hand-written kernel: its parameters are found by name), the same expression written differently (a*b instead of b*a: not
    recognised -> interpreted on the device from the Lepton tree and its symbolic derivative), a Morse bond (interpreted), an expression
    nested deeper than the interpreter's stack (the Reference kernel inside the same kernel object, a fallback force), the AMOEBA angle
    expression (hand-written kernel), another function of theta (interpreted) -- forces and energy against the Reference platform, and the counters say which ran where.  (Global
    parameters, periodic bonds and parameter updates of interpreted forces: the reference's TestCustomBondForce body, tests/hip.)"""
    code = r'''
import sys, numpy as np
sys.path.insert(0, %r)
from openmm_amd import harness as H
H.load_hip_platform(emulated=True)
rng = np.random.default_rng(3)
n = 60
pos = rng.uniform(0, 2.0, size=(n, 3))
bonds = np.stack([np.arange(0, n - 1), np.arange(1, n)], -1)
angles = np.stack([np.arange(0, n - 2), np.arange(1, n - 1), np.arange(2, n)], -1)
r0 = 0.3 + 0.5 * rng.random(len(bonds)); k = 100 * (1 + rng.random(len(bonds))); theta0 = 100 + 20 * rng.random(len(angles))
res = {}
for plat in ("Reference", "HIP"):
    s = H.System(); s.addParticles(np.full(n, 12.0))
    s.addCustomBondForce("k*(d^2 + -25.5*d^3 + 379.3125*d^4); d=r-r0", ["k", "r0"], bonds, np.stack([k, r0], -1))                 # native, parameters swapped
    s.addCustomBondForce("(d^2 + -25.5*d^3 + 379.3125*d^4)*k; d=r-r0", ["r0", "k"], bonds, np.stack([r0, 0.5 * k], -1))           # another shape: Reference
    s.addCustomBondForce("D*(1-exp(-a*(r-r0)))^2", ["D", "a", "r0"], bonds, np.stack([k, np.full(len(bonds), 2.0), r0], -1))        # interpreted
    deep = "r0*r" + "".join("+(r*%%d" %% (i + 2) for i in range(18)) + ")" * 18
    s.addCustomBondForce(deep, ["r0"], bonds[:7], r0[:7, None])                                                                   # too deep for the stack: Reference
    s.addCustomAngleForce("0.5*k*(theta-t0)^2 + k*cos(2*theta)/(1+theta)", ["t0", "k"], angles, np.stack([np.radians(theta0), 30 * np.ones(len(angles))], -1))   # interpreted
    s.addCustomAngleForce("k*(d^2 + -0.014*d^3 + 5.6e-05*d^4 + -7e-07*d^5 + 2.2e-08*d^6); d=57.29577951308232*theta-theta0", ["theta0", "k"], angles,
                          np.stack([theta0, 0.05 * np.ones(len(angles))], -1))                            # native
    ctx = H.Context(s, H.Integrator(H.VERLET, 0.001), plat)
    ctx.setPositions(pos)
    before = (H.valence_lists_launched(), H.interpreted_bond_launches()) if plat == "HIP" else 0
    st = ctx.getState(getForces=True, getEnergy=True)
    res[plat] = (st.forces, st.potentialEnergy)
    if plat == "HIP": print("LISTS", H.valence_lists_launched() - before[0], "INTERPRETED", H.interpreted_bond_launches() - before[1], "MODE", ctx.getPlatformProperty("IntegrationMode"))
    ctx.close()
print("DF", np.abs(res["Reference"][0] - res["HIP"][0]).max() / np.abs(res["Reference"][0]).max(), "DE", abs(res["Reference"][1] - res["HIP"][1]) / abs(res["Reference"][1]))
''' % ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "LISTS 2 INTERPRETED 3 MODE device" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
    import re
    df, de = (float(v) for v in re.search(r"DF (\S+) DE (\S+)", out.stdout).groups())
    assert df < 1e-9 and de < 1e-12, (df, de)


@needs_emu
def test_device_interpreter_evaluates_every_lepton_operation_like_the_reference():
    """One CustomIntegrator step whose per-DOF expressions use every operation the interpreter of kernels/custom_integrator.hip knows (all of
    Lepton's: arithmetic, powers, the transcendental functions, step / delta / select / min / max / abs / floor / ceil, constants folded and
    not), global variables, a ComputeSum, a ComputeGlobal that feeds a later per-DOF step -- on the emulated HIP platform
    (device mode) against the Reference platform: per-DOF variables to 1e-12 relative."""
    code = r'''
import sys, numpy as np
sys.path.insert(0, %r)
from openmm_amd import harness as H, testsystems as T
H.load_hip_platform(emulated=True)
w = T.water_box(3, seed=4, rigid=False, method=H.NoCutoff)
exprs = ["x + v*0.5 - 2/m", "x*v", "v/(abs(x)+1)", "(abs(x)+0.1)^1.7", "(abs(v)+0.2)^(abs(x)+0.3)", "-x", "sqrt(abs(x))", "exp(-x*x)", "log(abs(x)+1)",
         "sin(x)+cos(v)", "sec(x/10)+csc(x/10+1)", "tan(x/10)+cot(x/10+1)", "asin(sin(x))+acos(cos(v))+atan(x)", "atan2(x, v+0.1)", "sinh(x/4)+cosh(v/4)+tanh(x)",
         "erf(x)+erfc(v)", "step(x-1)+delta(step(v))", "x^2+v^3+1/(abs(x)+1)", "3+x", "3*x", "x^3.5*0+abs(x)^3.5", "min(x,v)+max(x,v)", "floor(4*x)+ceil(4*v)",
         "select(step(x-1), x, v)", "a*x+b*f/m+dt", "s*x + g"]
res = {}
for plat in ("Reference", "HIP"):
    s, nb = w.build()
    integ = H.CustomIntegrator(0.001, seed=3)
    integ.addGlobalVariable("a", 0.3); integ.addGlobalVariable("b", -1.25); integ.addGlobalVariable("s", 0.0); integ.addGlobalVariable("g", 0.0)
    for k in range(len(exprs)): integ.addPerDofVariable("r%%d" %% k, 0)
    integ.addComputeSum("s", "m*v*v/2")
    integ.addComputeGlobal("g", "sqrt(s)+a")
    for k, e in enumerate(exprs): integ.addComputePerDof("r%%d" %% k, e)
    ctx = H.Context(s, integ, plat)
    ctx.setPositions(w.positions)
    ctx.setVelocitiesToTemperature(300.0, 7)
    integ.step(1)
    res[plat] = [integ.getPerDofVariable(k, w.num_atoms) for k in range(len(exprs))] + [np.array([[integ.getGlobalVariable(2), integ.getGlobalVariable(3), 0.0]])]
    if plat == "HIP": mode = ctx.getPlatformProperty("IntegrationMode")
    ctx.close()
worst = 0.0
for k, (r, h) in enumerate(zip(res["Reference"], res["HIP"])):
    err = np.abs(r - h).max() / max(1.0, np.abs(r).max())
    name = (exprs + ["globals"])[k]
    # (the expression with f carries the single-precision pair arithmetic of this platform's forces)
    if not np.isfinite(r).all() or err > (1e-5 if "f" in name.replace("floor", "") else 1e-11): print("MISMATCH", k, name, err)
    worst = max(worst, err)
print("MODE", mode)
print("WORST", worst)
''' % ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "MISMATCH" not in out.stdout and "MODE device, custom integrator" in out.stdout, out.stdout[-3000:] + out.stderr[-3000:]


@needs_emu
def test_custom_integrator_with_constraints_pme_and_a_list_overflow(tmp_path):
    """Velocity Verlet written as a CustomIntegrator on a rigid TIP3P box with PME: device interpreter against the Reference platform (SETTLE
    through ConstrainPositions / ConstrainVelocities with the positions of the last constraint as reference, a ComputeSum global), and the
    same run with a neighbour-list overflow in the middle against the undisturbed one (tests/custom_integrator_case.py)."""
    from custom_integrator_case import run_custom_integrator_case
    r = run_custom_integrator_case(tmp_path, True)
    print(r)
    assert r["mode"] == "device, custom integrator"
    assert r["dpos"] < 5e-6 and r["dvel"] < 5e-4 and r["ke_rel"] < 1e-5 and r["ke_state_rel"] < 1e-5
    assert r["constraints"] < 1e-6
    assert r["overflows"] == 1 and r["times"][0] == r["times"][1]
    assert r["overflow_dpos"] < 1e-7 and r["overflow_dvel"] < 1e-5          # (the grid sums of two runs differ in their float rounding)


@needs_emu
def test_reference_custom_integrator_body_on_the_device_interpreter():
    """tests/TestCustomIntegrator.h of the reference on the emulated HIP platform -- a CustomIntegrator whose expressions all have a device
    form runs natively (HipCustomIntegrator.h: per-DOF computations as interpreted programs on the device, global computations and control
    flow on the host), the others (tabulated functions, vector functions, deriv()) in host mode.  The long-running tests of the body
    (thermostat statistics, RESPA energy conservation, ...: 30 minutes on the emulator) run on the GPU only (TestHipCustomIntegrator)."""
    exe = os.path.join(EMU_BUILD, "tests", "TestHipCustomIntegratorParts")
    if not os.path.exists(exe):
        pytest.skip("not built")
    picked = ["testSingleBond", "testConstraints", "testConstrainedMasslessParticles", "testSum", "testParameter", "testRandomDistributions", "testPerDofVariables", "testForceGroups",
              "testIfBlock", "testWhileBlock", "testChangingGlobal", "testEnergyParameterDerivatives", "testChangeDT", "testTabulatedFunction", "testAlternatingGroups",
              "testUpdateContextState", "testVectorFunctions", "testRecordEnergy", "testInitialTemperature", "testCheckpoint", "testSaveParameters"]
    out = subprocess.run([exe] + picked, capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0 and "%d tests, 0 failures" % len(picked) in out.stdout, out.stdout[-3000:] + out.stderr[-2000:]


@needs_emu
def test_amoeba_water_box_tile_scan_against_reference_kernel_and_full_scan(tmp_path):
    """1 536-atom AMOEBA water box (12 tiles of 128 slots in a 2.5 nm box, some pairs of tiles beyond the 0.7 nm cutoff), direct
    polarization: the tile-skipping pair scan of the native multipole and vdW kernels gives the Reference kernel's forces (float grids:
    1e-6) and, to the last bit of the fixed-point force sums, the forces of the scan over all atoms."""
    from amoeba_water_case import run_amoeba_water_case
    r = run_amoeba_water_case(tmp_path, True, 8, 32, False)
    print(r)
    assert r["reference"][0] < 5e-6 and r["reference"][1] < 5e-6
    assert r["full_scan"][0] < 1e-9 and r["full_scan"][1] < 1e-12


@needs_emu
def test_amoeba_list_builder_leaves_out_blocks_in_the_tile_frame(tmp_path):
    """3 000-atom AMOEBA water box (3.1 nm: every tile's half extent + list radius stays below half the box, so the builder works in the
    tile frame -- candidates moved once to the image nearest to the owner tile, 32-slot blocks that no owner of a wavefront can reach left
    out by their bounding boxes, round 5): the same forces as the scan over all atoms, to the last bit of the fixed-point sums."""
    from amoeba_water_case import run_amoeba_water_case
    r = run_amoeba_water_case(tmp_path, True, 10, 40, False, vdw_cutoff=0.8, with_reference=False)
    print(r)
    assert r["full_scan"][0] < 1e-9 and r["full_scan"][1] < 1e-12
    # the same with lists that start at 8 entries per atom: the builder reports what would have fitted, the plugin grows them
    r = run_amoeba_water_case(tmp_path, True, 10, 40, False, vdw_cutoff=0.8, with_reference=False, tiles_env={"OPENMM_HIP_AMOEBA_PAIR_CAP": "8"})
    assert r["full_scan"][0] < 1e-9 and r["full_scan"][1] < 1e-12


@needs_emu
def test_amoeba_mutual_solver_learns_of_overflowed_lists_at_its_first_wait(tmp_path):
    """Mutual polarization with pair lists that start at 8 entries per atom: the multipole call does not wait for its list builder (round 5:
    the overflow word travels to the host among the solver's sums), runs field kernels and solver iterations on the truncated lists, learns
    of the overflow at the solver's first wait and returns -2 before anything has been added to the forces or the history; the plugin
    grows the lists and calls again.  Forces and energy: those of the scan over all atoms (which grows its lists the same way)."""
    from amoeba_water_case import run_amoeba_water_case
    r = run_amoeba_water_case(tmp_path, True, 6, 24, True, with_reference=False, tiles_env={"OPENMM_HIP_AMOEBA_PAIR_CAP": "8"})
    print(r)
    assert r["full_scan"][0] < 1e-6 and r["full_scan"][1] < 1e-9          # (two converged solves: equal to the solver's tolerance, not to the last bit)


@needs_emu
def test_amoeba_dynamics_with_list_skin_and_predicted_dipoles_walks_the_same_trajectory(tmp_path):
    """Eight Verlet steps of a relaxed 375-atom AMOEBA water box (mutual polarization to 1e-6 D) on the emulator: lists with a Verlet skin
    rebuilt on displacement + the solver started from dipoles extrapolated from earlier steps + convergence decided on the device, against
    rebuilding the lists and solving from the direct dipoles at every step (tests/amoeba_dynamics_case.py; the GPU test runs 12 steps)."""
    from amoeba_dynamics_case import run_amoeba_dynamics_case
    r = run_amoeba_dynamics_case(tmp_path, True, steps=8, minimize=15)
    print(r)
    assert r["dpos"] < 1e-6 and r["dforce"] < 5e-5 and r["denergy"] < 1e-6
    ev, builds = r["now"]["evaluations"], r["now"]["builds"]
    assert ev[0] >= 8 and ev[1] >= 8 and builds[0] <= ev[0] // 2 and builds[1] <= ev[1] // 2, "the lists were not reused"
    assert sum(r["now"]["iterations"]) < sum(r["round3"]["iterations"]), "the extrapolated first guess saved no iterations"


@needs_emu
def test_amoeba2009_dhfr_solute_forces_and_mts_langevin_steps_against_reference_platform(tmp_path):
    """Every kind of term amoeba2009 puts on a protein (the 2 489-atom solute of the amoebapme benchmark System) on the emulated HIP platform
    against the Reference platform, by the force groups of examples/benchmark.py, and three steps of its MTSLangevinIntegrator (as a CustomIntegrator on
    the device interpreter; without friction, so that both platforms walk the same path), then the thermostat on its own (tests/amoeba_dhfr_case.py)."""
    from amoeba_dhfr_case import run_amoeba_dhfr_case
    r = run_amoeba_dhfr_case(tmp_path, True)
    print(r)
    assert r["native"][0] >= 1 and r["native"][1] >= 1, "the native AMOEBA kernels did not run"
    assert r["native"][2] >= 7, "the native kernels of the valence terms (bond, angle, in-plane angle, out-of-plane bend, stretch-bend, pi-torsion, torsion-torsion) did not run"
    assert r["force_valence"] < 1e-6 and r["energy_valence"] < 1e-9
    assert r["force_nonbonded"] < 2e-4 and r["energy_nonbonded"] < 1e-5
    assert r["dpos"] < 2e-6 and r["dvel"] < 2e-3
    assert r["mode"] == "device, custom integrator", "the MTS integrator did not run on the device interpreter"
    assert r["evaluations_per_step"][1] < 2, "the slow force group was evaluated more than once per step (+ the closing energy query)"
    assert 270 < r["thermostat_temperature"] < 330


@needs_emu
def test_multi_gpu_context_rejects_forces_with_plugin_native_kernels():
    """The AMOEBA kernels of libOpenMMAmoebaHIP.so evaluate the whole system on one GPU: a Context spread over ranks must refuse them
    (it used to accept them silently and return R times the energy) -- before any communicator is created."""
    code = r'''
import sys
sys.path.insert(0, %r)
from openmm_amd import harness as H, testsystems as T
H.load_amoeba_plugins(emulated=True)
w = T.amoeba_water_box(4, seed=3, polarization=H.Direct, cutoff=0.6, vdw_cutoff=0.6, grid=(16,) * 3, a_ewald=5.4459052)
s, mp, vdw = w.build()
try:
    H.Context(s, H.Integrator(H.VERLET, 0.001), "HIP", {"Ranks": "2", "Rank": "0", "CommId": "0" * 256})
except H.OpenMMError as e:
    print("REFUSED:", e)
''' % ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "REFUSED:" in out.stdout and "another plugin" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


@needs_emu
def test_amoeba_fallback_path_still_works_without_the_native_plugin():
    """HIP_AMOEBA_FALLBACK_ONLY=1: the AMOEBA plugin's own Reference kernels as fallback forces on a HIP Context (round 2's path)."""
    exe = os.path.join(EMU_BUILD, "tests", "TestHipAmoebaVdwForce")
    if not os.path.exists(exe):
        pytest.skip("not built")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=900, env=dict(os.environ, HIP_AMOEBA_FALLBACK_ONLY="1"))
    assert out.returncode == 0 and "Done" in out.stdout and "evaluations: vdw 0 multipole 0" in out.stdout, out.stdout[-2000:]


@needs_emu
def test_context_round_trip_through_plugin_loader():
    """Context('HIP') via Platform::loadPluginLibrary -> registerPlatforms(), forces vs the Reference platform, a few
    LangevinMiddle steps with SETTLE; run in a child process because the emulated and product plugins must not mix."""
    code = r'''
import sys, numpy as np
sys.path.insert(0, %r)
from openmm_amd import harness as H, testsystems as T
H.load_hip_platform(emulated=True)
assert "HIP" in H.platform_names()
w = T.water_box(5, seed=2, cutoff=0.7)
alpha = float(np.sqrt(-np.log(2 * w.ewald_tol)) / w.cutoff)
w.pme_params = (alpha, 16, 16, 16)
res = {}
for plat in ("Reference", "HIP"):
    s, nb = w.build()
    integ = H.Integrator(H.LANGEVIN_MIDDLE, 0.002, 300.0, 1.0, seed=5)
    c = H.Context(s, integ, plat)
    c.setPositions(w.positions)
    res[plat] = c.getState(getForces=True, getEnergy=True)
    if plat == "HIP":
        assert nb.getPMEParametersInContext(c)[1:] == (16, 16, 16)
        c.setVelocitiesToTemperature(300.0, 3)
        integ.step(5)
        p = c.getState(getPositions=True).positions.reshape(-1, 3, 3)
        d = np.linalg.norm(p[:, 0] - p[:, 1], axis=1)
        assert abs(d - T.TIP3P["dOH"]).max() < 1e-5, d
    c.close()
fr, fh = res["Reference"].forces, res["HIP"].forces
rms = np.sqrt((fr ** 2).sum(1).mean())
err = np.sqrt(((fh - fr) ** 2).sum(1)).max() / rms
assert err < 1e-4, err
assert abs(res["HIP"].potentialEnergy - res["Reference"].potentialEnergy) < 0.05
print("OK", err)
''' % ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900)
    assert "OK" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


@needs_emu
def test_fused_step_equals_staged_kernels_emulated():
    """One-launch LangevinMiddle step (SETTLE + thermostat + folded CM removal) against the staged kernels, and the folded
    exclusion correction against the term list, on a small water box (child process, emulated plugin)."""
    code = r'''
import os, sys, numpy as np
sys.path.insert(0, %r)
from openmm_amd import harness as H, testsystems as T
H.load_hip_platform(emulated=True)
w = T.water_box(4, seed=3, cutoff=0.6)
alpha = float(np.sqrt(-np.log(2 * w.ewald_tol)) / w.cutoff)
w.pme_params = (alpha, 12, 12, 12)
w.cm_remover = True
def run(env, kind, steps):
    os.environ.update(env)
    s, nb = w.build()
    integ = H.Integrator(kind, 0.002, 300.0, 1.0, seed=5, constraintTolerance=1e-6)
    c = H.Context(s, integ, "HIP")
    c.setPositions(w.positions)
    c.applyConstraints(1e-6)
    c.setVelocitiesToTemperature(300.0, 3)
    integ.step(steps)
    st = c.getState(getPositions=True, getVelocities=True, getForces=True, getEnergy=True)
    c.close()
    return st
for kind in (H.VERLET, H.LANGEVIN_MIDDLE):
    a = run({"OPENMM_HIP_DISABLE_FUSED_STEP": "0"}, kind, 4)
    b = run({"OPENMM_HIP_DISABLE_FUSED_STEP": "1"}, kind, 4)
    assert np.abs(a.positions - b.positions).max() < 1e-7, np.abs(a.positions - b.positions).max()
    assert np.abs(a.velocities - b.velocities).max() < 1e-4
a = run({"OPENMM_HIP_NO_FOLDED_EXCLUSIONS": "0"}, H.VERLET, 0)
b = run({"OPENMM_HIP_NO_FOLDED_EXCLUSIONS": "1"}, H.VERLET, 0)
rms = np.sqrt((b.forces ** 2).sum(1).mean())
assert np.sqrt(((a.forces - b.forces) ** 2).sum(1)).max() / rms < 2e-6
assert abs(a.potentialEnergy - b.potentialEnergy) < 5e-3      # both sum ~3e4 kJ/mol of single-precision erf terms
# pair kernel riding on the three FFT launches vs one launch per kernel
a = run({"OPENMM_HIP_NO_PAIRS_WITH_FFT": "0"}, H.VERLET, 3)
b = run({"OPENMM_HIP_NO_PAIRS_WITH_FFT": "1"}, H.VERLET, 3)
os.environ["OPENMM_HIP_NO_PAIRS_WITH_FFT"] = "0"
assert np.abs(a.positions - b.positions).max() < 1e-9
assert np.sqrt(((a.forces - b.forces) ** 2).sum(1)).max() / rms < 1e-6
assert abs(a.potentialEnergy - b.potentialEnergy) < 1e-3
print("OK")
''' % ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900)
    assert "OK" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


@needs_emu
def test_neighbour_list_overflow_is_recovered(tmp_path):
    """A device-triggered rebuild that runs out of rows freezes the integration on the device; the host grows the list and
    redoes the skipped steps: same trajectory as an undisturbed run (tests/overflow_case.py).  Tolerance: after the recovery the
    two runs hold different lists, so their float32 force sums differ in order (1e-5 of the RMS force, the same noise the
    multi-rank tests allow); over 40 steps that random-walks to 1e-4 nm/ps on a hydrogen.  A wrong replay (one step with
    another step's noise) is off by 1e-2 nm/ps."""
    from overflow_case import run_overflow_case
    print(run_overflow_case(tmp_path, True, 7, 20, 5e-6, 5e-4, step_counts=(5,)))      # found at the download; the lazy read-back is the next test's


@needs_emu
def test_neighbour_list_overflow_is_recovered_with_the_side_stream(tmp_path):
    """... and with reciprocal space on its own stream (the default above 60 000 atoms and on decomposed runs)."""
    from overflow_case import run_overflow_case
    print(run_overflow_case(tmp_path, True, 7, 20, 5e-6, 5e-4, props={"DisablePmeStream": "false"}, step_counts=(18,)))      # found by the lazy read-back


@needs_emu
def test_pruned_list_stays_complete_over_a_run(tmp_path):
    """The dual pair list on the emulated platform (tests/pruned_list_case.py): rows cut to cutoff + 0.036 nm from rows built with
    cutoff + 0.27 nm, re-cut on the device's own displacement check; forces after 14 steps against the Reference platform."""
    from pruned_list_case import run_pruned_list_case
    print(run_pruned_list_case(tmp_path, True, 10, 30, 1, 14))


@needs_emu
def test_native_ljpme_matches_the_reference_platform():
    """tests/ljpme_case.py on the emulated kernels (own process: one plugin build per process)."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); from openmm_amd import harness as H; H.load_hip_platform(emulated=True); "
            "from ljpme_case import run_ljpme_case; run_ljpme_case(); print('OK')") % (ROOT, os.path.join(ROOT, "tests"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900)
    assert "OK" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


@needs_emu
def test_custom_integrator_interpreter_through_the_c_abi(K):
    """ommhip_vm_per_dof with hand-written postfix programs: three computations in one launch (a per-DOF variable, v, x -- each reading what
    the one before wrote), a sum over the degrees of freedom, a massless particle left alone -- against numpy."""
    for name, (got, expected) in KC.run_vm(K).items():
        assert np.allclose(got, expected, rtol=1e-13, atol=1e-13), name


@needs_emu
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



@needs_emu
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


@needs_emu
def test_ewald_reciprocal_sum_through_the_c_abi(K):
    """ommhip_ewald_reciprocal against the numpy k-sum (pinned to the Reference platform and TestEwald.h's Gromacs
    golden): forces to 1e-9 of the largest (fixed-point quantum 2^-32), energy 1e-12."""
    f, e, f_or, e_or = KC.run_ewald_reciprocal(K)
    assert abs(e - e_or) < 1e-11 * abs(e_or), (e, e_or)
    assert np.abs(f - f_or).max() < 1e-9 * np.abs(f_or).max(), (np.abs(f - f_or).max(), np.abs(f_or).max())


@needs_emu
def test_verlet_trajectory_with_settle_shake_and_ccma_follows_the_reference_platform(tmp_path):
    """Ten deterministic Verlet steps (1 fs) of the constraint zoo -- SETTLE waters, SHAKE clusters, a CCMA stretch, PME + bonded forces --
    and of the chain with SETTLE + SHAKE only (fused one-launch step) on the emulated HIP platform against the Reference platform
    (tests/verlet_trajectory_case.py).  Positions within 1e-6 nm, velocities within 1e-4 nm/ps: the float32 pair / PME arithmetic of
    "mixed" precision is the only difference between the two."""
    from verlet_trajectory_case import run_verlet_trajectory_case
    r = run_verlet_trajectory_case(tmp_path, True)
    print(r)
    assert r["zoo"]["mode"] == "device" and r["chain"]["mode"] == "device"
    assert r["zoo"]["partition"] == "settle 660 shake 33 ccma 49", r["zoo"]["partition"]
    assert r["chain"]["partition"].endswith("ccma 0")
    for name in ("zoo", "chain"):
        assert r[name]["moved"] > 5e-3                      # the atoms went somewhere
        assert r[name]["dpos"] < 1e-6 and r[name]["dvel"] < 1e-4, r[name]
        assert r[name]["ke_rel"] < 1e-6 and r[name]["constraints"] < 1e-7
        assert r[name]["times"][0] == r[name]["times"][1]


@needs_emu
def test_ewald_ksum_forces_on_the_reference_tests_nacl_system_within_1e_4():
    """Row a9 at the north-star tolerance on the emulated platform: the amorphous NaCl of tests/TestEwald.h (classic Ewald k-sum) against the
    Reference platform, every atom within 1e-4 of the RMS force (the reference body itself asserts 1e-2)."""
    code = r'''
import sys, numpy as np
sys.path.insert(0, %r)
from openmm_amd import harness as H, testsystems as T
H.load_hip_platform(emulated=True)
w = T.nacl_amorph()
st = {}
for plat in ("Reference", "HIP"):
    system, nb = w.build()
    ctx = H.Context(system, H.Integrator(H.VERLET, 0.001), plat)
    ctx.setPositions(w.positions)
    st[plat] = ctx.getState(getForces=True, getEnergy=True)
    ctx.close()
fr, fh = st["Reference"].forces, st["HIP"].forces
err = np.sqrt(((fh - fr) ** 2).sum(1)).max() / np.sqrt((fr ** 2).sum(1).mean())
assert err < 1e-4, err
assert abs(st["HIP"].potentialEnergy - st["Reference"].potentialEnergy) < 1e-5 * abs(st["Reference"].potentialEnergy)
print("OK", err)
''' % ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=1800)
    assert "OK" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


@needs_emu
def test_stochastic_integrators_continue_their_noise_from_a_checkpoint(tmp_path):
    """A device CustomIntegrator with gaussian per-DOF noise and a host-drawn global (the benchmark's MTSLangevinIntegrator plus a ComputeGlobal
    draw) and the native LangevinMiddle integrator: 6 steps after a checkpoint are the same whether the run went on, the Context was rewound
    to the checkpoint, or a NEW Context with seed 0 loaded it (tests/checkpoint_case.py)."""
    from checkpoint_case import run_checkpoint_case
    r = run_checkpoint_case(tmp_path, True)
    print(r)
    assert r["custom"]["mode"] == "device, custom integrator" and r["native"]["mode"] == "device"
    for kind in ("custom", "native"):
        # loading a checkpoint re-sorts the atoms, so float sums (grid, pair forces) come in another order: not bitwise, but far below what other noise would do (~1e-3 nm)
        assert max(r[kind]["same"][0], r[kind]["new"][0]) < 1e-6 and max(r[kind]["same"][1], r[kind]["new"][1]) < 1e-4, r[kind]        # one step of foreign noise moves an oxygen by 5e-5 nm
        assert r[kind]["times"][0] == r[kind]["times"][1] == r[kind]["times"][2]


@needs_emu
def test_custom_integrator_too_deep_for_the_interpreter_runs_in_host_mode():
    """An expression that needs more than OMMHIP_VM_STACK slots is found when the Context is classified (HipIntegrateCustomStepKernel::supports),
    so the integrator runs on the Reference kernel instead of throwing at its first step."""
    code = r'''
import sys, numpy as np
sys.path.insert(0, %r)
from openmm_amd import harness as H, testsystems as T
H.load_hip_platform(emulated=True)
w = T.water_box(3, seed=2, cutoff=0.4, method=H.CutoffPeriodic)
modes = []
for depth in (8, 24):
    s, nb = w.build()
    integ = H.CustomIntegrator(0.001, seed=1, constraintTolerance=1e-7)
    expr = "v"
    for k in range(depth):
        expr = "1e-3*atan(v)+(%%s)" %% expr          # right-nested, and nothing Lepton's optimizer folds away (the depth is checked on the optimized tree, the one that is translated): one more stack slot per level
    integ.addComputePerDof("v", "v+dt*f/m")
    integ.addComputePerDof("x", "x+dt*(" + expr + ")")
    integ.addConstrainPositions()
    c = H.Context(s, integ, "HIP")
    c.setPositions(w.positions); c.applyConstraints(1e-7)
    integ.step(2)
    modes.append(c.getPlatformProperty("IntegrationMode"))
    assert np.all(np.isfinite(c.getState(getPositions=True).positions))
    c.close()
print("MODES", modes)
assert modes == ["device, custom integrator", "host"], modes
print("OK")
''' % ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900)
    assert "OK" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]
