"""world_size-2 `gloo` tests on CPU.
(1) The timing reduction bench.py uses for N > 1 (openmm_amd/multirank.py: MAX over ranks of the timed region).
(2) Force decomposition through the C ABI: two ranks build the neighbour list for one half of the i-blocks each
    (ommhip_neighbor_list.first_block / owned_blocks), run the pair kernel, and all-reduce their fixed-point force buffers --
    the sum must be bit for bit the single-rank buffer.
(3) The WHOLE step of a domain-decomposed run on two ranks against the single-rank run of the same box.
All kernels run on the CPU SIMT emulator build (tests/emu); the collectives of (3) go through the plugin's callback transport."""
import os
import subprocess
import sys

from conftest import ROOT

CHILD = r'''
import os, sys, time
sys.path.insert(0, %r)
import torch, torch.distributed as dist
from openmm_amd.multirank import max_over_ranks, ns_per_day
dist.init_process_group(backend="gloo")
rank = dist.get_rank()
elapsed = max_over_ranks(1.0 + rank, dist, device="cpu")          # rank 1 is slower
assert abs(elapsed - 2.0) < 1e-12
if rank == 0:
    # ONE simulation advanced 1000 steps of 2 fs in max(1, 2) = 2 s
    value = ns_per_day(elapsed, 1000, 2.0)
    assert abs(value - 2.0e-6 * 1000 / 2.0 * 86400) < 1e-9
    print("OK", value)
dist.destroy_process_group()
'''


def test_two_rank_timing_reduction_on_gloo(tmp_path):
    script = tmp_path / "child.py"
    script.write_text(CHILD % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29533", str(script)],
                         capture_output=True, text=True, timeout=600, env=env)
    assert "OK" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


DECOMP_CHILD = r'''
import os, sys
sys.path.insert(0, %r)
sys.path.insert(0, os.path.join(%r, "tests"))
import numpy as np, torch, torch.distributed as dist
from openmm_amd import capi
import kernel_cases as KC
from oracle import nonbonded as ONB
dist.init_process_group(backend="gloo")
rank, world = dist.get_rank(), dist.get_world_size()
K = capi.load(%r)
EXCL = [(i, i + 1) for i in range(0, 600, 3)] + [(i, i + 2) for i in range(0, 600, 3)]
n, cutoff, L = 1500, 0.7, 3.4
blocks = (n + 31) // 32
for compact in (False, True):
    # every rank: the whole evaluation (reference for the bit-for-bit comparison) ...
    full = KC.run_direct_space(K, n, ONB.PME, cutoff, L, EXCL, compact=compact)
    whole = KC.LAST_FIXED_POINT_FORCES.copy()
    # ... and its share of the i-blocks
    first = blocks * rank // world
    count = blocks * (rank + 1) // world - first
    part = KC.run_direct_space(K, n, ONB.PME, cutoff, L, EXCL, compact=compact, block_range=(first, count))
    mine = torch.from_numpy(KC.LAST_FIXED_POINT_FORCES.copy())
    energy = torch.tensor([part[1]], dtype=torch.float64)
    assert part[4][1] < full[4][1]                       # fewer chunks than the full list
    dist.all_reduce(mine, op=dist.ReduceOp.SUM)
    dist.all_reduce(energy, op=dist.ReduceOp.SUM)
    assert np.array_equal(mine.numpy(), whole), "summed fixed-point forces differ from the single-rank buffer"
    assert abs(float(energy) - full[1]) < 1e-9 * abs(full[1]) + 1e-6
if rank == 0:
    print("OK")
dist.destroy_process_group()
'''


def test_two_rank_force_decomposition_is_bit_exact(tmp_path):
    import pytest
    from conftest import EMU_BUILD
    emu_lib = os.path.join(EMU_BUILD, "libopenmm_hip_kernels.so")
    if not os.path.exists(emu_lib):
        pytest.skip("emulated kernel library not built (run __graft_entry__.build())")
    script = tmp_path / "decomp_child.py"
    script.write_text(DECOMP_CHILD % (ROOT, ROOT, emu_lib))
    # OPENMM_HIP_NL_PERSISTENT=0: one builder workgroup per i-block for the whole list AND for the shares.  The comparison is bit for bit, and
    # the order of the chunks in the list (the order in which the builder's workgroups allocate them -- deterministic only on the emulator)
    # decides which chunks of an i-block a pair wavefront sums in float before it converts to fixed point; the resident builder
    # workgroups that long lists get by default (the whole list here, not the shares) allocate in another order.
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", OPENMM_HIP_NL_PERSISTENT="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29541", str(script)],
                         capture_output=True, text=True, timeout=900, env=env)
    assert "OK" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


DD_CHILD = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np, torch, torch.distributed as dist
from openmm_amd import harness as H, testsystems as T, multirank as MR
dist.init_process_group(backend="gloo")
rank, world = dist.get_rank(), dist.get_world_size()
H.load_hip_platform(emulated=%r)
device = %r
if device == "rank": device = rank                 # one GPU per rank (RCCL)
TRANSPORT = os.environ.get("DD_TEST_TRANSPORT", "gloo")


def run(props, w, steps):
    system, nb = w.build()
    integ = H.Integrator(H.LANGEVIN_MIDDLE, 0.002, 300.0, 1.0, seed=7)
    ctx = H.Context(system, integ, "HIP", props)
    ctx.setPositions(w.positions)
    ctx.applyConstraints(1e-6)
    ctx.setVelocitiesToTemperature(300.0, 3)
    st0 = ctx.getState(getForces=True, getEnergy=True, getPositions=True)
    integ.step(steps)
    st1 = ctx.getState(getPositions=True, getVelocities=True, getEnergy=True, getForces=True)
    st1.box = ctx.getPeriodicBoxVectors()
    info = (ctx.getPlatformProperty("Ranks"), ctx.getPlatformProperty("CommId"))
    global DD_INFO
    DD_INFO = H.domain_info() if "Ranks" in props else None
    ctx.close()
    return st0, st1, info


os.environ["OPENMM_HIP_REORDER_INTERVAL"] = "3"      # a re-sort (units change owner) inside the short run
os.environ.setdefault("OPENMM_HIP_REORDER_LAG", "1")   # ... applied one step after its snapshot (the default lag is longer than the run)
EXTRA_CASES = %s
STEPS = %d
for label, w, grid in (("water, halo", T.water_box(8, seed=5), 24), ("solvated chain, halo", T.small_solvated_chain(seed=3), 24)) + EXTRA_CASES:
    # a 32^3 grid marks the case that runs with the (opt-in) tile spreading: 16 own planes per rank = one tile along x, clipped to the slab
    tiles = isinstance(grid, int) and grid >= 32
    if tiles: os.environ["OPENMM_HIP_TILE_SPREAD_MIN_ATOMS"] = "1"
    else: os.environ.pop("OPENMM_HIP_TILE_SPREAD_MIN_ATOMS", None)
    if grid:
        w.pme_params = (float(np.sqrt(-np.log(2 * w.ewald_tol)) / w.cutoff),) + (tuple(grid) if isinstance(grid, tuple) else (grid, grid, grid))
    w.cm_remover = True
    base = {} if device is None else {"DeviceIndex": str(device)}
    one0, one1, _ = run(dict(base), w, %d if STEPS <= 20 else 0)                # single-rank run of the same box, on every rank
    props = MR.domain_properties(dist, transport=TRANSPORT, device_index=device, emulated=%r)
    dd0, dd1, info = run(props, w, %d)
    assert info == (str(world), "rccl" if TRANSPORT == "rccl" else "callback"), info
    if "halo" in label:
        # ranks, halo mode, slots per rank, slots converted per step, bytes sent / received per step, re-sorts
        assert DD_INFO[1] == 1, ("expected the halo exchange", DD_INFO)
        if "sections" in label:
            assert DD_INFO[3] < world * DD_INFO[2] and DD_INFO[5] < 16 * DD_INFO[2] * (world - 1), ("the halo should be smaller than the box", DD_INFO)
        if "drift" in label:
            assert DD_INFO[6] >= 2, ("a drift-triggered re-sort was expected", DD_INFO)
        if "tight list" in label:
            assert DD_INFO[6] >= 2, ("a re-sort asked for by the nearly full list was expected", DD_INFO)
        if "half-shell" in label:
            assert DD_INFO[7] > 1, ("expected half-shell evaluation with partners from the lower neighbour", DD_INFO)
        if "both sides" in label:
            assert DD_INFO[7] == 0, DD_INFO
    if "replicated" in label:
        assert DD_INFO[1] == 0, DD_INFO
    rms = np.sqrt((one0.forces ** 2).sum(1).mean())
    err0 = np.abs(dd0.forces - one0.forces).max() / rms
    # (a triclinic box: the decomposed run holds every atom in the image whose three box coefficients lie in [0, 1), the single-rank run in
    # the image with x, y, z in [0, edge) -- other float32 roundings of the same positions; 24 000 atoms on the GPU: 3.5e-5 at the worst
    # atom, 1.7e-5 at the 99.9th percentile, and the same figures with replicated positions, tools/diag_triclinic_dd_noise.py)
    assert err0 < (6e-5 if "triclinic" in label else 3e-5), ("initial forces", err0)       # float32 summation-order noise; TestCudaNonbondedForce.cpp:37-96 allows 1e-5 of each force in its multi-device mode
    # (1e-6 of the magnitude, with the floor of tests/test_gpu_platform.py for lattice starts whose terms nearly cancel)
    assert abs(dd0.potentialEnergy - one0.potentialEnergy) < 1e-6 * max(abs(one0.potentialEnergy), 5.0 * w.num_atoms) + 1e-3, (dd0.potentialEnergy, one0.potentialEnergy)
    dpos = np.abs(dd1.positions - one1.positions).max()
    dvel = np.abs(dd1.velocities - one1.velocities).max()
    if "barostat" in label:
        # the same Monte Carlo decisions on N GPUs as on one: the box went the same way (and did move)
        assert np.abs(dd1.box - one1.box).max() < 1e-9 * one1.box.max(), (dd1.box, one1.box)
        assert np.abs(np.diag(one1.box) - np.diag(w.box)).max() > 1e-6, "no volume move was accepted: the case tests nothing"
    if STEPS <= 20:
        # float32 force noise (1e-5 of the RMS force) integrated over the run; the tile spreading rounds each contribution to max|q| 2^-24
        big = w.num_atoms > 5000            # more atoms, a larger maximum of the same noise
        # the larger force noise of the other periodic images (above), integrated: 24 000 atoms on the GPU gave dvel 3.8e-4 after 10 steps on one
        # box (the rectangular bar is 1.5e-4); the force bars above and below are what guards against missing pairs (those show as 1e-4 and more)
        loose = 6.0 if "triclinic" in label else 1.0
        assert dpos < loose * (1e-6 if big else 3e-7) and dvel < loose * (1.5e-4 if tiles or big else 5e-5), ("trajectory", dpos, dvel)
        assert abs(dd1.kineticEnergy - one1.kineticEnergy) < 1e-6 * one1.kineticEnergy
        err1 = np.abs(dd1.forces - one1.forces).max() / rms
        assert err1 < 1e-4, ("final forces", err1)
    else:
        # a long run: the two trajectories drift apart as any two float32 runs of a chaotic system do (the replicated-position run of
        # round 2 shows the same 2e-5 nm after 80 steps); what must hold is that the decomposed forces are right WHERE THE DECOMPOSED RUN
        # IS -- a fresh single-rank Context evaluates them at its final positions
        system, nb = w.build()
        ctx = H.Context(system, H.Integrator(H.VERLET, 0.001), "HIP", dict(base))
        ctx.setPositions(dd1.positions)
        ref1 = ctx.getState(getForces=True, getEnergy=True)
        ctx.close()
        err1 = np.abs(dd1.forces - ref1.forces).max() / rms
        assert err1 < 6e-5, ("final forces against a single-rank evaluation of the same positions", err1)
        assert abs(dd1.potentialEnergy - ref1.potentialEnergy) < 1e-6 * max(abs(ref1.potentialEnergy), 5.0 * w.num_atoms) + 1e-3
    # every rank reports the same State
    check = torch.tensor([dd1.potentialEnergy, dd1.kineticEnergy, float(dd1.positions.sum())], dtype=torch.float64)
    both = [torch.zeros_like(check) for _ in range(world)]
    dist.all_gather(both, check)
    assert all(torch.equal(b, both[0]) for b in both), both
    if rank == 0:
        print(label, "forces", err0, err1, "trajectory", dpos, dvel, "domain", DD_INFO, flush=True)
if rank == 0:
    print("OK")
dist.destroy_process_group()
'''


def _run_dd_child(tmp_path, emulated, device, steps, port, nproc=2, cases=None, extra_cases="()", env=None):
    """extra_cases: Python source of a tuple of further (label, workload, grid) cases; a cubic grid >= 32 runs with tile spreading, a tuple is (nx, ny, nz)."""
    script = tmp_path / "dd_child.py"
    text = DD_CHILD % (ROOT, emulated, device, extra_cases, steps, steps, emulated, steps)
    if cases is not None:
        text = text.replace('(("water, halo", T.water_box(8, seed=5), 24), ("solvated chain, halo", T.small_solvated_chain(seed=3), 24))', cases)
    script.write_text(text)
    if env and "OPENMM_HIP_REORDER_INTERVAL" in env:
        text = text.replace('os.environ["OPENMM_HIP_REORDER_INTERVAL"] = "3"', 'os.environ["OPENMM_HIP_REORDER_INTERVAL"] = "%s"' % env["OPENMM_HIP_REORDER_INTERVAL"])
        script.write_text(text)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", **(env or {}))
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                         capture_output=True, text=True, timeout=1500, env=env)
    assert "OK" in out.stdout, out.stdout[-3000:] + out.stderr[-4000:]
    return out.stdout


def test_two_rank_domain_decomposition_whole_step_on_emulator(tmp_path):
    """The WHOLE step of a decomposed run with world_size 2: positions all-gathered, list + pair kernel for the
    owned blocks (cross-rank pairs evaluated on both sides), slab-decomposed PME with its two all-to-alls and halo planes,
    per-rank integration with SETTLE and the CM-motion remover riding in the all-gather -- against the single-rank run of
    the same box.  Kernels run on the CPU SIMT emulator, collectives on gloo through the callback transport."""
    import pytest
    from conftest import EMU_BUILD
    if not os.path.exists(os.path.join(EMU_BUILD, "libOpenMMHIP.so")):
        pytest.skip("emulated plugin not built (run __graft_entry__.build())")
    _run_dd_child(tmp_path, True, None, 4, 29547, extra_cases='(("water, tile spreading", T.water_box(8, seed=5), 32),)')


def test_halo_exchange_with_distinct_sections_on_emulator(tmp_path):
    """Halo mode proper: a 3.7 nm box with a 0.5 nm cutoff, where a slab is wider than twice the halo, so a rank's
    range really has four sections (needed below / both / above / by nobody), a rank converts fewer slots than the box holds and
    receives less than an all-gather would bring -- two ranks (both neighbours are the same peer) and four (ring neighbours that
    are different ranks, second neighbours never seen).  Same bar as the replicated runs: trajectory of the single-rank run."""
    import pytest
    from conftest import EMU_BUILD
    if not os.path.exists(os.path.join(EMU_BUILD, "libOpenMMHIP.so")):
        pytest.skip("emulated plugin not built (run __graft_entry__.build())")
    env = {"OPENMM_HIP_DD_DRIFT": "0.03"}
    _run_dd_child(tmp_path, True, None, 4, 29561, env=env, cases='(("water, halo sections, half-shell", T.water_box(12, seed=5, cutoff=0.5), None),)')
    _run_dd_child(tmp_path, True, None, 4, 29565, nproc=4, env=env, cases='(("water, halo sections, half-shell, 4 ranks", T.water_box(12, seed=5, cutoff=0.5), None),)')


def test_half_shell_evaluation_with_force_return_on_emulator(tmp_path):
    """Half-shell mode: a pair -- and a bonded term -- that crosses a slab boundary is evaluated ONCE, by the rank above
    the boundary, which holds the lower rank's boundary section and returns the forces it computed on those atoms (ommhip_comm_halo_return);
    a rank sees only a thin section of its upper neighbour, for charge spreading.  Three ranks and a 150-atom chain (bonds, angles, torsions,
    1-4s, exclusions, X-H clusters) that lies across both inner boundaries; the same box with the pairs on both sides (the round-3 scheme,
    OPENMM_HIP_DD_BOTH_SIDES=1) as the control.  Same bar as every decomposed run: forces and trajectory of the single-rank run."""
    import pytest
    from conftest import EMU_BUILD
    if not os.path.exists(os.path.join(EMU_BUILD, "libOpenMMHIP.so")):
        pytest.skip("emulated plugin not built (run __graft_entry__.build())")
    chain = 'T.with_cutoff(T.small_solvated_chain(seed=3), 0.4)'
    _run_dd_child(tmp_path, True, None, 4, 29581, nproc=3, env={"OPENMM_HIP_DD_DRIFT": "0.02"}, cases='(("solvated chain, halo, half-shell", %s, 48),)' % chain)
    _run_dd_child(tmp_path, True, None, 4, 29585, nproc=3, env={"OPENMM_HIP_DD_DRIFT": "0.02", "OPENMM_HIP_DD_BOTH_SIDES": "1"}, cases='(("solvated chain, halo, both sides", %s, 48),)' % chain)


def test_triclinic_box_domain_decomposition_on_emulator(tmp_path):
    """A triclinic box on N ranks: the slabs are cut in the first box fraction (planes parallel to b and c), the wire records
    are the three box fractions, and every Cartesian length that enters the halo widths is stretched by |grad xi|.  Three ranks with distinct
    sections, half-shell evaluation and a re-sort inside the run; two ranks of a smaller box whose sections cover the slabs.  Same bar as every
    decomposed run: forces and trajectory of the single-rank run of the same box."""
    import pytest
    from conftest import EMU_BUILD
    if not os.path.exists(os.path.join(EMU_BUILD, "libOpenMMHIP.so")):
        pytest.skip("emulated plugin not built (run __graft_entry__.build())")
    env = {"OPENMM_HIP_DD_DRIFT": "0.03"}
    _run_dd_child(tmp_path, True, None, 4, 29641, nproc=3, env=env,
                  cases='(("water, triclinic, halo sections, half-shell", T.sheared(T.water_box(12, seed=5, cutoff=0.5), 0.6, -0.5, 0.8), None),)')
    _run_dd_child(tmp_path, True, None, 4, 29645, env=env, cases='(("water, triclinic, halo", T.sheared(T.water_box(8, seed=5), 0.5, -0.4, 0.3), 24),)')
    # a chain with bonds / angles / torsions / 1-4s / exclusions across both inner boundaries (the foreign atoms' double-precision positions
    # follow the wire records through the triclinic minimum image), half-shell evaluation; every atom sheared, constraints applied by the case
    _run_dd_child(tmp_path, True, None, 4, 29643, nproc=3, env={"OPENMM_HIP_DD_DRIFT": "0.02"},
                  cases='(("solvated chain, triclinic, halo, half-shell", T.sheared(T.with_cutoff(T.small_solvated_chain(seed=3), 0.4), 0.5, -0.4, 0.6, affine=True), 48),)')
    # ... and with a MonteCarloBarostat: the box changes (all three vectors scale), the slabs and sections are cut again for it
    _run_dd_child(tmp_path, True, None, 8, 29647, cases='(("water, triclinic, halo, barostat", T.with_barostat(T.sheared(T.water_box(8, seed=5), 0.5, -0.4, 0.3), 1.0, 300.0, 2, 11), 24),)')


def test_halo_drift_guard_triggers_a_common_resort_on_emulator(tmp_path):
    """An atom that drifts half the allowed margin raises a flag that travels in its rank's trailer; every rank finds it at the same
    evaluation and they re-sort together (no agreement collective).  With a margin of 0.06 nm the fastest oxygens cross the warning
    level (here 0.4 of it) within a few dozen steps; the order is applied four steps after its snapshot (OPENMM_HIP_REORDER_LAG): the run must re-sort by itself (OPENMM_HIP_REORDER_INTERVAL = 1000 never asks
    for one) and still follow the single-rank trajectory."""
    import pytest
    from conftest import EMU_BUILD
    if not os.path.exists(os.path.join(EMU_BUILD, "libOpenMMHIP.so")):
        pytest.skip("emulated plugin not built (run __graft_entry__.build())")
    _run_dd_child(tmp_path, True, None, 56, 29569, env={"OPENMM_HIP_DD_DRIFT": "0.06", "OPENMM_HIP_DD_WARN": "0.4", "OPENMM_HIP_REORDER_INTERVAL": "1000", "OPENMM_HIP_REORDER_LAG": "4"},
                  cases='(("water, halo drift", T.water_box(8, seed=5), 24),)')


def test_halo_drift_guard_resorts_at_once_when_the_margin_runs_out_on_emulator(tmp_path):
    """A hot system (the bench's 1M-atom lattice start melts at 1700 K): an atom uses up the margin before a re-sort that lags could
    apply.  At 80 % of the margin the flag in the trailer says so and every rank re-sorts at the next step instead -- here the lag
    (1000 steps) is longer than the run, so only that path can keep the 0.12 nm margin from overflowing (which raises)."""
    import pytest
    from conftest import EMU_BUILD
    if not os.path.exists(os.path.join(EMU_BUILD, "libOpenMMHIP.so")):
        pytest.skip("emulated plugin not built (run __graft_entry__.build())")
    _run_dd_child(tmp_path, True, None, 100, 29573, env={"OPENMM_HIP_DD_DRIFT": "0.12", "OPENMM_HIP_DD_WARN": "0.4", "OPENMM_HIP_REORDER_INTERVAL": "1000", "OPENMM_HIP_REORDER_LAG": "1000"},
                  cases='(("water, halo drift", T.water_box(8, seed=5), 24),)')


def test_nearly_full_list_triggers_a_common_resort_that_grows_it_on_emulator(tmp_path):
    """A rank whose neighbour list fills 7/8 of its allocation raises level 3 in its trailer: every rank re-sorts at the next step, and the
    rebuild after a re-sort is verified by the host and given 1.5 x its size -- so a list that grows during a run never gets as far as an
    overflow, which a decomposed run cannot undo (it raises).  The test hook leaves 8 % of room at the third evaluation; no other reason for
    a re-sort exists in the run (interval and lag 1000 steps, a drift margin nothing reaches)."""
    import pytest
    from conftest import EMU_BUILD
    if not os.path.exists(os.path.join(EMU_BUILD, "libOpenMMHIP.so")):
        pytest.skip("emulated plugin not built (run __graft_entry__.build())")
    _run_dd_child(tmp_path, True, None, 40, 29649, env={"OPENMM_HIP_DD_DRIFT": "0.2", "OPENMM_HIP_REORDER_INTERVAL": "1000", "OPENMM_HIP_REORDER_LAG": "1000",
                                                        "OPENMM_HIP_DEBUG_TIGHT_LIST_AFTER": "3"},
                  cases='(("water, halo, tight list", T.water_box(8, seed=5), 24),)')


def test_two_rank_run_with_the_barostat_on_emulator(tmp_path):
    """MonteCarloBarostat on a decomposed run: every rank scales all atoms from the owners' exact positions, the box
    changes, the re-sort cuts slabs and halo sections for the new box; the trial energies are rank-ordered sums and the barostat's random
    numbers come from its own seed, so all ranks take the decisions a single GPU takes.  Six steps, a volume move every second one."""
    import pytest
    from conftest import EMU_BUILD
    if not os.path.exists(os.path.join(EMU_BUILD, "libOpenMMHIP.so")):
        pytest.skip("emulated plugin not built (run __graft_entry__.build())")
    _run_dd_child(tmp_path, True, None, 6, 29577, cases='(("water, halo, barostat", T.with_barostat(T.water_box(8, seed=5), 1.0, 300.0, 2, 11), 24),)')


def test_eight_rank_domain_decomposition_on_emulator(tmp_path):
    """The decomposed step at the world size the scaling run uses (8 x-slabs): a 9.9 x 1.24 x 1.24 nm row of water boxes, 0.4 nm
    cutoff, so every slab (1.24 nm) is wider than twice the halo and the run is in halo mode with half-shell evaluation -- ring
    neighbours are six different pairs of ranks, second neighbours are never seen, the PME all-to-alls run among eight.  Same bar as
    the two-rank runs: forces and trajectory of the single-rank run of the same box.  The PME grid is pinned (192 x 24 x 24): a decomposed
    run rounds nx and ny up to multiples of the world size (HipKernels.cpp findLegalFftDimension), and two different grids differ by the
    Ewald tolerance (1e-3 of the RMS force here), not by float32 noise."""
    import pytest
    from conftest import EMU_BUILD
    if not os.path.exists(os.path.join(EMU_BUILD, "libOpenMMHIP.so")):
        pytest.skip("emulated plugin not built (run __graft_entry__.build())")
    out = _run_dd_child(tmp_path, True, None, 4, 29681, nproc=8, env={"OPENMM_HIP_DD_DRIFT": "0.03"},
                        cases='(("water row, halo sections, half-shell, 8 ranks", T.water_row(4, 8, seed=5), (192, 24, 24)),)')
    assert "domain [8, 1," in out, out[-1500:]            # eight ranks, halo mode


def test_four_rank_domain_decomposition_on_emulator(tmp_path):
    """Four slabs: every rank has two distinct ring neighbours for the potential planes, the all-to-alls move 4 x 4 chunks, and
    in a 2.5 nm box the 0.6 nm slabs are thinner than the cutoff -- each rank's partners span all the others."""
    import pytest
    from conftest import EMU_BUILD
    if not os.path.exists(os.path.join(EMU_BUILD, "libOpenMMHIP.so")):
        pytest.skip("emulated plugin not built (run __graft_entry__.build())")
    _run_dd_child(tmp_path, True, None, 3, 29557, nproc=4, cases='(("water, 4 ranks, replicated", T.water_box(8, seed=5), 24),)')


LAUNCHER = r"""
import json, os, sys, time
sys.path.insert(0, %r)
from openmm_amd import multirank as MR
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
hang = "import os, sys, time\nif os.environ['RANK'] == '1': time.sleep(600)\nprint('{\"attempt\": 0}')"
crash = "import os, sys\nif os.environ['RANK'] == '0': sys.exit(3)\nimport time; time.sleep(600)"
good = ("import os, torch, torch.distributed as dist\ndist.init_process_group('gloo')\nt = torch.ones(1); dist.all_reduce(t)\n"
        "print('note'); print('{\"attempt\": %%s, \"port\": %%s, \"sum\": %%d}' %% (os.environ['BENCH_ATTEMPT'], os.environ['MASTER_PORT'], int(t.item())), flush=True)\n"
        "dist.destroy_process_group()")      # (as bench.py ends: without it gloo's threads are torn down by the interpreter's exit, which aborts now and then on a busy machine)
t0 = time.monotonic()
idx, lines, notes = MR.run_attempts([[sys.executable, "-c", hang], [sys.executable, "-c", crash], [sys.executable, "-c", good]],
                                    rank, world, "127.0.0.1", int(os.environ["MASTER_PORT"]), timeout_s=20.0)      # (the good attempt imports torch: 8 s were not enough beside the 8-rank emulator test on 8 cores)
took = time.monotonic() - t0
out = json.loads([l for l in lines if l.startswith("{")][-1])
assert idx == 2 and out["attempt"] == 2 and out["port"] == int(os.environ["MASTER_PORT"]) + 3 and out["sum"] == world, (idx, out)
assert len(notes) == 2 and took < 80.0, (notes, took)
print("RANK", rank, "OK", notes, flush=True)
"""


def test_launchers_agree_on_a_fallback_when_one_rank_hangs_or_crashes(tmp_path):
    """bench.py's N > 1 safety net: attempt 0 never returns on rank 1 (killed after the timeout, and rank 0's successful child
    does not count), attempt 1 crashes on rank 0 (rank 1's hanging child is killed at once), attempt 2 works on both: its
    children form their own gloo group.  Started the way the driver starts bench.py (torch.distributed.run)."""
    script = tmp_path / "launcher.py"
    script.write_text(LAUNCHER % ROOT)
    proc = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                           "--master-addr", "127.0.0.1", "--master-port", "29611", str(script)],
                          capture_output=True, text=True, timeout=240)
    assert proc.returncode == 0 and "RANK 0 OK" in proc.stdout and "RANK 1 OK" in proc.stdout, proc.stdout + proc.stderr


def test_bench_multi_gpu_flow_on_emulator_falls_back_and_reports_one_line():
    """`bench.py --gpus 2` as the driver starts it, on the CPU emulator build: RCCL cannot start here, so both RCCL
    configurations fail on every rank, the launchers move to host-staged gloo together, and rank 0 prints ONE JSON line last:
    strong scaling of one box, with the failed attempts listed and the single-GPU time of the same box beside it."""
    import json
    import pytest
    from conftest import EMU_BUILD
    if not os.path.exists(os.path.join(EMU_BUILD, "libOpenMMHIP.so")):
        pytest.skip("emulator build missing")
    env = dict(os.environ, BENCH_EMULATED="1")
    proc = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                           "--master-port", "29671", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                           "--workload", "water1k", "--cpu-steps", "0", "--prepare-steps", "0", "--attempt-timeout", "200"],          # (WITH the roofline section, as the driver runs it: whatever rank 0 does there alone must not involve the Context -- the hang of rounds 4 - 5)
                          capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert proc.returncode == 0, proc.stdout[-2000:] + proc.stderr[-4000:]
    out = json.loads(proc.stdout.strip().splitlines()[-1])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["steps"] == 3 and out["value"] > 0
    assert "callback" in out["config"]["workload"] and len(out["config"]["attempts_failed"]) == 2
    assert out["single_gpu_same_box"]["value"] > 0
    assert out["single_gpu_same_box"]["initial_energy_kj_mol"]["rel_diff"] < 1e-5      # the decomposed run starts from the same energy as one GPU
