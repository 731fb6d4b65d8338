"""`-m gpu`: the domain-decomposed path on real hardware, as far as ONE GPU allows.

* one rank with a real RCCL communicator: every collective of the step goes through librccl (all-gather, grouped
  send/recv all-to-all, ring exchange -- with itself), on the plugin's stream, and the decomposed kernels (owned-slot pair
  kernel, slab PME with remapped transposes, trailer momentum) must reproduce the ordinary single-GPU run;
* two processes sharing the GPU (the reference's "one device listed twice" trick, platforms/cuda/tests/TestCudaNonbondedForce.cpp:37-96;
  RCCL refuses two ranks on one device, so the collectives go through the host-staged callback transport over gloo).
The 8-GPU run itself is the driver's (bench.py --gpus 8)."""
import os

import numpy as np
import pytest

from openmm_amd import harness as H, testsystems as T, multirank as MR

pytestmark = pytest.mark.gpu


def _run(w, props, steps, seed=7):
    system, nb = w.build()
    integ = H.Integrator(H.LANGEVIN_MIDDLE, 0.002, 300.0, 1.0, seed=seed)
    ctx = H.Context(system, integ, "HIP", props)
    ctx.setPositions(w.positions)
    ctx.applyConstraints(1e-6)
    ctx.setVelocitiesToTemperature(300.0, 3)
    st0 = ctx.getState(getForces=True, getEnergy=True)
    integ.step(steps)
    st1 = ctx.getState(getPositions=True, getVelocities=True, getEnergy=True, getForces=True)
    info = ctx.getPlatformProperty("CommId")
    ctx.close()
    return st0, st1, info


@pytest.mark.parametrize("n_side,grid,replicate", [(10, 32, False), (24, 0, False), (10, 32, True)])
def test_one_rank_with_rccl_reproduces_the_single_gpu_run(n_side, grid, replicate, monkeypatch):
    # (replicate: OPENMM_HIP_DD_REPLICATE=1, positions through the in-place all-gather of round 2 instead of the halo exchange -- the knob is
    #  read once per process, so the case runs in a child)
    if replicate:
        import subprocess, sys
        code = ("import os, sys, pytest; os.environ['OPENMM_HIP_DD_REPLICATE'] = '1'; "
                "sys.exit(pytest.main(['-q', '-x', '-p', 'no:cacheprovider', '-m', 'gpu', '%s::test_one_rank_with_rccl_reproduces_the_single_gpu_run[10-32-False]']))" % os.path.abspath(__file__))
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, cwd=os.path.dirname(os.path.abspath(__file__)))
        assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
        return
    H.load_hip_platform()
    w = T.water_box(n_side, seed=5)
    if grid:
        w.pme_params = (float(np.sqrt(-np.log(2 * w.ewald_tol)) / w.cutoff), grid, grid, grid)
    w.cm_remover = True
    one0, one1, _ = _run(w, {}, 20)
    dd0, dd1, transport = _run(w, {"Ranks": "1", "Rank": "0", "CommId": MR.new_rccl_id()}, 20)
    assert transport == "rccl"
    rms = np.sqrt((one0.forces ** 2).sum(1).mean())
    err = np.abs(dd0.forces - one0.forces).max() / rms
    print("one rank over RCCL vs plain: force max diff / rms %.3g, E %.6f vs %.6f" % (err, dd0.potentialEnergy, one0.potentialEnergy))
    assert err < 3e-5
    assert abs(dd0.potentialEnergy - one0.potentialEnergy) < 1e-6 * abs(one0.potentialEnergy) + 0.05
    # 20 steps: float32 force noise (different slot order, different summation order) grows, but slowly
    assert np.abs(dd1.positions - one1.positions).max() < 2e-5
    assert abs(dd1.kineticEnergy - one1.kineticEnergy) < 1e-4 * one1.kineticEnergy


def test_two_ranks_sharing_the_gpu_reproduce_the_single_gpu_run(tmp_path):
    from test_multirank_cpu import _run_dd_child
    out = _run_dd_child(tmp_path, False, 0, 10, 29561, env={"OPENMM_HIP_DD_DRIFT": "0.05"},
                        extra_cases='(("water, tile spreading", T.water_box(8, seed=5), 32), ("water, halo sections, half-shell", T.water_box(16, seed=5, cutoff=0.5), None), '
                                    '("water, halo, barostat", T.with_barostat(T.water_box(8, seed=5), 1.0, 300.0, 2, 11), 24))')
    print(out)


def test_two_ranks_over_rccl_on_two_gpus(tmp_path):
    """The real thing at its smallest: two processes, two GPUs, every collective of the decomposed step through RCCL (halo
    exchange of positions with the momentum trailers, the two all-to-alls of the slab FFT, the potential planes, the all-gathers
    of a State download) on two streams with two communicators -- against the single-GPU run of the same box, including the case
    whose ranges have all four sections.  Skips on a box with one GPU (the gpurun boxes); the pattern is the reference's
    multi-device test, platforms/cuda/tests/TestCudaNonbondedForce.cpp:37-96."""
    import ctypes
    from openmm_amd import capi
    count = ctypes.c_int(0)
    capi.load().device_count(ctypes.byref(count))        # (not torch.cuda: this process must not load a second HIP / RCCL user)
    if count.value < 2:
        pytest.skip("needs two GPUs (RCCL refuses two ranks on one device)")
    from test_multirank_cpu import _run_dd_child
    env = {"DD_TEST_TRANSPORT": "rccl", "OPENMM_HIP_DD_DRIFT": "0.05"}
    out = _run_dd_child(tmp_path, False, "rank", 10, 29581, env=env,
                        cases='(("water, halo", T.water_box(8, seed=5), 24), ("solvated chain, halo", T.small_solvated_chain(seed=3), 24), '
                              '("water, halo sections, half-shell", T.water_box(16, seed=5, cutoff=0.5), None))')
    print(out)


def test_three_ranks_sharing_the_gpu_in_a_triclinic_box(tmp_path):
    """Slabs of the first box fraction in a triclinic box, half-shell evaluation, a re-sort inside the run; tests/test_multirank_cpu.py runs
    the same on the emulator."""
    from test_multirank_cpu import _run_dd_child
    print(_run_dd_child(tmp_path, False, 0, 10, 29601, nproc=3, env={"OPENMM_HIP_DD_DRIFT": "0.03"},
                        cases='(("water, triclinic, halo sections, half-shell", T.sheared(T.water_box(12, seed=5, cutoff=0.5), 0.6, -0.5, 0.8), None), '
                              '("water, triclinic, halo sections, half-shell, larger", T.sheared(T.water_box(20, seed=5, cutoff=0.6), 1.5, -1.2, 2.0), None))'))
    print(_run_dd_child(tmp_path, False, 0, 6, 29605, nproc=3, env={"OPENMM_HIP_DD_DRIFT": "0.02"},
                        cases='(("solvated chain, triclinic, halo, half-shell", T.sheared(T.with_cutoff(T.small_solvated_chain(seed=3), 0.4), 0.5, -0.4, 0.6, affine=True), 48),)'))


def test_nearly_full_list_makes_the_ranks_resort_and_grow_it(tmp_path):
    """Two ranks sharing the GPU; the hook leaves the list 8 % of room at the third evaluation: level 3 in the trailer, a common re-sort (the
    only one of the run), the verified rebuild grows the allocation (tests/test_multirank_cpu.py runs the same on the emulator)."""
    from test_multirank_cpu import _run_dd_child
    print(_run_dd_child(tmp_path, False, 0, 40, 29611, env={"OPENMM_HIP_DD_DRIFT": "0.2", "OPENMM_HIP_REORDER_INTERVAL": "1000", "OPENMM_HIP_REORDER_LAG": "1000",
                                                           "OPENMM_HIP_DEBUG_TIGHT_LIST_AFTER": "3"},
                        cases='(("water, halo, tight list", T.water_box(8, seed=5), 24), ("water, halo sections, half-shell, tight list", T.water_box(16, seed=5, cutoff=0.5), None))'))


def test_three_ranks_sharing_the_gpu_half_shell_with_bonded_terms_across_boundaries(tmp_path):
    """Half-shell evaluation on the GPU: three ranks (sharing it, collectives over gloo), a chain with bonds / angles / torsions / 1-4s /
    exclusions across both inner slab boundaries -- pairs and terms evaluated once by the upper rank, the forces on the lower rank's atoms
    returned -- and the round-3 scheme (both sides) as the control; tests/test_multirank_cpu.py runs the same on the emulator."""
    from test_multirank_cpu import _run_dd_child
    chain = 'T.with_cutoff(T.small_solvated_chain(seed=3), 0.4)'
    print(_run_dd_child(tmp_path, False, 0, 6, 29591, nproc=3, env={"OPENMM_HIP_DD_DRIFT": "0.02"}, cases='(("solvated chain, halo, half-shell", %s, 48),)' % chain))
    print(_run_dd_child(tmp_path, False, 0, 6, 29595, nproc=3, env={"OPENMM_HIP_DD_DRIFT": "0.02", "OPENMM_HIP_DD_BOTH_SIDES": "1"}, cases='(("solvated chain, halo, both sides", %s, 48),)' % chain))


GOLDEN_CHILD = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np, torch.distributed as dist
from openmm_amd import harness as H, testsystems as T, multirank as MR
from openmm_amd.parity import force_parity
dist.init_process_group(backend="gloo")
rank, world = dist.get_rank(), dist.get_world_size()
H.load_hip_platform()
g = np.load(os.path.join(%r, "tests", "golden", "reference_forces_water985527_sample.npz"))
w = T.water_box(int(g["n_side"]), seed=int(g["seed"]))
w.pme_params = (float(g["pme"][0]), int(g["pme"][1]), int(g["pme"][2]), int(g["pme"][3]))
props = MR.domain_properties(dist, transport="gloo", device_index=0)
system, nb = w.build()
ctx = H.Context(system, H.Integrator(H.VERLET, 0.001), "HIP", props)
ctx.setPositions(w.positions)
st = ctx.getState(getForces=True, getEnergy=True)
info = H.domain_info()
ctx.close()
idx = g["indices"]
p = force_parity(w.positions, w.box, w.cutoff, st.forces[idx], g["forces"], subset=idx, rms=float(g["rms_force"]))
if rank == 0:
    print("water-1M on %%d ranks: force max-rel-err over the sampled atoms %%.3g (away from %%d edge pairs: %%.3g), E %%.3f vs %%.3f" %% (
        world, p["max_rel_err_all_atoms"], p["cutoff_edge_pairs"], p["max_rel_err"], st.potentialEnergy, float(g["energy"])), flush=True)
assert p["max_rel_err_all_atoms"] < 1e-4
assert info[1] == 1 and info[3] < world * info[2], ("slabs of 5.3 nm (4 ranks) or 2.7 nm (8): halo exchange expected", info)
assert info[7] > 1, ("half-shell evaluation expected", info)
if rank == 0:
    print("domain:", info, flush=True)
assert abs(st.potentialEnergy - float(g["energy"])) < 1e-5 * 5.0 * w.num_atoms
if rank == 0:
    print("OK")
dist.destroy_process_group()
'''


@pytest.mark.parametrize("world", [4, 8])
def test_million_atom_box_on_several_ranks_matches_the_reference_golden(tmp_path, world):
    """BASELINE.json configs[3] through the decomposed path: the 985 527-atom box on four and on eight ranks (sharing this GPU,
    collectives over gloo; eight 2.7 nm slabs are the decomposition the scaling run uses) against the sampled Reference-platform
    golden -- the same bar as the single-GPU test."""
    import subprocess
    import sys
    from conftest import ROOT
    script = tmp_path / "golden_child.py"
    script.write_text(GOLDEN_CHILD % (ROOT, ROOT))
    port = str(29571 + world)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
                          "--master-port", port, str(script)], capture_output=True, text=True, timeout=1500, env=env)
    assert "OK" in out.stdout, out.stdout[-3000:] + out.stderr[-4000:]
    print(out.stdout)


def test_million_atom_box_over_a_device_list_matches_the_reference_golden():
    """The same golden through the reference's OWN multi-device interface: ONE Context, `DeviceIndex = "0,0,0,0,0,0,0,0"` -- eight ranks on
    threads of this process (platform/HipParallel.h; a device named twice meets through the host-staged all-gather between the threads)
    -- and, after the forces, twenty LangevinMiddle steps with the state queries a user makes."""
    from openmm_amd.parity import force_parity
    from conftest import ROOT
    H.load_hip_platform()
    g = np.load(os.path.join(ROOT, "tests", "golden", "reference_forces_water985527_sample.npz"))
    w = T.water_box(int(g["n_side"]), seed=int(g["seed"]))
    w.pme_params = (float(g["pme"][0]), int(g["pme"][1]), int(g["pme"][2]), int(g["pme"][3]))
    system, nb = w.build()
    integ = H.Integrator(H.LANGEVIN_MIDDLE, 0.002, 300.0, 1.0, seed=3, constraintTolerance=1e-5)
    ctx = H.Context(system, integ, "HIP", {"DeviceIndex": ",".join(["0"] * 8)})
    assert ctx.getPlatformProperty("Ranks") == "8" and ctx.getPlatformProperty("DeviceIndex") == "0,0,0,0,0,0,0,0"
    ctx.setPositions(w.positions)
    st = ctx.getState(getForces=True, getEnergy=True)
    idx = g["indices"]
    p = force_parity(w.positions, w.box, w.cutoff, st.forces[idx], g["forces"], subset=idx, rms=float(g["rms_force"]))
    print("water-1M over a list of 8 devices: force max-rel-err over the sampled atoms %.3g, E %.3f vs %.3f" % (p["max_rel_err_all_atoms"], st.potentialEnergy, float(g["energy"])))
    assert p["max_rel_err_all_atoms"] < 1e-4
    assert abs(st.potentialEnergy - float(g["energy"])) < 1e-5 * 5.0 * w.num_atoms
    ctx.applyConstraints(1e-5)
    ctx.setVelocitiesToTemperature(300.0, 1)
    integ.step(20)
    after = ctx.getState(getPositions=True, getEnergy=True)
    assert np.all(np.isfinite(after.positions)) and np.isfinite(after.potentialEnergy)
    pos = after.positions.reshape(-1, 3, 3)
    assert np.abs(np.linalg.norm(pos[:, 0] - pos[:, 1], axis=1) - T.TIP3P["dOH"]).max() < 1e-4
    ctx.close()
    # the same twenty steps on ONE device (same thermostat seed: the noise is keyed by seed, step and atom): the same energies
    system1, nb1 = w.build()
    integ1 = H.Integrator(H.LANGEVIN_MIDDLE, 0.002, 300.0, 1.0, seed=3, constraintTolerance=1e-5)
    ctx1 = H.Context(system1, integ1, "HIP", {"DeviceIndex": "0"})
    ctx1.setPositions(w.positions)
    ctx1.applyConstraints(1e-5)
    ctx1.setVelocitiesToTemperature(300.0, 1)
    integ1.step(20)
    single = ctx1.getState(getPositions=True, getEnergy=True)
    ctx1.close()
    print("after 20 steps: E_pot %.3f (list of 8) vs %.3f (one device), E_kin %.3f vs %.3f, largest position difference %.2e nm" % (
        after.potentialEnergy, single.potentialEnergy, after.kineticEnergy, single.kineticEnergy, np.abs(after.positions - single.positions).max()))
    assert abs(after.potentialEnergy - single.potentialEnergy) < 1e-5 * abs(single.potentialEnergy)
    assert abs(after.kineticEnergy - single.kineticEnergy) < 1e-4 * abs(single.kineticEnergy)
    assert np.abs(after.positions - single.positions).max() < 1e-4


def test_bench_rank_alone_reports_every_rank_of_a_decomposition(tmp_path):
    """bench.py --rank-alone R (diagnostics: each rank of an R-rank decomposition alone on the GPU, collectives that cost nothing -- the compute side
    of the scaling limit): runs, and reports a time per rank; the transport is refused unless the process asks for it."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--rank-alone", "2", "--workload", "water24k", "--steps", "30", "--warmup", "10"],
                         capture_output=True, text=True, timeout=600)
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert line, out.stdout[-1000:] + out.stderr[-2000:]
    d = json.loads(line[-1])
    assert len(d["per_rank_alone_ms_per_step"]) == 2 and all(0.0 < t < 5.0 for t in d["per_rank_alone_ms_per_step"]), d
    w = T.water_box(6, seed=3)
    system, nb = w.build()
    with pytest.raises(H.OpenMMError):
        H.Context(system, H.Integrator(H.VERLET, 0.001), "HIP", {"Ranks": "2", "Rank": "0", "CommId": "alone"})


def test_one_rank_over_rccl_walks_every_collective_of_the_decomposed_step():
    """Every collective of include/openmm_hip_comm.h through librccl with a one-rank communicator, driven through the C ABI with the buffers
    of a decomposed step in miniature (VERDICT r4 "missing" 2: the first 8-GPU run must not be the first execution of these calls):
    ncclCommSplit (ommhip_comm_duplicate) and traffic on the split communicator from a second stream while the first is busy; the in-place
    all-gather; the all-to-all and the ring exchange of the slab FFT (grouped ncclSend / ncclRecv with the rank itself, distinct buffers:
    the data really travels); the halo exchange's group -- both boundary sections, in place; the halo return's group of three sends and
    three receives into the staging buffer followed by the integer add (the rank is its own upper neighbour: its section doubles, every
    other slot stays); the host all-gather of small records.  What one GPU cannot show -- two peers, xGMI -- is
    test_two_ranks_over_rccl_on_two_gpus."""
    import ctypes as C
    from openmm_amd import capi
    K = capi.load()
    lib = K.lib
    ident = MR.new_rccl_id()
    comm, side = C.c_void_p(), C.c_void_p()
    assert lib.ommhip_comm_create_rccl(ident.encode(), 0, 1, C.byref(comm)) == 0
    lib.ommhip_comm_transport.restype = C.c_char_p
    assert lib.ommhip_comm_transport(comm) == b"rccl"
    assert lib.ommhip_comm_duplicate(comm, C.byref(side)) == 0                    # ncclCommSplit
    assert lib.ommhip_comm_size(side) == 1 and lib.ommhip_comm_rank(side) == 0
    main_stream, side_stream = C.c_void_p(), C.c_void_p()
    K.stream_create(C.byref(main_stream)); K.stream_create(C.byref(side_stream))
    rng = np.random.default_rng(5)
    sz = C.c_size_t
    # ---- all-to-all + ring exchange on the split communicator / side stream, all-gather and halo traffic on the main one, interleaved
    n = 1 << 18
    a2a_send = rng.integers(0, 255, n, dtype=np.uint8)
    d_send, d_recv = K.upload(a2a_send), K.upload(np.zeros(n, np.uint8))
    assert lib.ommhip_comm_all_to_all(side, d_send, d_recv, sz(n), side_stream) == 0
    planes_down, planes_up = rng.normal(size=5000).astype(np.float32), rng.normal(size=3000).astype(np.float32)
    d_down, d_up = K.upload(planes_down), K.upload(planes_up)
    d_from_up, d_from_down = K.upload(np.zeros(5000, np.float32)), K.upload(np.zeros(3000, np.float32))
    assert lib.ommhip_comm_ring_exchange(side, d_down, d_from_up, sz(planes_down.nbytes), d_up, d_from_down, sz(planes_up.nbytes), side_stream) == 0
    # positions: in-place all-gather (replicated fallback) ...
    slots = 4096
    wire = rng.integers(0, 2 ** 32, (slots, 4), dtype=np.uint32)
    d_wire = K.upload(wire)
    assert lib.ommhip_comm_all_gather(comm, d_wire, sz(wire.nbytes), main_stream) == 0
    # ... and the halo exchange: down section = blocks [0, 40), up section = blocks [88, 128) of this rank's 128 blocks, trailer at the end
    class HaloPlan(C.Structure):
        _fields_ = [("rank_stride", sz), ("down_offset", sz * 64), ("down_bytes", sz * 64), ("up_offset", sz * 64), ("up_bytes", sz * 64), ("trailer_offset", sz), ("trailer_bytes", sz)]
    plan = HaloPlan()
    plan.rank_stride = wire.nbytes
    plan.down_offset[0], plan.down_bytes[0] = 0, 40 * 32 * 16
    plan.up_offset[0], plan.up_bytes[0] = 88 * 32 * 16, 40 * 32 * 16
    plan.trailer_offset, plan.trailer_bytes = (slots - 2) * 16, 32
    assert lib.ommhip_comm_halo_exchange(comm, d_wire, C.byref(plan), main_stream) == 0
    # forces: the halo return (half-shell evaluation) -- section [1024, 1024 + 700) of a 4096-slot SoA buffer
    class ReturnPlan(C.Structure):
        _fields_ = [("first_slot", C.c_int * 64), ("num_slots", C.c_int * 64)]
    rplan = ReturnPlan()
    rplan.first_slot[0], rplan.num_slots[0] = 1024, 700
    force = rng.integers(-2 ** 40, 2 ** 40, 3 * slots, dtype=np.int64)
    d_force, d_staging = K.upload(force), K.upload(np.zeros(3 * 700, np.int64))
    assert lib.ommhip_comm_halo_return(comm, d_force, slots, C.byref(rplan), d_staging, main_stream) == 0
    rec_in, rec_out = np.array([3.25, -1.5]), np.zeros(2)
    assert lib.ommhip_comm_all_gather_host(comm, rec_in.ctypes.data_as(C.c_void_p), rec_out.ctypes.data_as(C.c_void_p), sz(16), main_stream) == 0
    K.stream_sync(main_stream); K.stream_sync(side_stream)
    assert np.array_equal(K.download(d_recv, (n,), np.uint8), a2a_send)
    assert np.array_equal(K.download(d_from_up, (5000,), np.float32), planes_down)          # what went down comes back from above
    assert np.array_equal(K.download(d_from_down, (3000,), np.float32), planes_up)
    assert np.array_equal(K.download(d_wire, (slots, 4), np.uint32), wire)                  # sections landed on themselves
    assert np.array_equal(K.download(d_staging, (3, 700), np.int64), force.reshape(3, slots)[:, 1024:1724])      # the bytes went through ncclSend / ncclRecv
    expect = force.reshape(3, slots).copy()
    expect[:, 1024:1724] *= 2
    assert np.array_equal(K.download(d_force, (3, slots), np.int64), expect)
    assert np.array_equal(rec_out, rec_in)
    assert lib.ommhip_comm_destroy(side) == 0 and lib.ommhip_comm_destroy(comm) == 0


def test_bench_launcher_flow_with_two_ranks(tmp_path):
    """`bench.py --gpus 2` started the way the driver starts it (torch.distributed.run, no extra flags: the roofline section runs): on a box
    with one GPU both RCCL configurations end at once ("rank 1 has no GPU"), the launchers move to the host-staged transport together (two
    ranks on GPU 0), the 985 527-atom box is stepped on two ranks and rank 0 prints ONE JSON line.  Rounds 4 - 5 hung here -- rank 0 took four
    probe steps of the decomposed run alone in its roofline section -- and no test ran this flow on a GPU without --no-roofline."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    proc = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                           "--master-port", "29721", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5", "--attempt-timeout", "200"],
                          capture_output=True, text=True, timeout=1200, env=env, cwd=ROOT)
    assert proc.returncode == 0, proc.stdout[-2000:] + proc.stderr[-4000:]
    out = json.loads(proc.stdout.strip().splitlines()[-1])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["steps"] == 20 and out["value"] > 0
    assert out["single_gpu_same_box"]["value"] > out["value"] * 0.1
    assert out["single_gpu_same_box"]["initial_energy_kj_mol"]["rel_diff"] < 1e-5
    print(out["value"], out["ms_per_step"], out["config"]["workload"][-200:])
