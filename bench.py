#!/usr/bin/env python
"""bench.py -- ns/day of the MD hot path (NonbondedForce direct space + PME + LangevinMiddle/SETTLE) on the
OpenMM "HIP" platform, one process per GPU.

    python bench.py --gpus 1 --steps 3000 --warmup 300
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

A "step" is one MD step = one pass of the hot path.  Timing: warm-up steps, then EXACTLY K steps between two (barrier + device
synchronisation) brackets, MAX over ranks.  The synchronisation is ommhip_device_sync (PyTorch's wheels bundle a HIP runtime of their
own: torch.cuda.synchronize() does not see the plugin's streams).  examples/benchmark.py:9-18 instead ends its timed region with
getState(energy) -- one more evaluation plus a host round trip; that query runs behind the closing bracket here and the line reports
the figure with it counted as well (`closing_energy_query`; until round 5 `value` itself counted it).

N = 1 (the headline, BASELINE.json configs[1], examples/benchmark.py `pme`): DHFR -- the 23 558 atoms of
examples/5dfr_solv-cube_equil.pdb with amber99sb + tip3p parameters (openmm_amd/forcefield.py, fixture under tests/golden/),
6.223 nm cube, PME, cutoff 0.9 nm, Ewald tolerance 5e-4 (alpha 2.92/nm, grid 56^3), LangevinMiddleIntegrator 300 K, 1/ps,
HBonds constraints + rigid water, CMMotionRemover, dt 2 fs (--dt-fs 4 for the script's own step size).  The JSON line also
carries `scale_workload`: the single-GPU ns/day of the water-1M box below, measured in the same run, i.e. the N = 1 point
of the strong-scaling curve.

N > 1 (BASELINE.json configs[3]): ONE 985 527-atom TIP3P box (21.4 nm, PME grid 192^3) domain-decomposed over the N GPUs
(x slabs, positions all-gathered over RCCL every step, slab FFT with two all-to-alls) -- "scaling": "strong",
`value` = ns/day of that one simulation.  (A 23 558-atom system does not shard usefully over 8 GPUs: its halo is several
times its slab.)  --workload overrides either default.  The line also carries `single_gpu_same_box`: rank 0 runs the same box
on its GPU alone after the timed region (the other ranks wait), so every N > 1 line holds its own N = 1 reference.
The process torch.distributed.run starts on each rank is only a launcher: the measurement runs in a child process
(multirank.run_attempts), first with RCCL and reciprocal space on its own stream + communicator, then -- only if that
attempt fails or never returns on some rank -- with RCCL on a single stream and ncclAllGather instead of direct sends, and
last with host-staged gloo collectives; the
configuration that ran is named in config.workload and the failed attempts are listed in config.attempts_failed.

Rank 0 prints ONE JSON line with the contract fields plus `roofline` (the direct-space pair kernel -- on the fused
single-stream path the three launches it shares with the FFT stages -- HIP events on the stream the kernels run on),
`roofline_fft` (the 3-D FFT chain of reciprocal space on its own), `cpu_baseline` (the reference's own platforms/cpu built
into oracle/_ref, timed on this host on a bounded number of steps of the same System) and `force_parity` (HIP forces of the
final configuration against the reference's Reference platform).  DESIGN.md (d) defines every field.
"""
import argparse
import ctypes as C
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# multi-process GPU work on this stack needs dmabuf IPC (RCCL's hipIpcGetMemHandle fails otherwise); the GPU boxes export it -- kept if set
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBPS = 8000.0    # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP32_VECTOR_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: FP32 vector (non-matrix) peak
PAIR_FLOP = 45            # flop per evaluated pair (F_dir = 1024 T 45)
# Sources the PMC-profiled launches (pairs_fft_plane / pairs_fft_lines: pair kernel + FFT stages) are compiled from; profiles/pmc_pairs_fft.json
# is quoted only while their hash matches.  (neighbor.hip -- the list BUILDER -- was part of the set until the end of round 2; a change
# of the list FORMAT shows up in nonbonded.hip, which reads it.)
KERNEL_SOURCES = ("nonbonded.hip", "force_front.hip", "pme.hip", "common.h")
WORKLOADS = ["dhfr", "dhfr_like", "water1k", "water24k", "water98k", "apoa1", "water1m", "water1m_lattice"]
LATTICE_PREPARE_STEPS = 1000     # untimed relaxation of a generated (jittered-lattice) water box before warm-up and timing
EMULATED = os.environ.get("BENCH_EMULATED") == "1"     # tests only: the CPU SIMT emulator build of the plugin (tests/emu), to run the N > 1 flow without a GPU


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=3000)
    p.add_argument("--warmup", type=int, default=300)
    p.add_argument("--dt-fs", type=float, default=2.0)
    p.add_argument("--workload", default="auto", choices=["auto"] + WORKLOADS, help="auto = dhfr on one GPU, water1m (one box, decomposed) on several")
    p.add_argument("--cpu-steps", type=int, default=150, help="steps of the CPU-platform baseline (0 disables it and the force-parity check)")
    p.add_argument("--no-roofline", action="store_true")
    p.add_argument("--no-scale-workload", action="store_true", help="N = 1: skip the single-GPU run of the strong-scaling workload")
    p.add_argument("--prepare-steps", type=int, default=-1, help="untimed steps that relax a lattice start before warm-up (input preparation; default 1000 for the water boxes -- after 200 steps a jittered lattice is still melting: 9 %% more list rows and 40 %% more list rebuilds per step than after 3000, profiles/r04h_prepare_steps_water1m.txt -- 0 for fixtures)")
    p.add_argument("--transport", default="rccl", choices=["rccl", "gloo"], help="collectives of the decomposed run: RCCL (product) or host-staged gloo (rehearsal on one GPU)")
    p.add_argument("--profile-every", type=int, default=0, help="HIP-event timing of every n-th launch of each profiled kernel inside the timed region (0 = 7: a 20-step run then holds 3 samples)")
    p.add_argument("--no-pmc", action="store_true", help="N = 1: do not spawn the rocprofv3 child runs (FETCH_SIZE / WRITE_SIZE passes of the dominant launch group, kernel trace of the amoeba_dhfr workload) after the timed region")
    p.add_argument("--no-extra-workloads", action="store_true", help="N = 1: skip the short runs of BASELINE.json configs[2] (apoa1-sized) and of the benchmark script's own 4 fs step")
    p.add_argument("--decompose", action="store_true", help="N = 1: run through the decomposed path with a one-rank RCCL communicator (overhead check on one GPU)")
    p.add_argument("--props", default="", help="extra HIP platform properties, e.g. DisablePmeStream=true")
    p.add_argument("--serialize-ranks", action="store_true", help="diagnostics, gloo transport on one GPU: the ranks run their work between collectives one at a time, so each rank's compute time per step is measured on an idle GPU (per_rank_compute_ms_per_step); the wall-clock value is meaningless in this mode")
    p.add_argument("--amoeba-timeout", type=float, default=420.0, help="N = 1: seconds the child process with the two AMOEBA legs (tools/bench_amoeba_legs.py) may take")
    p.add_argument("--rank-alone", type=int, default=0, metavar="R", help="diagnostics on ONE GPU: every rank of an R-rank decomposition of the 1M-atom box in turn, ALONE -- collectives that cost "
                   "nothing and move nothing (CommId \"alone\"), dynamics frozen (1e-3 fs), a list rebuild forced every 8th step: a rank's step with its two streams overlapped and no communication "
                   "(per_rank_alone_ms_per_step); the forces are meaningless")
    p.add_argument("--attempt-timeout", type=float, default=200.0, help="N > 1: seconds a configuration may take before the launchers give up on it")
    return p.parse_args()


def supervise(args, rank, world):
    """N > 1, the process torch.distributed.run started: run the measurement in a child and fall back together (see the module text)."""
    from openmm_amd import multirank as MR
    base = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]
    extra = [kv for kv in args.props.split(",") if kv]
    attempts = []
    if args.transport == "rccl":
        attempts.append(base + ["--transport", "rccl"])
        if not any(kv.startswith("DisablePmeStream=") for kv in extra):
            # the most conservative RCCL configuration: one stream, one communicator, positions replicated by ncclAllGather instead of the
            # halo exchange's grouped sends and receives
            attempts.append((base + ["--transport", "rccl", "--props", ",".join(extra + ["DisablePmeStream=true"])], {"OPENMM_HIP_ALLGATHER": "ring", "OPENMM_HIP_DD_REPLICATE": "1"}))
    attempts.append(base + ["--transport", "gloo"])
    if args.serialize_ranks:
        os.environ["OMMHIP_COMM_DIAG"] = "1"          # inherited by the children
    say = lambda msg: print("bench.py launcher (rank %d): %s" % (rank, msg), file=sys.stderr, flush=True)
    idx, lines, notes = MR.run_attempts(attempts, rank, world, os.environ.get("MASTER_ADDR", "127.0.0.1"), int(os.environ.get("MASTER_PORT", "29500")),
                                        args.attempt_timeout, log=say if rank == 0 else None)
    if rank != 0:
        return
    result = [l for l in lines if l.startswith("{")]
    if not result:
        raise RuntimeError("the measurement finished without a result line")
    out = json.loads(result[-1])
    out["config"]["attempts_failed"] = notes
    for l in lines:
        if not l.startswith("{"):
            print(l)
    print(json.dumps(out), flush=True)


def make_workload(name, seed):
    from openmm_amd import testsystems as T
    if name == "dhfr":
        return T.dhfr()                          # the real benchmark System (5dfr_solv-cube_equil.pdb, amber99sb + tip3p)
    if name == "dhfr_like":
        return T.dhfr_like(seed=seed)            # round-1 stand-in: same size, synthetic chain
    if name == "water1k":
        return T.water_box(8, seed=seed)         # 1536 atoms: flow tests on the emulator
    if name == "water24k":
        return T.water_box(20, seed=seed)
    if name == "apoa1":
        return T.apoa1_like(seed=seed)           # 92 224 atoms in the apoa1 box (BASELINE.json configs[2] stand-in)
    if name == "water1m":
        return T.water_tiled(3)                  # 985 527 atoms, L = 21.4 nm (BASELINE.json configs[3]): 27 copies of an equilibrated tile
    if name == "water1m_lattice":
        return T.water_box(69, seed=seed)        # the same size as a jittered lattice (rounds 1-3 measured this: it melts at ~900 K)
    return T.water_box(32, seed=seed)


def default_prepare(w):
    """untimed steps before warm-up: none for an equilibrated fixture (`prepare_steps` of a tiled one: its copies decorrelate), the
    relaxation of a generated lattice otherwise"""
    return getattr(w, "prepare_steps", 0) if getattr(w, "velocities", None) is not None else LATTICE_PREPARE_STEPS


def start_platform(w, platform, dt_ps, warmup, props=None, seed=1, prepare=0):
    """Context on `platform`, positions/velocities set, `prepare` + `warmup` untimed steps done.  -> (system, nb, integrator, context)"""
    from openmm_amd import harness as H
    system, nb = w.build()
    integ = H.Integrator(H.LANGEVIN_MIDDLE, dt_ps, 300.0, 1.0, seed=seed, constraintTolerance=1e-5)
    ctx = H.Context(system, integ, platform, props)
    ctx.setPositions(w.positions)
    ctx.applyConstraints(1e-5)
    if getattr(w, "velocities", None) is not None:
        ctx.setVelocities(w.velocities)          # equilibrated start (tests/golden fixture)
    else:
        ctx.setVelocitiesToTemperature(300.0, 1)
    ctx.initial_potential_energy = ctx.getState(getEnergy=True).potentialEnergy      # same start on every platform / decomposition: a parity handle
    integ.step(prepare + warmup)
    ctx.getState(getEnergy=True)
    return system, nb, integ, ctx


def kernel_sources_sha():
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        with open(os.path.join(ROOT, "openmm_amd", "csrc", "kernels", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def collect_timers(kernels):
    calls, total_ms = C.c_longlong(), C.c_double()
    timers = {}
    for name, idx in (("nb_direct", 0), ("nl_update", 1), ("pme_spread", 2), ("pme_fft", 3), ("pme_interpolate", 4),
                      ("pairs_fft_stage0", 5), ("pairs_fft_stage1", 6), ("pairs_fft_stage2", 7)):
        kernels.lib.ommhip_profile_collect(idx, C.byref(calls), C.byref(total_ms))
        timers[name] = {"calls": calls.value, "avg_us": (1e3 * total_ms.value / calls.value) if calls.value else None}
    return timers


def kernel_rooflines(kernels, plugin, integ, ctx, num_atoms, grid, cutoff_pairs=None, steps=40):
    """Rooflines of the step's kernels other than the 3-D FFT on a Context whose reciprocal space runs on the MAIN stream (every kernel on its
    own, HIP events around each launch): SURVEY.md 8(d)'s algorithmic bytes (B_dir, B_spr, B_int, B_nl -- the list rebuild per REBUILD, its
    launches that found nothing to do left out) over the measured durations, and the pair kernel's FP32 issue fraction.  -> dict"""
    before = (C.c_longlong * 8)()
    plugin.ommhip_plugin_nl_stats(before)
    kernels.lib.ommhip_profile_reset()
    kernels.lib.ommhip_profile_enable_timers(1, 0x1f, 4 * steps)
    integ.step(steps)
    ctx.getState(getEnergy=True)
    kernels.lib.ommhip_profile_enable(0)
    t = collect_timers(kernels)
    after = (C.c_longlong * 8)()
    plugin.ommhip_plugin_nl_stats(after)
    rows, rebuilds = int(after[3]), int(after[5] - before[5])
    n, g, p = num_atoms, grid[0] * grid[1] * grid[2], 125

    def obj(kernel, bytes_, us, extra=None):
        if not us:
            return {"kernel": kernel, "avg_us": None}
        a = bytes_ / (us * 1e-6) / 1e9
        o = {"bound": "hbm", "kernel": kernel, "algorithmic_bytes": int(bytes_), "avg_us": round(us, 2), "achieved": round(a, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(a / HBM_PEAK_GBPS, 5)}
        o.update(extra or {})
        return o
    out = {"source": "%d steps after the timed region on a Context with reciprocal space on the main stream (DisablePmeStream=true), HIP events around every launch; bytes: SURVEY.md 8(d)" % steps}
    pair_us = t["nb_direct"]["avg_us"]
    out["direct"] = obj("nb_direct", 48 * n + 52 * 32 * (2 * rows), pair_us, {"rows": rows, "formula": "B_dir = 48 N + 52*32*T, T = 2 tiles per 64-slot row"})
    if pair_us:
        evals = rows * 64 * 32
        fi = {"bound": "fp32 vector issue", "peak": FP32_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s", "flop_per_pair": PAIR_FLOP, "pair_evals_per_launch": evals,
              "achieved": round(evals * PAIR_FLOP / (pair_us * 1e-6) / 1e12, 3)}
        fi["frac"] = round(fi["achieved"] / FP32_VECTOR_PEAK_TFLOPS, 5)
        if cutoff_pairs:
            fi["pairs_inside_cutoff"] = int(cutoff_pairs)
            fi["evals_per_useful_pair"] = round(evals / max(cutoff_pairs, 1), 3)
            fi["useful_frac"] = round(cutoff_pairs * PAIR_FLOP / (pair_us * 1e-6) / 1e12 / FP32_VECTOR_PEAK_TFLOPS, 5)
        out["direct"]["fp32_issue"] = fi
    out["spread"] = obj("pme_spread (LDS bricks, coalesced atomics)", 16 * n + 4 * p * n * 2 + 4 * g, t["pme_spread"]["avg_us"], {"formula": "B_spr = 16 N + 4 P N 2 (RMW) + 4 G (clear)"})
    out["interpolate"] = obj("pme_interpolate", 16 * n + 4 * p * n + 24 * n, t["pme_interpolate"]["avg_us"], {"formula": "B_int = 16 N + 4 P N + 24 N"})
    nl_us = t["nl_update"]["avg_us"]
    if nl_us and rebuilds > 0 and t["nl_update"]["calls"] > 0:
        # the timer brackets the rebuild launches of EVERY step (they leave at once when no rebuild is due): time per rebuild = the sum over the calls
        per_rebuild = nl_us * t["nl_update"]["calls"] / rebuilds
        out["list_rebuild"] = obj("nl_bin_blocks + nl_find_interactions (per rebuild)", 16 * n + 32 * (n // 32) + 4 * 32 * (2 * rows), per_rebuild,
                                  {"rebuilds": rebuilds, "steps": steps, "formula": "B_nl = 16 N + 32 Nblk + 4*32*T_list per rebuild"})
    return out


DEVICE_SYNC = [lambda: None]        # set by main(): ommhip_device_sync on this rank's GPU (torch.cuda.synchronize() belongs to PyTorch's own HIP runtime and does not see the plugin's streams)
CLOSING_QUERY_S = []                # seconds of the energy query behind every timed region, in call order


def timed_run(integ, ctx, steps, barrier):
    """EXACTLY `steps` steps between two (barrier + device synchronisation) brackets.  examples/benchmark.py also counts the
    getState(getEnergy=True) it ends with -- one more evaluation with energies and a host round trip; here that query runs behind the
    closing bracket, timed on its own (CLOSING_QUERY_S; the line reports the headline figure both ways)."""
    barrier()
    DEVICE_SYNC[0]()
    t0 = time.perf_counter()
    integ.step(steps)
    DEVICE_SYNC[0]()
    elapsed = time.perf_counter() - t0
    st = ctx.getState(getEnergy=True)
    CLOSING_QUERY_S.append(time.perf_counter() - t0 - elapsed)
    barrier()
    return elapsed, st


from openmm_amd.profiling import rocprof_child  # noqa: E402


def measure_group_traffic():
    """HBM-side bytes of the dominant launch group (pairs_fft_plane x 2 + pairs_fft_lines) of THIS build on THIS box: two rocprofv3 child runs of
    the default workload (FETCH_SIZE and WRITE_SIZE cannot share a pass), 40 + 10 steps each.  FETCH_SIZE is doubled (gfx950: it tallies the
    128-byte requests of wide streaming reads at 64 bytes, MI355X_MICROARCH.md "HBM"); WRITE_SIZE is taken as reported (uncalibrated); both
    are in KB.  -> dict for roofline.traffic / traffic_source."""
    child = [sys.executable, os.path.abspath(__file__), "--steps", "40", "--warmup", "10", "--cpu-steps", "0", "--no-roofline", "--no-scale-workload", "--no-extra-workloads", "--no-pmc"]
    kb = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        per, _ = rocprof_child(child, pmc=counter)
        for key in ("pairs_fft_plane", "pairs_fft_lines"):
            sub = per[per["Kernel_Name"].str.contains(key)]
            if len(sub) == 0:
                raise RuntimeError("no dispatch of %s in the %s pass" % (key, counter))
            kb[(counter, key)] = (float(sub["value"].mean()), int(len(sub)), float(sub["dur_us"].mean()))
    fetch = 2.0 * (2 * kb[("FETCH_SIZE", "pairs_fft_plane")][0] + kb[("FETCH_SIZE", "pairs_fft_lines")][0])
    write = 2 * kb[("WRITE_SIZE", "pairs_fft_plane")][0] + kb[("WRITE_SIZE", "pairs_fft_lines")][0]
    return {"traffic": int(1024.0 * (fetch + write)),
            "detail": {"fetch_size_kb_per_launch": {"pairs_fft_plane (x2)": round(kb[("FETCH_SIZE", "pairs_fft_plane")][0], 1), "pairs_fft_lines": round(kb[("FETCH_SIZE", "pairs_fft_lines")][0], 1)},
                       "write_size_kb_per_launch": {"pairs_fft_plane (x2)": round(kb[("WRITE_SIZE", "pairs_fft_plane")][0], 1), "pairs_fft_lines": round(kb[("WRITE_SIZE", "pairs_fft_lines")][0], 1)},
                       "dispatches_averaged": {"pairs_fft_plane": kb[("FETCH_SIZE", "pairs_fft_plane")][1], "pairs_fft_lines": kb[("FETCH_SIZE", "pairs_fft_lines")][1]},
                       "kernel_us_under_the_profiler": {"pairs_fft_plane": round(kb[("FETCH_SIZE", "pairs_fft_plane")][2], 2), "pairs_fft_lines": round(kb[("FETCH_SIZE", "pairs_fft_lines")][2], 2)},
                       "correction": "FETCH_SIZE x 2 (gfx950, MI355X_MICROARCH.md HBM section); WRITE_SIZE as reported; KB -> bytes x 1024"}}


def rank_alone(args, H, np):
    """--rank-alone R: see the option's help.  One JSON line."""
    os.environ["OPENMM_HIP_ALLOW_ALONE_COMM"] = "1"
    os.environ.setdefault("OPENMM_HIP_DEBUG_REBUILD_EVERY", "8")
    ranks = args.rank_alone
    w = make_workload(args.workload if args.workload != "auto" else "water1m", seed=1)
    per_rank = []
    for r in range(ranks):
        system, nb = w.build()
        integ = H.Integrator(H.LANGEVIN_MIDDLE, 1e-6, 300.0, 1.0, seed=1, constraintTolerance=1e-5)          # 1e-3 fs: nothing moves, the halo stays what it was given
        ctx = H.Context(system, integ, "HIP", {"DeviceIndex": "0", "Ranks": str(ranks), "Rank": str(r), "CommId": "alone"})
        ctx.setPositions(w.positions)
        ctx.applyConstraints(1e-5)
        if getattr(w, "velocities", None) is not None:
            ctx.setVelocities(w.velocities)
        integ.step(args.warmup if args.warmup < 300 else 40)
        ctx.getState(getEnergy=True)
        steps = args.steps if args.steps < 3000 else 400
        t0 = time.perf_counter()
        integ.step(steps)
        ctx.getState(getEnergy=True)
        per_rank.append(round(1e3 * (time.perf_counter() - t0) / steps, 4))
        info = None
        try:
            info = H.domain_info()
        except Exception:
            pass
        ctx.close()
    out = {"per_rank_alone_ms_per_step": per_rank, "ranks": ranks, "steps": steps, "workload": w.name,
           "rebuild_every": int(os.environ["OPENMM_HIP_DEBUG_REBUILD_EVERY"]),
           "domain_of_the_last_rank": list(info) if info is not None else None,
           "note": "every rank of the decomposition ALONE on one GPU: collectives return at once and move nothing, dynamics frozen (1e-3 fs), a list rebuild forced "
                   "every rebuild_every-th step; the two streams of the rank overlap as they do over RCCL.  The compute side of the scaling limit: forces are meaningless"}
    print(json.dumps(out), flush=True)


def main():
    args = parse_args()
    if os.environ.get("BENCH_DEBUG_HANG"):
        # diagnostics: after that many seconds every thread's Python stack goes to stderr and the process ends (where does a run that never returns sit?)
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["BENCH_DEBUG_HANG"]), exit=True)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and os.environ.get("BENCH_CHILD") != "1":
        return supervise(args, rank, world)
    dist = None
    if world > 1:
        # control plane (communicator id, barriers, the MAX of the timings): gloo.  The data path is the plugin's own RCCL
        # communicator(s); keeping torch's NCCL process group out of the process leaves one RCCL user per GPU.
        import torch
        import torch.distributed as dist
        ndev = torch.cuda.device_count()
        if local_rank >= ndev and not EMULATED:
            if args.transport != "gloo":
                raise RuntimeError("rank %d has no GPU (%d visible)" % (local_rank, ndev))
            local_rank = local_rank % max(ndev, 1)       # rehearsal of the N > 1 flow on a box with fewer GPUs (host-staged transport only)
        if EMULATED:
            local_rank = 0
        else:
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="gloo")

    import numpy as np
    from openmm_amd import capi, harness as H, multirank as MR
    H.load_hip_platform(emulated=EMULATED)
    if args.rank_alone > 1 and world == 1:
        return rank_alone(args, H, np)
    kernels = capi.load(os.path.join(H.EMU_DIR, "libopenmm_hip_kernels.so")) if EMULATED else capi.load()

    def device_sync():
        rc = kernels.lib.ommhip_device_sync(local_rank)
        if rc != 0:
            raise RuntimeError("ommhip_device_sync(%d) failed: %d" % (local_rank, rc))
    DEVICE_SYNC[0] = device_sync
    plugin = C.CDLL(os.path.join(H.EMU_DIR if EMULATED else H.LIB_DIR, "libOpenMMHIP.so"))

    workload = args.workload if args.workload != "auto" else ("dhfr" if world == 1 else "water1m")
    decomposed = world > 1
    dt_ps = args.dt_fs * 1e-3
    w = make_workload(workload, seed=1)
    prepare = args.prepare_steps if args.prepare_steps >= 0 else default_prepare(w)
    props = {"DeviceIndex": str(local_rank)}
    for kv in filter(None, args.props.split(",")):
        k, v = kv.split("=")
        props[k] = v
    transport = None
    if args.decompose and world == 1:
        props.update({"Ranks": "1", "Rank": "0", "CommId": MR.new_rccl_id()})
    if decomposed:
        # ONE box over all ranks.  The plugin runs its own collectives (RCCL); the launcher only distributes the communicator id.
        # a failure here (or a collective that never returns) ends this child; the launchers then move to the next configuration
        transport = args.transport
        props.update(MR.domain_properties(dist, transport=transport, emulated=EMULATED, serialize=args.serialize_ranks and transport == "gloo"))
    system, nb, integ, ctx = start_platform(w, "HIP", dt_ps, args.warmup, props, seed=1, prepare=prepare)
    device_name = ctx.getPlatformProperty("DeviceName")
    integration_mode, fallback_forces = ctx.getPlatformProperty("IntegrationMode"), ctx.getPlatformProperty("FallbackForces")
    if integration_mode != "device" or fallback_forces != "none":
        raise RuntimeError("the timed Context does not run natively: IntegrationMode %r, FallbackForces %r" % (integration_mode, fallback_forces))
    e0_run = ctx.initial_potential_energy
    if decomposed:
        transport = ctx.getPlatformProperty("CommId")       # what the plugin actually uses
        if ctx.getPlatformProperty("DisablePmeStream") == "true":
            transport += ", single stream"

    def barrier():
        if dist is not None:
            import torch
            if not EMULATED:
                torch.cuda.synchronize()
            dist.barrier()

    profile = not args.no_roofline
    if profile:
        kernels.lib.ommhip_profile_reset()
        # the timed region is perturbed as little as possible: events are created beforehand; a short region (the driver's 20 steps) times
        # only the dominant launch group, a long one all five timers -- at every 7th launch either way (three samples in 20 steps)
        short = args.steps < 100 and args.profile_every <= 0
        kernels.lib.ommhip_profile_enable_timers(args.profile_every if args.profile_every > 0 else 7, 0x1 if short else 0x1f, 64 if short else 512)
    serialized = decomposed and args.serialize_ranks and args.transport == "gloo"
    if serialized:
        barrier()
        MR.serial_reset(EMULATED)
    elapsed, st = timed_run(integ, ctx, args.steps, (lambda: None) if serialized else barrier)
    if serialized:
        compute_s = MR.serial_release(EMULATED)
        mine = [1e3 * compute_s / args.steps, MR.SERIAL["collectives"] / float(args.steps), 1e3 * MR.SERIAL["segments_s"] / args.steps]
        everyone = [None] * world
        dist.all_gather_object(everyone, mine)
    if profile:
        kernels.lib.ommhip_profile_enable(0)
    with_query = MR.max_over_ranks(elapsed + CLOSING_QUERY_S[-1], dist, device="cpu")
    elapsed = MR.max_over_ranks(elapsed, dist, device="cpu")
    if not np.isfinite(st.potentialEnergy):
        raise RuntimeError("simulation blew up: potential energy is not finite")

    ms_per_step = 1e3 * elapsed / args.steps
    value = MR.ns_per_day(elapsed, args.steps, args.dt_fs)          # ONE simulation, whatever the number of GPUs
    grid = nb.getPMEParametersInContext(ctx)[1:]
    out = {
        "metric": "ns/day (DHFR PME 2 fs) at 1/2/4/8 MI355X; force max-rel-err vs Reference",
        "value": round(value, 3), "unit": "ns/day", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 5), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32",
        # N = 1: one run of the real benchmark System; N > 1: ONE fixed box over N GPUs (total work fixed)
        "data": ("real: examples/5dfr_solv-cube_equil.pdb with amber99sb + tip3p parameters (committed fixture tests/golden/dhfr_5dfr_amber99sb_tip3p.npz, equilibrated coordinates and velocities)"
                 if workload == "dhfr" else
                 ("synthetic: %s -- 27 copies of a 36 501-atom TIP3P box equilibrated for 60 ps at 300 K on this platform (tests/golden/water_tile_36501_equilibrated.npz, tools/make_water_tile.py)" % w.name)
                 if workload == "water1m" else "synthetic: %s (generated coordinates, standard TIP3P parameters)" % w.name),
        "config": {"workload": "%s: %d atoms, PME cutoff 0.9 nm grid %s, LangevinMiddle %.0f fs, HBonds constraints + rigid water; %s" % (
            w.name, w.num_atoms, "x".join(str(g) for g in grid), args.dt_fs,
            ("ONE box domain-decomposed over %d GPUs (x slabs, %s collectives)" % (world, transport)) if decomposed else "single GPU"),
            "precision": "mixed (f32 forces, fixed-point accumulation, f64 integration)", "device": device_name,
            # the timed Context integrates on the device and no Force of it runs as a Reference kernel on the host (checked below)
            "integration_mode": integration_mode, "fallback_forces": fallback_forces,
            "prepare_steps": prepare,
            "protocol": "%d untimed warm-up steps; then EXACTLY %d steps between two (barrier + device synchronisation) brackets -- ommhip_device_sync, because "
                        "torch.cuda.synchronize() belongs to PyTorch's own HIP runtime and does not see the plugin's streams; max over ranks" % (args.warmup, args.steps)},
        # examples/benchmark.py ends its timed region with getState(getEnergy=True) (one more evaluation, with energies, and a host round trip); until
        # round 5 this line's `value` counted it too.  Now it runs behind the closing bracket -- the figure with it inside is kept for comparison
        "closing_energy_query": {"ms": round(1e3 * (with_query - elapsed), 4),
                                 "value_with_it_inside_the_timed_region": round(MR.ns_per_day(with_query, args.steps, args.dt_fs), 3), "unit": "ns/day"},
    }
    if serialized:
        out["per_rank_compute_ms_per_step"] = {"ranks": [round(e[0], 4) for e in everyone], "collectives_per_step": everyone[0][1],
                                               "including_host_staging": [round(e[2], 4) for e in everyone],
                                               "note": "ranks serialized on ONE GPU (one rank's kernels at a time): each rank's wall time minus its time inside collectives, "
                                                       "the latter counted from a device-wide synchronisation at their start -- i.e. its step without communication; "
                                                       "`value` and ms_per_step are NOT a measurement in this mode"}
    if decomposed:
        try:
            di = H.domain_info()
            half = len(di) > 7 and di[7] > 0
            out["config"]["parallelism"] = "dd%d (x slabs; %s + 2 all-to-alls + potential planes%s per step)" % (
                world, "halo exchange of positions with the two neighbouring slabs" if di[1] else "all-gather of positions",
                " + forces on the lower neighbour's atoms returned (pairs across a boundary evaluated once)" if half else "")
            out["domain"] = {"exchange": "halo" if di[1] else "all-gather", "slots_per_rank": di[2], "slots_converted_per_step_rank0": di[3],
                             "slots_converted_over_slots_per_rank": round(di[3] / max(1, di[2]), 3),
                             "position_bytes_sent_per_step_rank0": di[4], "position_bytes_received_per_step_rank0": di[5], "re_sorts": di[6],
                             "pairs_across_boundaries": "once, by the upper rank (half-shell)" if half else "on both sides",
                             "force_bytes_returned_per_step_rank0": 24 * (di[7] - 1) if half else 0}
        except Exception as e:
            out["config"]["parallelism"] = "dd%d (x slabs)" % world
            out["domain"] = {"error": str(e)}

    if rank == 0:
        # ---- roofline of the dominant kernel (direct-space pair kernel)
        if profile:
            stats = (C.c_longlong * 8)()
            plugin.ommhip_plugin_nl_stats(stats)
            chunks, rows = stats[2], stats[3]
            timers = collect_timers(kernels)
            probe = None
            if world == 1 and timers["pme_fft"]["calls"] == 0 and timers["nb_direct"]["calls"] > 0:          # (ONE process only: in a decomposed run every rank would have to step along -- rank 0 stepping alone waits for ever in its first collective: the N > 1 hang of rounds 4 - 5, profiles/r11/r11be_*)
                # a short region timed the dominant launch group only: four more steps with every timer say whether the FFT stages
                # ran as launches of their own or inside the pair launches
                kernels.lib.ommhip_profile_enable_timers(1, 0x1f, 64)
                before = collect_timers(kernels)
                integ.step(4)
                ctx.getState(getEnergy=True)
                kernels.lib.ommhip_profile_enable(0)
                probe = collect_timers(kernels)
                for k in probe:
                    if k != "nb_direct":
                        timers[k] = probe[k]
                timers["nb_direct"] = before["nb_direct"]
            # The three fused launches one by one (their own dispatch timestamps: an event pair riding on every launch) are sampled AFTER the
            # timed region -- 32 more steps, every launch -- because an event pair per launch costs ~40 us of host time per sampled step
            # (a 20-step region with 10 such samples read 1 170 ns/day against 1 390 without: profiles/r09k_*); inside the region only the
            # pair that brackets the group is taken (start stamped by the first launch, stop by the last), which costs ~1 %.
            if world == 1 and timers["nb_direct"]["calls"] > 0:          # (one process: nobody else has to step along)
                keep = dict(timers)
                kernels.lib.ommhip_profile_enable_timers(1, 0x1 | (0x7 << 5), 64)
                integ.step(32)
                ctx.getState(getEnergy=True)
                kernels.lib.ommhip_profile_enable(0)
                after = collect_timers(kernels)
                for k in range(3):
                    timers["pairs_fft_stage%d" % k] = after.get("pairs_fft_stage%d" % k, {"calls": 0, "avg_us": None})
                for k in keep:
                    if not k.startswith("pairs_fft_stage"):
                        timers[k] = keep[k]
            # algorithmic bytes of one launch: per row 64 j-slots x (index 4 + mask 4 + posq 16 + sigEps 8 + force 24)
            # plus per chunk 32 i-atoms x (posq 16 + sigEps 8 + force 24)
            algo_bytes_with_list_words = rows * 64 * 56 + chunks * 32 * 48
            # `frac` is quoted on SURVEY.md 8(d)'s own formula: B_dir = 48 N + 52 * 32 * T with T = 32 x 32 tiles = 2 per 64-slot row (the i side
            # once per atom, no mask word); the accounting above -- i side once per chunk, the mask word of every j slot -- is ~20 % more
            # generous and is reported beside it as frac_counting_list_words ("what's weak" 6)
            algo_bytes = 48 * w.num_atoms + 52 * 32 * (2 * rows)
            avg_us = timers["nb_direct"]["avg_us"]
            # Fused single-stream path: the pair kernel rides on the three FFT launches (ommhip_pairs_with_fft), the timer then
            # brackets those three launches and the algorithmic bytes include the FFT stages' grid traffic: real grid read +
            # complex written, complex read + written + influence function read, complex read + real written
            kernel_name = "nb_direct"
            fused = timers["pme_fft"]["calls"] == 0 if probe is not None else timers["pme_fft"]["calls"] * 2 < timers["nb_direct"]["calls"]
            gx, gy, gz = grid
            real_b, cplx_b = gx * gy * gz * 4, gx * gy * (gz // 2 + 1) * 8
            if fused:
                algo_bytes += (real_b + cplx_b) + (2 * cplx_b + cplx_b // 2) + (cplx_b + real_b)
                algo_bytes_with_list_words += (real_b + cplx_b) + (2 * cplx_b + cplx_b // 2) + (cplx_b + real_b)
                kernel_name = "pairs_fft_plane + pairs_fft_lines + pairs_fft_plane (pair kernel riding on the 3 FFT launches)"
            # avg_us so far: HIP events around the group of launches (the gaps between the three fused launches included).  The launches' own
            # dispatch timestamps -- start / stop events riding on each launch's packet -- give the kernel time proper: their sum is what the
            # roofline fraction uses (and what a rocprofv3 kernel trace of the same command shows); the span is reported beside it.
            span_us = avg_us
            stage_us = [timers.get("pairs_fft_stage%d" % k, {}).get("avg_us") for k in range(3)]
            if fused and all(v for v in stage_us):
                avg_us = sum(stage_us)
            achieved = algo_bytes / (avg_us * 1e-6) / 1e9 if avg_us else None
            achieved_list_words = algo_bytes_with_list_words / (avg_us * 1e-6) / 1e9 if avg_us else None
            # HBM traffic of the same kernel from the PMC passes (rocprofv3 cannot run inside this process; the counters are
            # collected by tools/gpu_pmc2.sh on the same command and committed under profiles/ with the hash of the kernel
            # sources they were taken from; a summary of other sources is stale and is not quoted)
            traffic, traffic_source = None, None
            sha = kernel_sources_sha()
            pmc_file = os.path.join(ROOT, "profiles", "pmc_pairs_fft.json" if fused else "pmc_nb_direct.json")
            if workload == "dhfr" and os.path.exists(pmc_file):
                with open(pmc_file) as f:
                    pmc = json.load(f)
                if pmc.get("kernel_sources_sha") == sha:
                    traffic = pmc["traffic_bytes_per_launch"]
                    traffic_source = ("NOT measured in this run: rocprofv3 PMC counters cannot be read from inside the process, so the figure is a committed PMC visit of "
                                      "the same command on the same kernel sources (hash checked): %s; visit %s" % (pmc["source"], pmc.get("visit", "date and box not recorded")))
                else:
                    traffic_source = "stale: %s was collected from other kernel sources (%s, now %s)" % (os.path.basename(pmc_file), pmc.get("kernel_sources_sha"), sha)
            # ... unless rocprofv3 is on this box: then the counters are collected now, from two child runs of this very command line's
            # workload (after the timed region; ~15 s each)
            traffic_detail = None
            if workload == "dhfr" and fused and world == 1 and not args.no_pmc and not EMULATED and os.environ.get("BENCH_PROFILER_CHILD") != "1":
                try:
                    m = measure_group_traffic()
                    traffic, traffic_detail = m["traffic"], m["detail"]
                    traffic_source = ("measured in this run: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, --kernel-trace only) over two child runs of the "
                                      "same workload on this box after the timed region (40 steps each); bytes per launch group = 2 x pairs_fft_plane + pairs_fft_lines")
                except Exception as e:
                    traffic_source = (traffic_source or "") + " [in-run PMC passes failed: %s]" % str(e)[:200]
            out["roofline"] = {"bound": "hbm", "kernel": kernel_name, "achieved": round(achieved, 2) if achieved else None, "peak": HBM_PEAK_GBPS,
                               "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 5) if achieved else None, "traffic": traffic,
                               "traffic_source": traffic_source, "traffic_detail": traffic_detail,
                               "algorithmic_bytes_formula": "SURVEY.md 8(d): B_dir = 48 N + 52*32*T, T = 2 tiles per 64-slot row" + (" + grid traffic of the three FFT stages as implemented (real read + complex written; complex read + written + influence function; complex read + real written)" if fused else ""),
                               "frac_counting_list_words": round(achieved_list_words / HBM_PEAK_GBPS, 5) if achieved_list_words else None,
                               "algorithmic_bytes_counting_list_words": int(algo_bytes_with_list_words),
                               "algorithmic_bytes_per_launch": int(algo_bytes), "avg_kernel_us": round(avg_us, 3) if avg_us else None,
                               "avg_kernel_us_source": ("sum of the three launches' own dispatch timestamps (hipExtLaunchKernelGGL start / stop events on every launch of 32 steps after the timed region; avg_span_us_including_launch_gaps is the in-region sample of the group)" if fused and all(v for v in stage_us)
                                                        else "HIP events around the launch"),
                               "avg_span_us_including_launch_gaps": round(span_us, 3) if span_us else None,
                               "rows": int(rows), "chunks": int(chunks), "rebuilds": int(stats[5]),
                               "rows_as_built": int(stats[7]),          # before the per-step pruning to the cutoff itself (`rows` is what the pair kernel walks)
                               "pair_evals_per_launch": int(rows) * 64 * 32, "kernel_timers_us": timers, "kernel_sources_sha": sha,
                               "note": "working set is cache-resident at DHFR size (the kernel is FP32-issue bound there, see fp32_issue); see DESIGN.md (d)"}
            # the bound that actually applies to the pair kernel at this size: FP32 vector issue.  Evaluated pairs x 45 flop over
            # the kernel's own time -- stand-alone launches when the timed region ran it fused with the FFT stages (filled in below) -- and
            # the part of it spent on pairs inside the cutoff (counted on the final configuration with a k-d tree).
            out["roofline"]["fp32_issue"] = {"bound": "fp32 vector issue", "peak": FP32_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s", "flop_per_pair": PAIR_FLOP,
                                             "pair_evals_per_launch": int(rows) * 64 * 32}
            try:
                if w.num_atoms <= 120000 and not decomposed:          # (a State query is a collective in a decomposed run: never from rank 0 alone)
                    from scipy.spatial import cKDTree
                    endp = ctx.getState(getPositions=True).positions
                    Lbox = np.diag(np.asarray(w.box, float))
                    wrapped = np.mod(endp, Lbox[None, :])
                    wrapped[wrapped >= Lbox[None, :]] = 0.0
                    inside = int(cKDTree(wrapped, boxsize=Lbox).count_neighbors(cKDTree(wrapped, boxsize=Lbox), w.cutoff) - w.num_atoms) // 2
                    out["roofline"]["fp32_issue"]["pairs_inside_cutoff"] = inside
                    out["roofline"]["fp32_issue"]["evals_per_useful_pair"] = round(int(rows) * 64 * 32 / max(inside, 1), 3)
            except Exception as e:
                out["roofline"]["fp32_issue"]["pairs_inside_cutoff_error"] = str(e)
            # ---- the 3-D FFT chain on its own: 96 * Hc algorithmic bytes over its measured duration.  On the fused
            #      path the FFT stages share launches with the pair kernel, so a short extra run with separate launches times them.
            if not decomposed:
                try:
                    fft = timers["pme_fft"]
                    if fused:
                        os.environ["OPENMM_HIP_NO_PAIRS_WITH_FFT"] = "1"          # read per evaluation by the plugin
                        kernels.lib.ommhip_profile_reset()
                        kernels.lib.ommhip_profile_enable(1)
                        integ.step(100)
                        ctx.getState(getEnergy=True)
                        kernels.lib.ommhip_profile_enable(0)
                        os.environ.pop("OPENMM_HIP_NO_PAIRS_WITH_FFT")
                        sep = collect_timers(kernels)
                        fft = sep["pme_fft"]
                        out["roofline"]["separate_launch_timers_us"] = {k: sep[k] for k in ("nb_direct", "pme_fft")}
                    pair_us = (sep if fused else timers)["nb_direct"]["avg_us"]
                    if pair_us:
                        fi = out["roofline"]["fp32_issue"]
                        fi["kernel_us"] = round(pair_us, 3)
                        fi["kernel"] = "nb_direct as a launch of its own" + (" (100 extra steps after the timed region)" if fused else "")
                        fi["achieved"] = round(fi["pair_evals_per_launch"] * PAIR_FLOP / (pair_us * 1e-6) / 1e12, 3)
                        fi["frac"] = round(fi["achieved"] / FP32_VECTOR_PEAK_TFLOPS, 5)
                        if fi.get("pairs_inside_cutoff"):
                            fi["useful_frac"] = round(fi["pairs_inside_cutoff"] * PAIR_FLOP / (pair_us * 1e-6) / 1e12 / FP32_VECTOR_PEAK_TFLOPS, 5)
                    hc = gx * gy * (gz // 2 + 1)
                    if fft["avg_us"]:
                        a = 96.0 * hc / (fft["avg_us"] * 1e-6) / 1e9
                        out["roofline_fft"] = {"bound": "hbm", "kernel": "forward plane/line transforms + x transform with convolution + backward transforms",
                                               "grid": [gx, gy, gz], "algorithmic_bytes": 96 * hc, "avg_us": round(fft["avg_us"], 3), "achieved": round(a, 2),
                                               "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(a / HBM_PEAK_GBPS, 5), "calls": fft["calls"]}
                except Exception as e:
                    out["roofline_fft"] = {"error": str(e)}
        # ---- CPU baseline: the reference's platforms/cpu on the same System, bounded sample
        if args.cpu_steps > 0 and world == 1:        # rank 0 at N = 1 only
            try:
                H.load_cpu_platform()
                csys, cnb, cinteg, cctx = start_platform(w, "CPU", dt_ps, 5)
                t0 = time.perf_counter()
                cinteg.step(args.cpu_steps)          # (a synchronous platform: the steps are done when the call returns)
                cpu_elapsed = time.perf_counter() - t0
                cctx.getState(getEnergy=True)
                threads = cctx.getPlatformProperty("Threads")
                out["cpu_baseline"] = {"value": round(args.dt_fs * 1e-6 * args.cpu_steps / cpu_elapsed * 86400.0, 4), "unit": "ns/day",
                                       "cores": int(threads) if threads else os.cpu_count(), "kind": "reference",
                                       "sample": "%d steps of the same System on platforms/cpu from oracle/_ref (%.1f s; PME via the reference's single-threaded fftpack path, FFTW plugin not available)" % (args.cpu_steps, cpu_elapsed),
                                       "ms_per_step": round(1e3 * cpu_elapsed / args.cpu_steps, 3)}
                cctx.close()
            except Exception as e:  # the baseline must never take the GPU number down with it
                out["cpu_baseline"] = {"value": None, "unit": "ns/day", "cores": os.cpu_count(), "kind": "reference", "sample": "failed: %s" % e}
            try:
                # second half of BASELINE.json's metric: force max-rel-err vs the Reference platform (oracle/_ref) on the
                # configuration the timed run ended in -- SURVEY.md §8(d): max_i |dF_i| / max(|F_ref,i|, RMS force), ALL atoms
                if w.num_atoms > 120000:
                    raise RuntimeError("skipped at this size (the Reference platform needs minutes); see tests/test_gpu_platform.py::test_water1m_forces_within_1e4_of_reference")
                if decomposed:
                    raise RuntimeError("skipped in a decomposed run (a State query from rank 0 alone would wait for the other ranks); see tests/test_gpu_multirank.py")
                end = ctx.getState(getPositions=True, getForces=True)
                rsys, rnb = w.build()
                rctx = H.Context(rsys, H.Integrator(H.VERLET, 0.001), "Reference")
                rctx.setPositions(end.positions)
                f_ref = rctx.getState(getForces=True).forces
                rctx.close()
                from openmm_amd.parity import force_parity
                fp = force_parity(end.positions, w.box, w.cutoff, end.forces, f_ref)
                out["force_parity"] = {"max_rel_err_vs_reference": fp["max_rel_err_all_atoms"], "tolerance": 1e-4,
                                       "max_rel_err_all_atoms": fp["max_rel_err_all_atoms"], "median_rel_diff": fp["median_rel_diff"],
                                       "atoms_above_tolerance": fp["atoms_above_tolerance"], "cutoff_edge_pairs": fp["cutoff_edge_pairs"],
                                       "max_rel_err_away_from_cutoff_edge_pairs": fp["max_rel_err"], "edge_band_nm": fp["edge_band_nm"],
                                       "oracle": "platforms/reference from oracle/_ref, final configuration of the timed run",
                                       "note": "headline = maximum over ALL atoms; the pair kernel works on block-relative coordinates, so only pairs within "
                                               "%.0e nm of the cutoff (where the truncated force jumps) can differ from the reference's side of it" % fp["edge_band_nm"]}
            except Exception as e:
                out["force_parity"] = {"max_rel_err_vs_reference": None, "error": str(e)}
    ctx.close()

    # ---- N > 1: the same box on rank 0's GPU alone, in the same job (the other ranks wait at the barrier below)
    if decomposed and rank == 0 and not args.no_scale_workload:
        try:
            ssys, snb, sinteg, sctx = start_platform(w, "HIP", dt_ps, args.warmup, {"DeviceIndex": str(local_rank)}, seed=1, prepare=prepare)
            s_elapsed, s_st = timed_run(sinteg, sctx, args.steps, lambda: None)
            e0_single = sctx.initial_potential_energy
            out["single_gpu_same_box"] = {"initial_energy_kj_mol": {"decomposed": e0_run, "single_gpu": e0_single,
                                                                    "rel_diff": abs(e0_run - e0_single) / max(abs(e0_single), 1.0)},
                                          "value": round(MR.ns_per_day(s_elapsed, args.steps, args.dt_fs), 3), "unit": "ns/day",
                                          "ms_per_step": round(1e3 * s_elapsed / args.steps, 5), "steps": args.steps, "warmup": args.warmup,
                                          "prepare_steps": prepare, "note": "same System, same protocol, rank 0's GPU alone, measured after the decomposed run"}
            sctx.close()
        except Exception as e:
            out["single_gpu_same_box"] = {"value": None, "error": str(e)}
    if dist is not None:
        dist.barrier()

    # ---- N = 1: the single-GPU point of the strong-scaling curve (same workload, same protocol as the N > 1 runs)
    if world == 1 and workload == "dhfr" and not args.no_scale_workload:
        try:
            sw = make_workload("water1m", seed=1)
            ssys, snb, sinteg, sctx = start_platform(sw, "HIP", dt_ps, args.warmup, {"DeviceIndex": str(local_rank)}, seed=1, prepare=default_prepare(sw))
            s_elapsed, s_st = timed_run(sinteg, sctx, args.steps, barrier)
            out["scale_workload"] = {"workload": "%s: ONE box of %d atoms, PME grid %s, single GPU (the N = 1 point of the strong-scaling curve that "
                                                 "bench.py --gpus N reports for N > 1)" % (sw.name, sw.num_atoms, "x".join(str(g) for g in snb.getPMEParametersInContext(sctx)[1:])),
                                     "value": round(MR.ns_per_day(s_elapsed, args.steps, args.dt_fs), 3), "unit": "ns/day",
                                     "ms_per_step": round(1e3 * s_elapsed / args.steps, 5), "steps": args.steps, "prepare_steps": default_prepare(sw)}
            sgrid = snb.getPMEParametersInContext(sctx)[1:]
            sctx.close()
            if not args.no_roofline:
                # the 3-D FFT chain of this grid on its own: the same box with reciprocal space on the main stream (in the timed run above the
                # chain shares the chip with the pair kernel on a side stream, and its timer measures that overlap), HIP events around the chain
                fsys, fnb, finteg, fctx = start_platform(sw, "HIP", dt_ps, 5, {"DeviceIndex": str(local_rank), "DisablePmeStream": "true"}, seed=1, prepare=0)
                try:
                    # pairs inside the cutoff of this box: the water tile's number density x the cutoff sphere (a homogeneous liquid; counted exactly for DHFR)
                    vol = float(np.prod(np.diag(np.asarray(sw.box, float))))
                    pairs_in = 0.5 * sw.num_atoms * (sw.num_atoms / vol) * 4.0 / 3.0 * np.pi * sw.cutoff ** 3
                    out["scale_workload"]["roofline_kernels"] = kernel_rooflines(kernels, plugin, finteg, fctx, sw.num_atoms, sgrid, cutoff_pairs=pairs_in)
                    out["scale_workload"]["roofline_kernels"]["direct"]["fp32_issue"]["pairs_inside_cutoff_source"] = "number density x cutoff sphere (homogeneous liquid)"
                except Exception as e:
                    out["scale_workload"]["roofline_kernels"] = {"error": str(e)[:200]}
                ft = collect_timers(kernels)["pme_fft"]
                fctx.close()
                if ft["avg_us"]:
                    hc = sgrid[0] * sgrid[1] * (sgrid[2] // 2 + 1)
                    fa = 96.0 * hc / (ft["avg_us"] * 1e-6) / 1e9
                    out["scale_workload"]["roofline_fft"] = {"bound": "hbm", "kernel": "forward plane transforms + x transform with convolution + backward plane transforms, reciprocal space on the main stream",
                                                             "grid": list(sgrid), "algorithmic_bytes": 96 * hc, "avg_us": round(ft["avg_us"], 3), "achieved": round(fa, 2), "peak": HBM_PEAK_GBPS,
                                                             "unit": "GB/s", "frac": round(fa / HBM_PEAK_GBPS, 5), "calls": ft["calls"]}
        except Exception as e:
            out["scale_workload"] = {"value": None, "error": str(e)}
    # ---- N = 1: driver-timed figures for BASELINE.json configs[2] (apoa1-sized) and for the benchmark script's own 4 fs step
    if world == 1 and workload == "dhfr" and not args.no_extra_workloads and abs(args.dt_fs - 2.0) < 1e-9:
        out["extra_workloads"] = {}
        for key, wl_name, dt_fs in (("dhfr_4fs", "dhfr", 4.0), ("apoa1", "apoa1", 2.0)):
            try:
                xw = w if wl_name == "dhfr" else make_workload(wl_name, seed=1)
                xprep = default_prepare(xw)
                xsys, xnb, xinteg, xctx = start_platform(xw, "HIP", dt_fs * 1e-3, args.warmup, {"DeviceIndex": str(local_rank)}, seed=1, prepare=xprep)
                x_elapsed, x_st = timed_run(xinteg, xctx, args.steps, barrier)
                if not np.isfinite(x_st.potentialEnergy):
                    raise RuntimeError("potential energy is not finite")
                out["extra_workloads"][key] = {"workload": "%s: %d atoms, PME grid %s, LangevinMiddle %.0f fs, single GPU" % (
                                                   xw.name, xw.num_atoms, "x".join(str(g) for g in xnb.getPMEParametersInContext(xctx)[1:]), dt_fs),
                                               "value": round(MR.ns_per_day(x_elapsed, args.steps, dt_fs), 3), "unit": "ns/day",
                                               "ms_per_step": round(1e3 * x_elapsed / args.steps, 5), "steps": args.steps, "warmup": args.warmup, "prepare_steps": xprep}
                xgrid = xnb.getPMEParametersInContext(xctx)[1:]
                xctx.close()
                if key == "apoa1" and not args.no_roofline:
                    # BASELINE.json configs[2]: rooflines of its kernels, each on its own (a second Context with reciprocal space on the main stream)
                    try:
                        rsys, rnb, rinteg, rctx = start_platform(xw, "HIP", dt_fs * 1e-3, 5, {"DeviceIndex": str(local_rank), "DisablePmeStream": "true"}, seed=1, prepare=0)
                        vol = float(np.prod(np.diag(np.asarray(xw.box, float))))
                        pairs_in = 0.5 * xw.num_atoms * (xw.num_atoms / vol) * 4.0 / 3.0 * np.pi * xw.cutoff ** 3
                        out["extra_workloads"][key]["roofline_kernels"] = kernel_rooflines(kernels, plugin, rinteg, rctx, xw.num_atoms, xgrid, cutoff_pairs=pairs_in)
                        rctx.close()
                    except Exception as e:
                        out["extra_workloads"][key]["roofline_kernels"] = {"error": str(e)[:200]}
            except Exception as e:
                out["extra_workloads"][key] = {"value": None, "error": str(e)}
        # BASELINE.json configs[4] (amoeba-pme, amoeba2009 DHFR) and the larger AMOEBA water tile: ONE child process with ONE overall timeout
        # (tools/bench_amoeba_legs.py) -- the headline process ends as soon as its own legs do
        try:
            import subprocess
            cmd = [sys.executable, os.path.join(ROOT, "tools", "bench_amoeba_legs.py"), "--steps", str(args.steps), "--device", str(local_rank)] + (["--no-pmc"] if args.no_pmc else [])
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=args.amoeba_timeout)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if r.returncode != 0 or not line:
                raise RuntimeError("exit code %d: %s" % (r.returncode, (r.stderr or r.stdout)[-300:]))
            out["extra_workloads"].update(json.loads(line[-1]))
        except Exception as e:
            out["extra_workloads"]["amoeba_water"] = out["extra_workloads"]["amoeba_dhfr"] = {"value": None, "error": "AMOEBA legs (child process): %s" % str(e)[:400]}
    if rank == 0:
        # librccl prints a version banner through C stdio, which is flushed at exit -- after Python's own output -- when stdout
        # is a pipe or a file: push it out first so that the JSON line is the last line
        sys.stdout.flush()
        try:
            C.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()
    if os.environ.get("BENCH_CHILD") == "1" and os.environ.get("BENCH_NO_HARD_EXIT") != "1":          # (a profiler writes its output at a regular exit: tools/gpu_visit.sh serialtimeline)
        # a launcher's child (N > 1): its exit code decides whether the launchers keep this configuration -- nothing that happens while the
        # interpreter and the libraries (RCCL, gloo, HIP) tear themselves down may turn a finished run into a failed attempt
        sys.stdout.flush(); sys.stderr.flush()
        os._exit(0)


if __name__ == "__main__":
    main()
