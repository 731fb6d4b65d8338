#!/usr/bin/env python
"""bench.py -- ns/day of the MD hot path (NonbondedForce direct space + PME + LangevinMiddle/SETTLE) on the
OpenMM "HIP" platform, one process per GPU.

    python bench.py --gpus 1 --steps 3000 --warmup 300
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): a DHFR-sized system -- 23 558 atoms in a 6.223 nm
cube, PME, cutoff 0.9 nm, Ewald tolerance 5e-4 (alpha 2.92/nm, grid 56^3), LangevinMiddleIntegrator 300 K,
1/ps, X-H constraints + rigid water, dt 2 fs -- generated synthetically (openmm_amd/testsystems.py:dhfr_like).
A "step" is one MD step = one pass of the hot path.  The timing protocol is the one of examples/benchmark.py:9-18:
warm-up steps, then time step(K) followed by getState(energy), which forces a device sync.

Multi-GPU: the path does not shard in this round (domain decomposition is SURVEY.md §8e, planned); --gpus N runs
N independent replicas of the workload, one per GPU, and reports the aggregate ns/day ("scaling": "weak").

Rank 0 prints ONE JSON line with the contract fields plus `roofline` (the direct-space pair kernel -- in the default
single-stream mode the three launches it shares with the FFT stages of reciprocal space -- measured with HIP events on
the stream the kernels run on), `cpu_baseline` (the reference's own platforms/cpu built into oracle/_ref, timed on this
host on a bounded number of steps of the same System) and `force_parity` (HIP forces of the final configuration against
the reference's Reference platform, the second half of BASELINE.json's metric).  DESIGN.md (d) defines every field.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0    # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=3000)
    p.add_argument("--warmup", type=int, default=300)
    p.add_argument("--dt-fs", type=float, default=2.0)
    p.add_argument("--workload", default="dhfr", choices=["dhfr", "water24k", "water98k", "water1m"])
    p.add_argument("--cpu-steps", type=int, default=150, help="steps of the CPU-platform baseline (0 disables it and the force-parity check)")
    p.add_argument("--no-roofline", action="store_true")
    p.add_argument("--profile-every", type=int, default=8, help="HIP-event timing of every n-th launch of each profiled kernel inside the timed region")
    p.add_argument("--props", default="", help="extra HIP platform properties, e.g. DisablePmeStream=true")
    return p.parse_args()


def make_workload(name, seed):
    from openmm_amd import testsystems as T
    if name == "dhfr":
        return T.dhfr_like(seed=seed)
    if name == "water24k":
        return T.water_box(20, seed=seed)
    if name == "water1m":
        return T.water_box(69, seed=seed)        # 985 527 atoms, L = 21.4 nm (BASELINE.json configs[3]; lattice start)
    return T.water_box(32, seed=seed)


def run_platform(w, platform, dt_ps, steps, warmup, props=None, seed=1):
    """-> (seconds for `steps`, final State, context)"""
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
    integ.step(warmup)
    ctx.getState(getEnergy=True)
    return system, nb, integ, ctx


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")      # "gloo" only to rehearse the N > 1 flow on a 1-GPU box
        ndev = torch.cuda.device_count()
        if backend == "nccl":
            if local_rank >= ndev:
                raise RuntimeError("rank %d has no GPU (%d visible)" % (local_rank, ndev))
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            local_rank = local_rank % max(ndev, 1)
            dist.init_process_group(backend=backend)

    import numpy as np
    from openmm_amd import capi, harness as H
    H.load_hip_platform()
    kernels = capi.load()
    plugin = C.CDLL(os.path.join(H.LIB_DIR, "libOpenMMHIP.so"))

    dt_ps = args.dt_fs * 1e-3
    w = make_workload(args.workload, seed=1)        # every replica starts from the equilibrated fixture; the thermostat seeds differ
    props = {"DeviceIndex": str(local_rank)}
    for kv in filter(None, args.props.split(",")):
        k, v = kv.split("=")
        props[k] = v
    system, nb, integ, ctx = run_platform(w, "HIP", dt_ps, 0, args.warmup, props, seed=1 + rank)
    device_name = ctx.getPlatformProperty("DeviceName")

    def barrier():
        if dist is not None:
            import torch
            torch.cuda.synchronize()
            dist.barrier()
    on_gpu = dist is None or dist.get_backend() == "nccl"

    profile = not args.no_roofline
    if profile:
        kernels.lib.ommhip_profile_reset()
        kernels.lib.ommhip_profile_enable(max(1, args.profile_every))
    barrier()
    t0 = time.perf_counter()
    integ.step(args.steps)
    st = ctx.getState(getEnergy=True)        # blocks until the device is idle
    elapsed = time.perf_counter() - t0
    barrier()
    if profile:
        kernels.lib.ommhip_profile_enable(0)
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if on_gpu else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if not np.isfinite(st.potentialEnergy):
        raise RuntimeError("simulation blew up: potential energy is not finite")

    ms_per_step = 1e3 * elapsed / args.steps
    ns_per_day_one = args.dt_fs * 1e-6 * args.steps / elapsed * 86400.0
    value = ns_per_day_one * world
    out = {
        "metric": "ns/day (DHFR PME 2 fs) at 1/2/4/8 MI355X; force max-rel-err vs Reference",
        "value": round(value, 3), "unit": "ns/day", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 5), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%s: %d atoms, PME cutoff 0.9 nm grid %s, LangevinMiddle %.0f fs, X-H constraints + rigid water; %s" % (
            w.name, w.num_atoms, "x".join(str(g) for g in nb.getPMEParametersInContext(ctx)[1:]), args.dt_fs,
            "independent replicas, one per GPU" if world > 1 else "single GPU"),
            "precision": "mixed (f32 forces, fixed-point accumulation, f64 integration)", "device": device_name},
    }

    if rank == 0:
        # ---- roofline of the dominant kernel (direct-space pair kernel)
        if profile:
            stats = (C.c_longlong * 8)()
            plugin.ommhip_plugin_nl_stats(stats)
            chunks, rows = stats[2], stats[3]
            calls, total_ms = C.c_longlong(), C.c_double()
            timers = {}
            for name, idx in (("nb_direct", 0), ("nl_update", 1), ("pme_spread", 2), ("pme_fft", 3), ("pme_interpolate", 4)):
                kernels.lib.ommhip_profile_collect(idx, C.byref(calls), C.byref(total_ms))
                timers[name] = {"calls": calls.value, "avg_us": (1e3 * total_ms.value / calls.value) if calls.value else None}
            # algorithmic bytes of one launch: per row 64 j-slots x (index 4 + mask 4 + posq 16 + sigEps 8 + force 24)
            # plus per chunk 32 i-atoms x (posq 16 + sigEps 8 + force 24)
            algo_bytes = rows * 64 * 56 + chunks * 32 * 48
            avg_us = timers["nb_direct"]["avg_us"]
            # Single-stream default: the pair kernel rides on the three FFT launches (ommhip_pairs_with_fft), the timer then
            # brackets those three launches and the algorithmic bytes include the FFT stages' grid traffic: real grid read +
            # complex written, complex read + written + influence function read, complex read + real written
            kernel_name = "nb_direct"
            fused = timers["pme_fft"]["calls"] * 2 < timers["nb_direct"]["calls"]
            if fused:
                gx, gy, gz = nb.getPMEParametersInContext(ctx)[1:]
                real_b, cplx_b = gx * gy * gz * 4, gx * gy * (gz // 2 + 1) * 8
                algo_bytes += (real_b + cplx_b) + (2 * cplx_b + cplx_b // 2) + (cplx_b + real_b)
                kernel_name = "pairs_fft_plane + pairs_fft_lines + pairs_fft_plane (pair kernel riding on the 3 FFT launches)"
            achieved = algo_bytes / (avg_us * 1e-6) / 1e9 if avg_us else None
            # HBM traffic of the same kernel from the PMC passes (rocprofv3 cannot run inside this process; the counters were
            # collected by tools/gpu_pmc.sh on the same command and are committed under profiles/)
            traffic, traffic_source = None, None
            pmc_file = os.path.join(ROOT, "profiles", "r01o_pmc_pairs_fft.json" if fused else "r01_pmc_nb_direct.json")
            if args.workload == "dhfr" and os.path.exists(pmc_file):
                with open(pmc_file) as f:
                    pmc = json.load(f)
                traffic, traffic_source = pmc["traffic_bytes_per_launch"], pmc["source"]
            out["roofline"] = {"bound": "hbm", "kernel": kernel_name, "achieved": round(achieved, 2) if achieved else None, "peak": HBM_PEAK_GBPS,
                               "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 5) if achieved else None, "traffic": traffic,
                               "traffic_source": traffic_source,
                               "algorithmic_bytes_per_launch": int(algo_bytes), "avg_kernel_us": round(avg_us, 3) if avg_us else None,
                               "rows": int(rows), "chunks": int(chunks), "rebuilds": int(stats[5]),
                               "pair_evals_per_launch": int(rows) * 64 * 32, "kernel_timers_us": timers,
                               "note": "working set is cache-resident at this size; the kernel is FP32-VALU bound, see DESIGN.md"}
        # ---- CPU baseline: the reference's platforms/cpu on the same System, bounded sample
        if args.cpu_steps > 0 and world == 1:        # rank 0 at N = 1 only
            try:
                H.load_cpu_platform()
                csys, cnb, cinteg, cctx = run_platform(w, "CPU", dt_ps, 0, 5)
                t0 = time.perf_counter()
                cinteg.step(args.cpu_steps)
                cctx.getState(getEnergy=True)
                cpu_elapsed = time.perf_counter() - t0
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
                # configuration the timed run ended in -- SURVEY.md §8(d): max_i |dF_i| / max(|F_ref,i|, RMS force)
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
        print(json.dumps(out), flush=True)
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
