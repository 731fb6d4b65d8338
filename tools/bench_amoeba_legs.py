"""The two AMOEBA legs of bench.py's `extra_workloads` (amoeba_water: 36 501-atom water tile; amoeba_dhfr = BASELINE.json configs[4]) in a process of
their own: bench.py runs this script as ONE child with ONE overall timeout, so that its headline process ends as soon as its own legs do and a
hung AMOEBA leg (or a hung profiler child of it) cannot take the driver's run with it.
    python tools/bench_amoeba_legs.py --steps K [--device D] [--no-pmc]
prints one JSON line: {"amoeba_water": {...}, "amoeba_dhfr": {...}}"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
HBM_PEAK_GBPS = 8000.0


def run(steps, device, no_pmc):
    import numpy as np
    from openmm_amd import harness as H, multirank as MR
    from openmm_amd.profiling import rocprof_child
    H.load_hip_platform()
    from openmm_amd import capi
    kernels = capi.load()

    def device_sync():          # the brackets of bench.py's timed regions (timed_run there): the steps alone, the closing energy query behind them
        rc = kernels.lib.ommhip_device_sync(device)
        if rc != 0:
            raise RuntimeError("ommhip_device_sync(%d) failed: %d" % (device, rc))
    legs = {}
    # BASELINE.json configs[4] (amoeba-pme) stand-in: 12 167 AMOEBA waters (the equilibrated tile), multipole PME with mutual polarization
    # (epsilon 1e-5, cutoff 0.7 nm, benchmark.py:58-68) + buffered 14-7 vdW (0.9 nm) on the native kernels, Verlet 1 fs, bounded steps
    try:
        from openmm_amd import testsystems as T
        H.load_amoeba_plugins()
        before = H.amoeba_native_evaluations()
        aw = T.amoeba_water_tile(cutoff=0.7, vdw_cutoff=0.9, polarization=H.Mutual, epsilon=1e-5, ewald_tol=7.5e-4, grid=(80, 80, 80), a_ewald=float(np.sqrt(-np.log(2 * 7.5e-4)) / 0.7))
        asys, amp, avdw = aw.build()
        ainteg = H.Integrator(H.VERLET, 0.001)
        actx = H.Context(asys, ainteg, "HIP", {"DeviceIndex": str(device)})
        actx.setPositions(aw.positions)
        # parity at the benchmarked size: the forces of the initial configuration against the AMOEBA plugin's Reference kernels on the
        # Reference platform (committed golden of 12 000 sampled atoms, tools/make_golden_amoeba_water_tile.py; the golden was converged
        # to 1e-6 D, this run solves to 1e-5 D as benchmark.py does: the figure includes that difference)
        a_parity = None
        try:
            g = np.load(os.path.join(ROOT, "tests", "golden", "reference_forces_amoeba_water_tile_36501_mutual_sample.npz"))

            def rel_err(f):
                return np.linalg.norm(f[g["indices"]] - g["forces"], axis=1) / np.maximum(np.linalg.norm(g["forces"], axis=1), float(g["rms_force"]))
            rel_run = rel_err(actx.getState(getForces=True).forces)
            # the kernels' own distance from the Reference: the same System solved to the golden's 1e-6 D in a Context of its own
            pw = T.amoeba_water_tile(cutoff=0.7, vdw_cutoff=0.9, polarization=H.Mutual, epsilon=1e-6, ewald_tol=7.5e-4, grid=(80, 80, 80), a_ewald=float(np.sqrt(-np.log(2 * 7.5e-4)) / 0.7))
            psys, _, _ = pw.build()
            pctx = H.Context(psys, H.Integrator(H.VERLET, 0.001), "HIP", {"DeviceIndex": str(device)})
            pctx.setPositions(pw.positions)
            rel = rel_err(pctx.getState(getForces=True).forces)
            pctx.close()
            a_parity = {"max_rel_err_vs_reference": float(rel.max()), "tolerance": 1e-4, "atoms_above_tolerance": int((rel > 1e-4).sum()), "sampled_atoms": int(len(rel)),
                        "max_rel_err_at_the_run_epsilon": float(rel_run.max()),
                        "reference": "AMOEBA Reference kernels on the Reference platform, mutual epsilon 1e-6 (tests/golden/reference_forces_amoeba_water_tile_36501_mutual_sample.npz); "
                                     "max_rel_err_vs_reference: this platform solved to the same 1e-6 D; max_rel_err_at_the_run_epsilon: solved to the 1e-5 D of the timed run (benchmark.py's setting)"}
        except Exception as e:
            a_parity = {"max_rel_err_vs_reference": None, "error": str(e)}
        actx.setVelocitiesToTemperature(300.0, 5)
        ainteg.step(6)                 # the solver's first guess uses the dipoles of up to four earlier steps
        actx.getState(getEnergy=True)
        a_steps = max(5, min(steps, 20))
        builds0, solves0 = H.amoeba_list_builds(), H.amoeba_solver_iterations()
        device_sync()
        t0 = time.perf_counter()
        ainteg.step(a_steps)
        device_sync()
        a_elapsed = time.perf_counter() - t0
        a_st = actx.getState(getEnergy=True)
        after = H.amoeba_native_evaluations()
        builds1, solves1 = H.amoeba_list_builds(), H.amoeba_solver_iterations()
        if not np.isfinite(a_st.potentialEnergy):
            raise RuntimeError("potential energy is not finite")
        if after[0] - before[0] < a_steps or after[1] - before[1] < a_steps:
            raise RuntimeError("the native AMOEBA kernels did not run (evaluations vdw %d multipole %d)" % (after[0] - before[0], after[1] - before[1]))
        legs["amoeba_water"] = {"workload": "%s: %d atoms, AmoebaMultipoleForce PME 80x80x80 mutual polarization (epsilon 1e-5, cutoff 0.7 nm) + AmoebaVdwForce "
                                                              "(0.9 nm) on the native kernels, harmonic bonds / angles, Verlet 1 fs, single GPU" % (aw.name, aw.num_atoms),
                                                  "value": round(MR.ns_per_day(a_elapsed, a_steps, 1.0), 4), "unit": "ns/day", "ms_per_step": round(1e3 * a_elapsed / a_steps, 3),
                                                  "steps": a_steps, "warmup": 6, "dtype": "mixed: f32 pair arithmetic (covalently related pairs, sums, frames, solver vectors f64), f32 grids and solver pair cache",
                                                  "force_parity": a_parity,
                                                  "solver_iterations_per_solve": round((solves1[1] - solves0[1]) / max(1, solves1[0] - solves0[0]), 2),
                                                  "pair_list_builds_per_step": {"vdw": round((builds1[0] - builds0[0]) / a_steps, 3), "multipole": round((builds1[1] - builds0[1]) / a_steps, 3)},
                                                  "note": "the AMOEBA nonbonded kernels on a larger, water-only System (kept as the series of rounds 3-4); BASELINE.json configs[4] itself is extra_workloads.amoeba_dhfr"}
        actx.close()
    except Exception as e:
        legs["amoeba_water"] = {"value": None, "error": str(e)}
    # BASELINE.json configs[4] itself: examples/benchmark.py `amoebapme` -- DHFR in water (23 558 atoms), amoeba2009.xml, multipole PME cutoff
    # 0.7 nm / tolerance 7.5e-4 / mutual polarization to 1e-5 D, vdW cutoff 0.9 nm, no constraints, MTSLangevinIntegrator(300 K, 1/ps, 2 fs,
    # [(0, 2), (1, 1)]) with the multipoles and vdW in force group 1 (benchmark.py:58-78).  System from the fixture of
    # tools/make_amoeba_dhfr_fixture.py (openmm_amd/forcefield_amoeba.py reading the reference's force-field file).
    try:
        from openmm_amd import testsystems as T
        H.load_amoeba_plugins()
        g = np.load(os.path.join(ROOT, "tests", "golden", "reference_forces_amoeba_dhfr.npz"))

        def dhfr_context(epsilon, pin_grid):
            w = T.amoeba_dhfr(epsilon=epsilon, pin_grid=pin_grid)
            sysd, mpd, vdwd = w.build()
            integ = H.MTSLangevinIntegrator(300.0, 1.0, 0.002, [(0, 2), (1, 1)], seed=7)
            ctx = H.Context(sysd, integ, "HIP", {"DeviceIndex": str(device)})
            ctx.setPositions(w.positions)
            return w, integ, ctx
        d_parity = None
        try:
            # the kernels' distance from the Reference platform at the benchmarked size: solved to the golden's 1e-6 D
            pw, pinteg, pctx = dhfr_context(1e-6, True)
            d_parity = {"tolerance": 1e-4, "reference": "Reference platform (the reference's kernels), mutual epsilon 1e-6, all 23 558 atoms (tests/golden/reference_forces_amoeba_dhfr.npz)"}
            ref_nb = g["forces_vdw"].astype(np.float64) + g["forces_multipole"].astype(np.float64)
            for name, groups, ref in (("valence", 1, g["forces_valence"].astype(np.float64)), ("multipole_and_vdw", 2, ref_nb)):
                f = pctx.getState(getForces=True, groups=groups).forces
                rms = float(np.sqrt((ref ** 2).sum(1).mean()))
                rel = np.linalg.norm(f - ref, axis=1) / np.maximum(np.linalg.norm(ref, axis=1), rms)
                d_parity["max_rel_err_" + name] = float(rel.max())
                d_parity["atoms_above_tolerance_" + name] = int((rel > 1e-4).sum())
            pctx.close()
            # ... and with the induced dipoles converged to the 1e-5 D of the timed run (benchmark.py's setting) instead of the golden's 1e-6:
            # what is left is the truncation of the solve, which the Reference platform shows as well at that setting
            # (profiles/r11/reference_platform_at_run_epsilon.txt: the Reference platform at 1e-5 D against its own 1e-6 D forces)
            rw, rinteg, rctx = dhfr_context(1e-5, True)
            f = rctx.getState(getForces=True, groups=2).forces
            rms = float(np.sqrt((ref_nb ** 2).sum(1).mean()))
            rel = np.linalg.norm(f - ref_nb, axis=1) / np.maximum(np.linalg.norm(ref_nb, axis=1), rms)
            d_parity["max_rel_err_at_the_run_epsilon"] = float(rel.max())
            d_parity["atoms_above_tolerance_at_the_run_epsilon"] = int((rel > 1e-4).sum())
            rctx.close()
        except Exception as e:
            d_parity = {"error": str(e)}
        before = H.amoeba_native_evaluations()
        dw, dinteg, dctx = dhfr_context(1e-5, False)
        dctx.setVelocitiesToTemperature(300.0, 5)
        dinteg.step(6)
        dctx.getState(getEnergy=True)
        d_steps = max(5, min(steps, 20))
        builds0, solves0 = H.amoeba_list_builds(), H.amoeba_solver_iterations()
        device_sync()
        t0 = time.perf_counter()
        dinteg.step(d_steps)
        device_sync()
        d_elapsed = time.perf_counter() - t0
        d_st = dctx.getState(getEnergy=True)
        after = H.amoeba_native_evaluations()
        builds1, solves1 = H.amoeba_list_builds(), H.amoeba_solver_iterations()
        if not np.isfinite(d_st.potentialEnergy):
            raise RuntimeError("potential energy is not finite")
        if after[0] - before[0] < d_steps or after[1] - before[1] < d_steps:
            raise RuntimeError("the native AMOEBA kernels did not run (evaluations vdw %d multipole %d)" % (after[0] - before[0], after[1] - before[1]))
        legs["amoeba_dhfr"] = {"workload": "%s: %d atoms, examples/benchmark.py amoebapme -- AmoebaMultipoleForce PME (mutual, epsilon 1e-5, cutoff 0.7 nm, tolerance 7.5e-4) + "
                                                             "AmoebaVdwForce (0.9 nm) + all amoeba2009 valence terms, no constraints, MTSLangevinIntegrator 2 fs [(0,2),(1,1)], single GPU" % (dw.name, dw.num_atoms),
                                                 "value": round(MR.ns_per_day(d_elapsed, d_steps, 2.0), 4), "unit": "ns/day", "ms_per_step": round(1e3 * d_elapsed / d_steps, 3),
                                                 "steps": d_steps, "warmup": 6, "force_parity": d_parity,
                                                 "solver_iterations_per_solve": round((solves1[1] - solves0[1]) / max(1, solves1[0] - solves0[0]), 2),
                                                 "pair_list_builds_per_step": {"vdw": round((builds1[0] - builds0[0]) / d_steps, 3), "multipole": round((builds1[1] - builds0[1]) / d_steps, 3)}}
        # roofline of this workload's dominant kernel, k_mp_dipole_field (the induced field of the current dipoles, once per solver
        # iteration): per pair-list entry it streams the entry (4 B), the partner's packed dipoles (24 B) and the cached geometry / damped
        # chain coefficients (20 B); entries = ordered pairs inside multipole cutoff + list skin (counted here with a k-d tree on the
        # final configuration); duration = the kernel's own dispatches in a rocprofv3 --kernel-trace child run of tools/bench_amoeba.py
        if not no_pmc and os.environ.get("BENCH_PROFILER_CHILD") != "1":
            try:
                from scipy.spatial import cKDTree
                endp = dctx.getState(getPositions=True).positions
                Lbox = np.diag(np.asarray(dw.box, float))
                wrapped = np.mod(endp, Lbox[None, :])
                wrapped[wrapped >= Lbox[None, :]] = 0.0
                skin = float(os.environ.get("OPENMM_HIP_AMOEBA_SKIN", "0.05"))
                tree = cKDTree(wrapped, boxsize=Lbox)
                entries = int(tree.count_neighbors(tree, 0.7 + skin)) - dw.num_atoms          # ordered pairs: every atom lists all its partners
                per, child_out = rocprof_child([sys.executable, os.path.join(ROOT, "tools", "bench_amoeba.py"), "--dhfr", "--steps", "10"], timeout=300)
                sub = per[per["Kernel_Name"].str.contains("k_mp_dipole_field")]
                sub = sub[~sub["Kernel_Name"].str.contains("gradient")]
                a_us = float(sub["dur_us"].mean())
                a_bytes = 48 * entries
                a_ach = a_bytes / (a_us * 1e-6) / 1e9
                total_us = float(per["dur_us"].sum())
                top = per.groupby("Kernel_Name")["dur_us"].agg(["sum", "count", "mean"]).sort_values("sum", ascending=False).head(6)
                legs["amoeba_dhfr"]["roofline"] = {
                    "bound": "hbm", "kernel": "k_mp_dipole_field", "achieved": round(a_ach, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(a_ach / HBM_PEAK_GBPS, 5),
                    "traffic": None, "algorithmic_bytes_per_launch": int(a_bytes), "list_entries": entries, "bytes_per_entry": 48,
                    "avg_kernel_us": round(a_us, 3), "launches_in_the_traced_run": int(len(sub)), "share_of_kernel_time_in_the_traced_run": round(float(sub["dur_us"].sum()) / total_us, 4),
                    "flops_per_launch": int(entries * 2 * 30), "flops_note": "two dipole sets x ~30 double-precision flop per entry: 0.03 of the fp64 vector peak -- the kernel streams its cache",
                    "source": "rocprofv3 --kernel-trace child run of tools/bench_amoeba.py --dhfr --steps 10 on this box (16 steps with the warm-up), after the timed region",
                    "top_kernels_us": {str(k)[:60]: {"total": round(float(v["sum"]), 1), "calls": int(v["count"]), "avg": round(float(v["mean"]), 2)} for k, v in top.iterrows()}}
                # HBM-side traffic of the three long kernels of this workload from the PMC counters: two more child runs (FETCH_SIZE and
                # WRITE_SIZE cannot share a pass; --kernel-trace only beside --pmc), FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes
                try:
                    child = [sys.executable, os.path.join(ROOT, "tools", "bench_amoeba.py"), "--dhfr", "--steps", "6"]
                    kb = {}
                    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
                        cper, _ = rocprof_child(child, pmc=counter, timeout=300)
                        for key, pattern, exclude in (("k_mp_dipole_field", "k_mp_dipole_field", "gradient"), ("k_mp_forces<true>", "k_mp_forces<true>", None), ("k_mp_field<true>", "k_mp_field<true>", None)):
                            csub = cper[cper["Kernel_Name"].str.contains(pattern, regex=False)]
                            if exclude is not None:
                                csub = csub[~csub["Kernel_Name"].str.contains(exclude)]
                            if len(csub) > 0:
                                kb[(counter, key)] = (float(csub["value"].mean()), float(csub["dur_us"].mean()), int(len(csub)))
                    traffic = {}
                    for key in ("k_mp_dipole_field", "k_mp_forces<true>", "k_mp_field<true>"):
                        if ("FETCH_SIZE", key) in kb and ("WRITE_SIZE", key) in kb:
                            bytes_ = 1024.0 * (2.0 * kb[("FETCH_SIZE", key)][0] + kb[("WRITE_SIZE", key)][0])
                            us = kb[("FETCH_SIZE", key)][1]
                            traffic[key] = {"traffic_bytes_per_launch": int(bytes_), "fetch_kb_as_reported": round(kb[("FETCH_SIZE", key)][0], 1), "write_kb_as_reported": round(kb[("WRITE_SIZE", key)][0], 1),
                                            "kernel_us_under_the_profiler": round(us, 2), "hbm_side_gb_per_s": round(bytes_ / (us * 1e-6) / 1e9, 1),
                                            "frac_of_hbm_peak": round(bytes_ / (us * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4), "dispatches_averaged": kb[("FETCH_SIZE", key)][2]}
                    if "k_mp_dipole_field" in traffic:
                        legs["amoeba_dhfr"]["roofline"]["traffic"] = traffic["k_mp_dipole_field"]["traffic_bytes_per_launch"]
                    legs["amoeba_dhfr"]["roofline"]["traffic_by_kernel"] = traffic
                    legs["amoeba_dhfr"]["roofline"]["traffic_source"] = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE child runs (separate passes, --kernel-trace only) of tools/bench_amoeba.py "
                                                                         "--dhfr --steps 6 on this box; FETCH_SIZE x 2 (gfx950), KB -> bytes x 1024")
                except Exception as e:
                    legs["amoeba_dhfr"]["roofline"]["traffic_source"] = "PMC passes failed: %s" % str(e)[:200]
            except Exception as e:
                legs["amoeba_dhfr"]["roofline"] = {"error": str(e)[:300]}
        dctx.close()
    except Exception as e:
        legs["amoeba_dhfr"] = {"value": None, "error": str(e)}
    return legs


if __name__ == "__main__":
    p = argparse.ArgumentParser()
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--device", type=int, default=0)
    p.add_argument("--no-pmc", action="store_true")
    a = p.parse_args()
    result = run(a.steps, a.device, a.no_pmc)
    sys.stdout.flush()
    print(json.dumps(result), flush=True)
