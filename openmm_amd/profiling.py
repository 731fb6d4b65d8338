"""rocprofv3 child runs for bench.py and tools/bench_amoeba_legs.py (measurement plumbing, not product code)."""
import os


def rocprof_child(command, pmc=None, timeout=240):
    """Run `command` (argv list) under rocprofv3 in a scratch directory and return the per-dispatch table of its kernels as a pandas
    DataFrame (Kernel_Name, dur_us[, counter value]) -- or raise.  The recipe of MI355X_MICROARCH.md / tools/gpu_visit.sh: counters in a
    pass of their own with --kernel-trace only (no other trace domain beside --pmc), one counter per pass, run from /tmp."""
    import glob
    import shutil
    import subprocess
    import tempfile
    import pandas as pd
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        raise RuntimeError("rocprofv3 not found")
    tmp = tempfile.mkdtemp(prefix="ommhip_prof_", dir="/tmp")
    try:
        cmd = [exe] + (["--pmc", pmc] if pmc else []) + ["--kernel-trace", "--output-format", "csv", "-d", tmp, "-o", "p", "--"] + command
        env = dict(os.environ, TMPDIR="/tmp", BENCH_PROFILER_CHILD="1")
        r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout)
        if r.returncode != 0:
            raise RuntimeError("rocprofv3 child exited with %d: %s" % (r.returncode, (r.stderr or r.stdout)[-300:]))
        if pmc:
            files = glob.glob(os.path.join(tmp, "**", "*counter_collection.csv"), recursive=True)
            if not files:
                raise RuntimeError("no counter_collection.csv")
            df = pd.read_csv(files[0])
            df = df[df["Counter_Name"] == pmc]
            per = df.groupby(["Dispatch_Id", "Kernel_Name"], as_index=False).agg(value=("Counter_Value", "sum"), start=("Start_Timestamp", "first"), end=("End_Timestamp", "first"))
        else:
            files = glob.glob(os.path.join(tmp, "**", "*kernel_trace.csv"), recursive=True)
            if not files:
                raise RuntimeError("no kernel_trace.csv")
            df = pd.read_csv(files[0])
            per = df.rename(columns={"Start_Timestamp": "start", "End_Timestamp": "end"})[["Kernel_Name", "start", "end"]].copy()
            per["value"] = 0.0
        per["dur_us"] = (per["end"] - per["start"]) / 1e3
        return per, r.stdout
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
