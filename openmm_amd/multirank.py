"""Launcher side of a domain-decomposed run: one process per GPU (torch.distributed.run), one OpenMM Context per process,
all of them driving ONE simulation box.

The plugin does its collectives itself (RCCL, include/openmm_hip_comm.h); what it needs from the launcher is the
communicator's identity.  `domain_properties()` returns the platform properties that put a Context into the run:

    props = multirank.domain_properties(dist)                      # {"Ranks": ..., "Rank": ..., "CommId": ..., "DeviceIndex": ...}
    ctx = harness.Context(system, integrator, "HIP", props)

transport "rccl" (default): rank 0 creates the ncclUniqueId through the C ABI (ommhip_comm_unique_id) and broadcasts it with
torch.distributed.  transport "gloo": the host-staged callback transport -- the plugin hands host buffers to a ctypes
callback that performs the all-gather with torch.distributed (gloo); for tests on the CPU emulator and on one GPU shared
by several ranks (RCCL refuses two ranks on one device).

`max_over_ranks()` is the timing reduction of bench.py (the slowest rank decides the step time).

`run_attempts()` is the launcher's safety net: each rank's launcher process runs the actual work in a child process and the
launchers agree, through a TCPStore of their own, whether an attempt worked on every rank.  A collective that never returns
on some rank (a stuck GPU queue cannot be recovered from inside the process) is ended by killing exactly that child, and all
ranks move on to the next configuration together instead of hanging the node.
"""
import ctypes as C
import os
import subprocess
import threading
import time

_KEEP_ALIVE = []      # ctypes callbacks must outlive the Contexts that hold their address


def _kernel_lib(emulated=False):
    from . import capi, harness
    path = os.path.join(harness.EMU_DIR if emulated else harness.LIB_DIR, "libopenmm_hip_kernels.so")
    return capi.load(path).lib


def new_rccl_id(emulated=False):
    """A fresh ncclUniqueId (256 hex characters) from the C ABI; call on ONE rank and distribute the result."""
    buf = C.create_string_buffer(257)
    rc = _kernel_lib(emulated).ommhip_comm_unique_id(buf)
    if rc != 0:
        raise RuntimeError("ommhip_comm_unique_id failed (%d): is librccl available?" % rc)
    return buf.value.decode()


def rccl_comm_id(dist, emulated=False):
    """The 256-character hex ncclUniqueId of this run, the same on every rank."""
    ident = [None]
    if dist.get_rank() == 0:
        ident[0] = new_rccl_id(emulated)
    dist.broadcast_object_list(ident, src=0)
    return ident[0]


SERIAL = {"on": False, "active": False, "armed": False, "segments_s": 0.0, "collectives": 0, "t_return": None, "dist": None, "group": None}


def gloo_all_gather_callback(dist, group=None, serialize=False):
    """-> "callback:<fn>:<user>" CommId of the host-staged transport; the all-gather runs on `group` (a gloo group).

    serialize=True (diagnostics on a box with fewer GPUs than ranks): the ranks run their work between two collectives ONE AT A
    TIME -- rank r leaves a collective only when rank r-1 has finished the work that follows it -- so each rank's segments are
    timed on an otherwise idle GPU.  SERIAL["segments_s"] accumulates this rank's time between collectives (host clock, device
    idle at both ends); call serial_release() after the last step so that the next rank can finish."""
    import numpy as np
    import torch
    os.environ["OPENMM_HIP_ALLOW_CALLBACK_COMM"] = "1"          # the plugin takes a callback address from a process that asks for it only
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    proto = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)
    SERIAL.update({"on": serialize, "dist": dist, "group": group})
    token = torch.zeros(1, dtype=torch.int32)

    def all_gather(user, send, recv, nbytes):
        try:
            active = serialize and SERIAL["active"]       # only between serial_reset() and serial_release(): no other collectives there
            if active:
                if torch.cuda.is_available():
                    torch.cuda.synchronize()          # everything this rank queued since the last collective has run
                SERIAL["segments_s"] += time.perf_counter() - SERIAL["t_return"]
                if SERIAL["armed"] and rank + 1 < world:
                    dist.send(token, dst=rank + 1, group=group)          # my segment is done: the next rank (waiting below) may run its own
                SERIAL["collectives"] += 1
            src = np.ctypeslib.as_array(C.cast(send, C.POINTER(C.c_uint8)), shape=(nbytes,))
            dst = np.ctypeslib.as_array(C.cast(recv, C.POINTER(C.c_uint8)), shape=(world * nbytes,))
            out = torch.empty(world * nbytes, dtype=torch.uint8)
            dist.all_gather_into_tensor(out, torch.from_numpy(src.copy()), group=group)
            dst[:] = out.numpy()
            if active:
                if rank > 0:
                    dist.recv(token, src=rank - 1, group=group)              # wait until the rank before me has run its segment
                SERIAL["armed"] = True
                SERIAL["t_return"] = time.perf_counter()
            return 0
        except Exception as e:      # never let an exception cross the C boundary
            print("gloo all-gather callback failed:", e, flush=True)
            return 1
    fn = proto(all_gather)
    _KEEP_ALIVE.append(fn)
    return "callback:%d:0" % C.cast(fn, C.c_void_p).value


def serial_reset(emulated=False):
    """Start of the measured region (call right after a barrier)."""
    lib = _kernel_lib(emulated)
    lib.ommhip_comm_diag_seconds.restype = C.c_double
    lib.ommhip_comm_diag_seconds(1)
    SERIAL["t_region"] = time.perf_counter()
    SERIAL.update({"segments_s": 0.0, "collectives": 0, "t_return": time.perf_counter(), "active": SERIAL["on"], "armed": False})


def serial_release(emulated=False):
    """End of a serialized run: account the last segment and let the next rank finish its own.
    -> this rank's wall time of the region minus the time inside collectives (seconds): its steps without communication."""
    if not SERIAL["active"]:
        return None
    import torch
    dist, group = SERIAL["dist"], SERIAL["group"]
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    SERIAL["segments_s"] += time.perf_counter() - SERIAL["t_return"]
    if SERIAL["armed"] and dist.get_rank(group) + 1 < dist.get_world_size(group):
        dist.send(torch.zeros(1, dtype=torch.int32), dst=dist.get_rank(group) + 1, group=group)
    SERIAL.update({"active": False, "armed": False})
    lib = _kernel_lib(emulated)
    lib.ommhip_comm_diag_seconds.restype = C.c_double
    return (time.perf_counter() - SERIAL["t_region"]) - lib.ommhip_comm_diag_seconds(0)


def domain_properties(dist, transport="rccl", device_index=None, group=None, emulated=False, serialize=False):
    """Platform properties of this rank's Context in a decomposed run."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    props = {"Ranks": str(world), "Rank": str(rank)}
    if device_index is not None:
        props["DeviceIndex"] = str(device_index)
    if transport == "rccl":
        props["CommId"] = rccl_comm_id(dist, emulated)
    elif transport == "gloo":
        props["CommId"] = gloo_all_gather_callback(dist, group, serialize)
    else:
        raise ValueError("unknown transport %r" % transport)
    return props


def max_over_ranks(elapsed_s, dist=None, device="cuda"):
    """Elapsed time of the slowest rank (bench.py contract: MAX over ranks of the timed region)."""
    if dist is None:
        return elapsed_s
    import torch
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def ns_per_day(elapsed_s, steps, dt_fs):
    return dt_fs * 1e-6 * steps / elapsed_s * 86400.0


STORE_PORT_OFFSET = 17      # the launchers' own store; children rendezvous on MASTER_PORT + 1 + attempt


def run_attempts(commands, rank, world, master_addr, master_port, timeout_s, env=None, poll_s=0.2, log=None):
    """Run commands[0] as a child process on every rank; if it fails, hangs past `timeout_s` or fails on ANY other rank, kill
    this rank's child (its exact pid) and try commands[1], and so on.  Children get MASTER_PORT = master_port + 1 + attempt and
    BENCH_ATTEMPT = attempt in their environment; a command may be a pair (argv, {extra environment}).  -> (attempt index, stdout lines of this rank's child, [failure notes]);
    raises RuntimeError when every command failed."""
    from datetime import timedelta
    import torch.distributed as dist
    store = dist.TCPStore(master_addr, master_port + STORE_PORT_OFFSET, world, is_master=(rank == 0), timeout=timedelta(seconds=timeout_s + 120),
                          wait_for_workers=True)
    notes = []
    for i, cmd in enumerate(commands):
        extra_env = {}
        if isinstance(cmd, tuple):            # (argv, {environment of this attempt})
            cmd, extra_env = cmd
        child_env = dict(os.environ if env is None else env)
        child_env.update(extra_env)
        # torch.distributed.run tells its workers to rendezvous through the agent's own store; the children form a group of their own
        child_env.pop("TORCHELASTIC_USE_AGENT_STORE", None)
        child_env.update({"MASTER_ADDR": master_addr, "MASTER_PORT": str(master_port + 1 + i), "BENCH_ATTEMPT": str(i), "BENCH_CHILD": "1"})
        # the child's stderr goes to a file of its own: the tail of it is what explains a failed attempt in the result line
        # (a run on hardware nobody can log into afterwards) -- and is passed on to this process's stderr either way
        import tempfile
        errfile = tempfile.TemporaryFile(mode="w+", prefix="bench_attempt%d_rank%d_" % (i, rank))
        proc = subprocess.Popen(cmd, env=child_env, stdout=subprocess.PIPE, stderr=errfile, text=True)
        lines = []
        reader = threading.Thread(target=lambda: lines.extend(l.rstrip("\n") for l in proc.stdout), daemon=True)
        reader.start()
        deadline = time.monotonic() + timeout_s
        why = None
        while proc.poll() is None:
            if store.add("fail%d" % i, 0) > 0:
                why = "another rank's attempt failed"
            elif time.monotonic() > deadline:
                why = "no result after %.0f s" % timeout_s
            if why is not None:
                proc.kill()
                break
            time.sleep(poll_s)
        proc.wait()
        reader.join(5.0)
        if why is None and proc.returncode != 0:
            why = "exit code %d" % proc.returncode
        err_tail = ""
        try:
            errfile.seek(0)
            err_text = errfile.read()
            errfile.close()
            if err_text:
                import sys as _sys
                _sys.stderr.write(err_text)
                _sys.stderr.flush()
            err_lines = [l for l in err_text.splitlines() if l.strip()]
            err_tail = " | ".join(err_lines[-3:])[-400:]
        except Exception:
            pass
        if why is not None and err_tail and (proc.returncode not in (0, None, -9) or why.startswith("no result")):
            why += " (rank %d stderr: %s)" % (rank, err_tail)
            try:      # the rank whose child actually failed tells the others why (rank 0 writes the result line)
                if store.add("haserr%d" % i, 1) == 1:
                    store.set("errtext%d" % i, "rank %d stderr: %s" % (rank, err_tail))
            except Exception:
                pass
        if why is not None:
            store.add("fail%d" % i, 1)
        store.add("done%d" % i, 1)
        wait_until = time.monotonic() + timeout_s + 60
        while store.add("done%d" % i, 0) < world and time.monotonic() < wait_until:
            time.sleep(poll_s)
        failed = store.add("fail%d" % i, 0) > 0
        if not failed or i == len(commands) - 1:
            store.add("left", 1)
            while rank == 0 and store.add("left", 0) < world and time.monotonic() < wait_until:      # the store lives in rank 0
                time.sleep(poll_s)
        if not failed:
            return i, lines, notes
        reason = why or "failed on another rank"
        if "stderr:" not in reason:
            try:
                if store.add("haserr%d" % i, 0) > 0:
                    reason += " (" + store.get("errtext%d" % i).decode(errors="replace") + ")"
            except Exception:
                pass
        notes.append("attempt %d (%s): %s" % (i, " ".join(cmd[-4:]), reason))
        if log is not None:
            log(notes[-1])
    raise RuntimeError("every configuration failed: " + "; ".join(notes))
