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
"""
import ctypes as C
import os

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


def gloo_all_gather_callback(dist, group=None):
    """-> "callback:<fn>:<user>" CommId of the host-staged transport; the all-gather runs on `group` (a gloo group)."""
    import numpy as np
    import torch
    world = dist.get_world_size(group)
    proto = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)

    def all_gather(user, send, recv, nbytes):
        try:
            src = np.ctypeslib.as_array(C.cast(send, C.POINTER(C.c_uint8)), shape=(nbytes,))
            dst = np.ctypeslib.as_array(C.cast(recv, C.POINTER(C.c_uint8)), shape=(world * nbytes,))
            out = torch.empty(world * nbytes, dtype=torch.uint8)
            dist.all_gather_into_tensor(out, torch.from_numpy(src.copy()), group=group)
            dst[:] = out.numpy()
            return 0
        except Exception as e:      # never let an exception cross the C boundary
            print("gloo all-gather callback failed:", e, flush=True)
            return 1
    fn = proto(all_gather)
    _KEEP_ALIVE.append(fn)
    return "callback:%d:0" % C.cast(fn, C.c_void_p).value


def domain_properties(dist, transport="rccl", device_index=None, group=None, emulated=False):
    """Platform properties of this rank's Context in a decomposed run."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    props = {"Ranks": str(world), "Rank": str(rank)}
    if device_index is not None:
        props["DeviceIndex"] = str(device_index)
    if transport == "rccl":
        props["CommId"] = rccl_comm_id(dist, emulated)
    elif transport == "gloo":
        props["CommId"] = gloo_all_gather_callback(dist, group)
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
