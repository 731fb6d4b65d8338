"""ctypes binding of the C ABI declared in include/openmm_hip_kernels.h.

`load()` opens the product library openmm_amd/lib/libopenmm_hip_kernels.so (hipcc, gfx950) and
raises if it is missing -- there is no CPU fallback on the product path.  Tests that exercise
host-side logic without a GPU pass the path of the CPU-emulated twin explicitly
(tests/emu/_build/libopenmm_hip_kernels.so); nothing in the package does that on its own.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
PRODUCT_LIB = os.path.join(_HERE, "lib", "libopenmm_hip_kernels.so")

TILE = 32
ROW = 64
CHUNK_ROWS = 2
NL_STATE_INTS = 12


class NeighborList(C.Structure):
    _fields_ = [
        ("num_atoms", C.c_int), ("padded_atoms", C.c_int), ("max_chunks", C.c_int), ("pbc", C.c_int),
        ("cutoff", C.c_double), ("padding", C.c_double), ("box", C.c_double * 6),
        ("posq", C.c_void_p), ("posq_ref", C.c_void_p), ("atom_of_slot", C.c_void_p), ("slot_of_atom", C.c_void_p),
        ("excl_start", C.c_void_p), ("excl_atoms", C.c_void_p), ("excl_block_range", C.c_void_p), ("state", C.c_void_p),
        ("block_center", C.c_void_p), ("block_half", C.c_void_p), ("chunk_info", C.c_void_p),
        ("row_j", C.c_void_p), ("row_mask", C.c_void_p), ("excl_slot_start", C.c_void_p), ("excl_slots", C.c_void_p),
        ("cell_start", C.c_void_p), ("cell_blocks", C.c_void_p), ("cell_boxes", C.c_void_p), ("cell_meta", C.c_void_p), ("max_cells", C.c_int), ("cell_min_blocks", C.c_int),
        ("first_block", C.c_int), ("owned_blocks", C.c_int), ("posq_rel", C.c_void_p),
        ("dd_mode", C.c_int), ("dd_half_shell", C.c_int), ("dd_eval_slot0", C.c_int), ("dd_eval_slot1", C.c_int), ("pos_wire", C.c_void_p), ("pos_scatter", C.c_void_p), ("posq_rel_lo", C.c_void_p),
        ("num_active_ranges", C.c_int), ("active_range", C.c_int * 8), ("wire_ref", C.c_void_p), ("dd_guard_atom", C.c_void_p),
        ("dd_warn", C.c_uint), ("dd_max", C.c_uint), ("dd_flags", C.c_void_p), ("dd_ranks", C.c_int), ("dd_slots_per_rank", C.c_int), ("dd_trailer_slot", C.c_int),
        ("chunk_info_inner", C.c_void_p), ("row_j_inner", C.c_void_p), ("row_mask_inner", C.c_void_p), ("block_runs", C.c_void_p),
        ("posq_ref_inner", C.c_void_p), ("inner_padding", C.c_double),
    ]


class NonbondedParams(C.Structure):
    _fields_ = [
        ("ewald", C.c_int), ("use_switch", C.c_int), ("ewald_alpha", C.c_double), ("krf", C.c_double), ("crf", C.c_double),
        ("switch_distance", C.c_double), ("direct_grid", C.c_int), ("ljpme", C.c_int), ("dispersion_alpha", C.c_double),
    ]


class Pme(C.Structure):
    _fields_ = [
        ("nx", C.c_int), ("ny", C.c_int), ("nz", C.c_int), ("alpha", C.c_double), ("box", C.c_double * 6),
        ("moduli_x", C.c_void_p), ("moduli_y", C.c_void_p), ("moduli_z", C.c_void_p),
        ("eterm", C.c_void_p), ("grid_real", C.c_void_p), ("grid_complex", C.c_void_p),
        ("twiddle_x", C.c_void_p), ("twiddle_y", C.c_void_p), ("twiddle_z", C.c_void_p), ("spread_mode", C.c_int),
        ("grid_precleared", C.c_int), ("fft_mode", C.c_int),
        ("excl_start", C.c_void_p), ("excl_atoms", C.c_void_p), ("atom_of_slot", C.c_void_p), ("pos", C.c_void_p),
        ("charge", C.c_void_p), ("excl_periodic", C.c_int), ("phases", C.c_int), ("deterministic", C.c_int), ("max_charge", C.c_double), ("dispersion", C.c_int),
        ("dd_ranks", C.c_int), ("dd_rank", C.c_int), ("dd_halo", C.c_int), ("grid_complex2", C.c_void_p), ("comm", C.c_void_p), ("dd_error", C.c_void_p),
        ("tile_count", C.c_void_p), ("tile_blocks", C.c_void_p), ("tile_cap", C.c_int), ("max_tiles", C.c_int), ("block_center", C.c_void_p), ("block_half", C.c_void_p),
        ("dd_num_active_ranges", C.c_int), ("dd_active_range", C.c_int * 8),
    ]


PME_ALL, PME_SPREAD_ONLY, PME_AFTER_SPREAD, PME_INTERPOLATE_ONLY = 0, 1, 2, 3      # Pme.phases


class ValenceList(C.Structure):
    """ommhip_valence_list (include/openmm_hip_kernels.h)"""
    _fields_ = [("kind", C.c_int), ("num_terms", C.c_int), ("atoms", C.c_void_p), ("params", C.c_void_p), ("coefficients", C.c_double * 6), ("grids", C.c_void_p)]


class VmInstruction(C.Structure):
    _fields_ = [("op", C.c_int), ("arg", C.c_int), ("value", C.c_double)]


class VmStep(C.Structure):
    _fields_ = [("first", C.c_int), ("count", C.c_int), ("target", C.c_int), ("uses_random", C.c_int), ("force", C.c_void_p), ("draw", C.c_ulonglong)]


class VmState(C.Structure):
    _fields_ = [("num_atoms", C.c_int), ("num_per_dof", C.c_int), ("pos", C.c_void_p), ("vel", C.c_void_p), ("per_dof", C.c_void_p), ("globals", C.c_void_p),
                ("program", C.c_void_p), ("seed", C.c_ulonglong), ("sum_scratch", C.c_void_p), ("sum_result", C.c_void_p)]


class VmBonds(C.Structure):
    _fields_ = [("num_bonds", C.c_int), ("num_params", C.c_int), ("param_stride", C.c_int), ("periodic", C.c_int), ("atoms", C.c_void_p), ("params", C.c_void_p),
                ("program", C.c_void_p), ("energy_first", C.c_int), ("energy_count", C.c_int), ("deriv_first", C.c_int), ("deriv_count", C.c_int),
                ("globals", C.c_void_p), ("box", C.c_double * 6)]


class Ccma(C.Structure):
    """ommhip_ccma (include/openmm_hip_kernels.h)"""
    _fields_ = [("num_constraints", C.c_int), ("atoms", C.c_void_p), ("distance", C.c_void_p), ("delta", C.c_void_p), ("delta2", C.c_void_p),
                ("row_start", C.c_void_p), ("col", C.c_void_p), ("value", C.c_void_p), ("converged", C.c_void_p)]


class KernelError(RuntimeError):
    pass


_P, _I, _D, _Z = C.c_void_p, C.c_int, C.c_double, C.c_size_t
_D6 = C.POINTER(C.c_double)
# argument types of every entry point of include/openmm_hip_kernels.h (all return int)
SIGNATURES = {
    "device_count": [C.POINTER(C.c_int)],
    "set_device": [_I],
    "device_info": [_I, C.c_char_p, _I, C.POINTER(C.c_int), C.POINTER(C.c_size_t)],
    "malloc": [C.POINTER(C.c_void_p), _Z],
    "free": [_P],
    "host_malloc": [C.POINTER(C.c_void_p), _Z],
    "host_free": [_P],
    "memcpy_h2d": [_P, _P, _Z, _P],
    "memcpy_d2h": [_P, _P, _Z, _P],
    "memcpy_d2d": [_P, _P, _Z, _P],
    "memset": [_P, _I, _Z, _P],
    "stream_create": [C.POINTER(C.c_void_p)],
    "stream_create_priority": [C.POINTER(C.c_void_p), _I],
    "event_create_untimed": [C.POINTER(C.c_void_p)],
    "stream_destroy": [_P],
    "stream_sync": [_P],
    "device_sync": [C.c_int],
    "event_create": [C.POINTER(C.c_void_p)],
    "event_destroy": [_P],
    "event_record": [_P, _P],
    "event_sync": [_P],
    "event_elapsed_ms": [_P, _P, C.POINTER(C.c_float)],
    "stream_wait_event": [_P, _P],
    "positions_to_posq": [_P, _P, _P, _I, _D6, _P, _P],
    "set_slot_params": [_P, _P, _P, _P, _I, _P, _P, _P],
    "forces_to_double": [_P, _P, _I, _I, _P, _P],
    "add_forces_from_double": [_P, _P, _I, _I, _P, _P],
    "reduce_energy": [_P, _I, _P, _P],
    "nl_update": [C.POINTER(NeighborList), _P],
    "nl_step": [C.POINTER(NeighborList), _P, _P, _P],
    "nl_prepare": [C.POINTER(NeighborList), _P, _P, _P, _Z, _P, _Z, _P],
    "nl_rebuild_if_requested": [C.POINTER(NeighborList), _P],
    "force_front": [C.POINTER(NeighborList), C.POINTER(Pme), _I, _P, _P, _P, _P, _I, _I, _P],
    "pairs_with_fft": [C.POINTER(NeighborList), C.POINTER(NonbondedParams), _P, C.POINTER(Pme), _P, _P, _I, _I, _P],
    "fft_supported_size": [_I],
    "pme_build_eterm": [C.POINTER(Pme), _P],
    "pme_reciprocal": [C.POINTER(Pme), _P, _I, _P, _P, _I, _I, _P],
    "fft3d_r2c_c2r": [C.POINTER(Pme), _I, _P],
    "test_transpose_reduce": [_P, _P, _I, _P],
    "nb_direct": [C.POINTER(NeighborList), C.POINTER(NonbondedParams), _P, _P, _P, _I, _I, _P],
    "valence_forces": [_I, C.POINTER(ValenceList), _P, _P, _I, _P, _P, _I, _I, _P],
    "vm_per_dof": [C.POINTER(VmState), _I, C.POINTER(VmStep), _P],
    "vm_bond_forces": [C.POINTER(VmBonds), _P, _P, _I, _P, _P, _I, _I, _P],
    "vm_angle_forces": [C.POINTER(VmBonds), _P, _P, _I, _P, _P, _I, _I, _P],
    "forces_to_atom_order": [_P, _P, _I, _I, _P, _P],
    "settle": [_I, _P, _P, _P, _P, _P, _I, _P],
    "shake": [_I, _P, _P, _P, _P, _P, _I, _D, _I, _P],
    "constrain_clusters": [_I, _P, _P, _I, _P, _P, _P, _P, _P, _I, _D, _I, _P],
    "ccma_iteration": [C.POINTER(Ccma), _P, _P, _P, _I, _D, _I, _P],
    "ccma_iterations": [C.POINTER(Ccma), _P, _P, _P, _I, _D, _I, _P],
    "ewald_reciprocal": [_P, _P, _P, _I, _I, _D6, _D, _I, _I, _I, _P, _P, _P, _I, _I, _P],
}


class Kernels:
    """Thin checked wrapper: every call raises KernelError on a non-zero return code."""

    def __init__(self, path=None):
        path = path or PRODUCT_LIB
        if not os.path.exists(path):
            raise KernelError("HIP kernel library not found: %s (run `python -c 'import __graft_entry__ as g; g.build()'`)" % path)
        self.path = path
        self.lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
        self.lib.ommhip_error_string.restype = C.c_char_p
        # the ctypes mirrors above against the structs the library was compiled with
        self.lib.ommhip_struct_size.restype = C.c_size_t
        self.lib.ommhip_struct_size.argtypes = [C.c_int]
        for which, mirror in ((0, NeighborList), (1, NonbondedParams), (2, Pme), (6, Ccma), (7, ValenceList), (8, VmInstruction), (9, VmStep), (10, VmState), (11, VmBonds)):
            if self.lib.ommhip_struct_size(which) != C.sizeof(mirror):
                raise KernelError("%s: ctypes mirror of struct %d has %d bytes, the library's has %d -- openmm_amd/capi.py is out of date with include/openmm_hip_kernels.h"
                                  % (path, which, C.sizeof(mirror), self.lib.ommhip_struct_size(which)))

    def __getattr__(self, name):
        fn = getattr(self.lib, "ommhip_" + name)
        fn.restype = C.c_int
        fn.argtypes = SIGNATURES[name]

        def call(*args):
            rc = fn(*args)
            if rc != 0:
                raise KernelError("ommhip_%s failed: %d (%s)" % (name, rc, self.lib.ommhip_error_string(rc).decode()))
            return rc
        return call

    # ---- small conveniences used by tests and bench
    def malloc(self, nbytes):
        p = C.c_void_p()
        self.__getattr__("malloc")(C.byref(p), C.c_size_t(nbytes))
        return p

    def upload(self, array, stream=None):
        import numpy as np
        array = np.ascontiguousarray(array)
        p = self.malloc(max(array.nbytes, 16))
        self.memcpy_h2d(p, array.ctypes.data_as(C.c_void_p), C.c_size_t(array.nbytes), stream)
        self.stream_sync(stream)
        return p

    def download(self, ptr, shape, dtype, stream=None):
        import numpy as np
        out = np.empty(shape, dtype=dtype)
        self.memcpy_d2h(out.ctypes.data_as(C.c_void_p), ptr, C.c_size_t(out.nbytes), stream)
        self.stream_sync(stream)
        return out


def load(path=None):
    return Kernels(path)
