"""Host time of the order computation of a decomposed run (HipContext::computeOrderDecomposed) for the bench's 985 527-atom water box:
the work every rank does at a re-sort, off the step but bounding how often a run may re-sort.  Runs without a GPU (the CPU emulator's
twin of the plugin); also prints what the decomposition looks like (halo mode, half-shell, slots a rank converts per step).

    python tools/time_resort_host.py [ranks=8] [tiles_per_side=3] [repeats=3]
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openmm_amd import harness as H, testsystems as T

ranks = int(sys.argv[1]) if len(sys.argv) > 1 else 8
tiles = int(sys.argv[2]) if len(sys.argv) > 2 else 3
repeats = int(sys.argv[3]) if len(sys.argv) > 3 else 3
w = T.water_tiled(tiles)
system, nb = w.build()
H.lib()
plugin_path = os.path.join(H.EMU_DIR, "libOpenMMHIP.so") if os.environ.get("BENCH_EMULATED", "1") == "1" else os.path.join(H.LIB_DIR, "libOpenMMHIP.so")
plugin = C.CDLL(plugin_path)
plugin.ommhip_plugin_time_decomposed_order.restype = C.c_double
xyz = np.ascontiguousarray(w.positions, dtype=np.float64)
nx = 64 * tiles
for rank in (0, ranks // 2):
    info = (C.c_longlong * 12)()
    ms = plugin.ommhip_plugin_time_decomposed_order(system.h, xyz.ctypes.data_as(C.POINTER(C.c_double)), ranks, rank, nx, repeats, info)
    print("rank %d of %d, %d atoms: order computed in %.1f ms; halo mode %d, half-shell %d, slots per rank %d, converted per step %d (%.2f x), pair partners from below %d, drift margin %.3f nm"
          % (rank, ranks, w.num_atoms, ms, info[0], info[1], info[2], info[3], info[3] / max(1, info[2]), info[4], info[5] * 1e-6))
    print("    largest bounding-box edge of its %d blocks: median %.2f nm, 99th percentile %.2f nm, largest %.2f nm" % (info[9], info[6] * 1e-3, info[7] * 1e-3, info[8] * 1e-3))
