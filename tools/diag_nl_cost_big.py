#!/usr/bin/env python
"""Per-i-block cost of the neighbour-list build on a big water box, cell-binned vs full block scan (run on the GPU box)."""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openmm_amd import harness as H, testsystems as T
H.load_hip_platform()
plugin = C.CDLL(os.path.join(H.LIB_DIR, "libOpenMMHIP.so"))
n_side = int(sys.argv[1]) if len(sys.argv) > 1 else 69
w = T.water_box(n_side, seed=1)
for mode, env in (("cells", "0"), ("full scan", "100000000")):
    os.environ["OPENMM_HIP_NL_CELL_MIN_BLOCKS"] = env
    s, nb = w.build()
    integ = H.Integrator(H.VERLET, 0.0005)
    c = H.Context(s, integ, "HIP")
    c.setPositions(w.positions)
    t0 = time.perf_counter(); c.getState(getForces=True); t1 = time.perf_counter()
    n = 40000
    ticks, cand = (C.c_float * n)(), (C.c_float * n)()
    nblk = plugin.ommhip_plugin_nl_block_costs(ticks, cand, n)
    t = np.array(ticks[:nblk]); k = np.array(cand[:nblk])
    print("%-10s blocks %d  ticks mean %.0f p99 %.0f max %.0f sum %.3g | candidates mean %.1f max %.0f | first evaluation %.1f ms" % (
        mode, nblk, t.mean(), np.percentile(t, 99), t.max(), t.sum(), k.mean(), k.max(), 1e3 * (t1 - t0)))
    c.close()
