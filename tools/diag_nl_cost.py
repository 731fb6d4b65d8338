#!/usr/bin/env python
"""Per-i-block cost of the neighbour-list build on the DHFR-like workload (run on the GPU box)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openmm_amd import harness as H, testsystems as T

H.load_hip_platform()
plugin = C.CDLL(os.path.join(H.LIB_DIR, "libOpenMMHIP.so"))
w = T.dhfr_like(seed=1)
s, nb = w.build()
integ = H.Integrator(H.LANGEVIN_MIDDLE, 0.002, 300.0, 1.0, seed=1, constraintTolerance=1e-5)
c = H.Context(s, integ, "HIP")
c.setPositions(w.positions)
c.setVelocities(w.velocities)
integ.step(40)
c.getState(getEnergy=True)
n = 1024
ticks, cand = (C.c_float * n)(), (C.c_float * n)()
nb_blocks = plugin.ommhip_plugin_nl_block_costs(ticks, cand, n)
t = np.array(ticks[:nb_blocks]); k = np.array(cand[:nb_blocks])
print("blocks", nb_blocks, "ticks: mean %.0f median %.0f p90 %.0f p99 %.0f max %.0f" % (t.mean(), np.median(t), np.percentile(t, 90), np.percentile(t, 99), t.max()))
print("candidates: mean %.0f median %.0f max %.0f" % (k.mean(), np.median(k), k.max()))
order = np.argsort(-t)[:12]
for b in order:
    print("  block %4d ticks %8.0f candidates %4.0f" % (b, t[b], k[b]))
print("corr(ticks, candidates) %.2f" % np.corrcoef(t, k)[0, 1])
for lo, hi in ((0, 80), (80, 160), (160, 400), (400, 737)):
    m = (np.arange(nb_blocks) >= lo) & (np.arange(nb_blocks) < hi)
    print("  blocks %3d-%3d: mean ticks %.0f mean cand %.0f" % (lo, hi, t[m].mean(), k[m].mean()))
