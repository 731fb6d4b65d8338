"""Which pair explains the largest HIP-vs-Reference force difference?  Runs the driver's DHFR protocol (warm-up + steps) a few times,
compares the final forces with the Reference platform and, for the worst atoms, lists the neighbours whose double-precision
distance lies within 1e-6 nm of the cutoff together with the size of the truncated pair force there (the jump a pair counted on
the other side of the cutoff produces).  usage: diag_parity_outlier.py [runs] [steps]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from scipy.spatial import cKDTree
from scipy.special import erfc
from openmm_amd import harness as H, testsystems as T

EMU = os.environ.get("BENCH_EMULATED") == "1"
H.load_hip_platform(emulated=EMU)
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 3
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 25
w = T.dhfr()
alpha = float(np.sqrt(-np.log(2 * w.ewald_tol)) / w.cutoff)
L = np.diag(np.asarray(w.box, float))
for run in range(runs):
    system, nb = w.build()
    integ = H.Integrator(H.LANGEVIN_MIDDLE, 0.002, 300.0, 1.0, seed=1, constraintTolerance=1e-5)
    ctx = H.Context(system, integ, "HIP")
    ctx.setPositions(w.positions)
    ctx.applyConstraints(1e-5)
    ctx.setVelocities(w.velocities)
    integ.step(steps + run)
    end = ctx.getState(getPositions=True, getForces=True)
    ctx.close()
    rsys, rnb = w.build()
    rctx = H.Context(rsys, H.Integrator(H.VERLET, 0.001), "Reference")
    rctx.setPositions(end.positions)
    f_ref = rctx.getState(getForces=True).forces
    rctx.close()
    pos, f_hip = end.positions, end.forces
    rms = np.sqrt((f_ref ** 2).sum(1).mean())
    rel = np.linalg.norm(f_hip - f_ref, axis=1) / np.maximum(np.linalg.norm(f_ref, axis=1), rms)
    wrapped = np.mod(pos, L[None, :]); wrapped[wrapped >= L[None, :]] = 0
    tree = cKDTree(wrapped, boxsize=L)
    print("run %d (%d steps): max %.3g  rms force %.1f" % (run, steps + run, rel.max(), rms), flush=True)
    for a in np.argsort(rel)[::-1][:2]:
        nbrs = np.array(tree.query_ball_point(wrapped[a], w.cutoff + 1e-5))
        d = pos[nbrs] - pos[a]; d -= np.round(d / L) * L
        r = np.linalg.norm(d, axis=1)
        near = np.abs(r - w.cutoff) < 1e-6
        dF = np.linalg.norm(f_hip[a] - f_ref[a])
        desc = []
        for j, rr in zip(nbrs[near], r[near]):
            qq = 138.935456 * w.charge[a] * w.charge[j]
            ar = alpha * rr
            jump = abs(qq) * (erfc(ar) + 2 * ar / np.sqrt(np.pi) * np.exp(-ar * ar)) / rr ** 2
            desc.append("j=%d r-rc=%+.2e |F_pair(rc)|=%.4f" % (j, rr - w.cutoff, jump))
        print("   atom %d rel %.3g |dF| %.4f  near-cutoff pairs: %s" % (a, rel[a], dF, "; ".join(desc) if desc else "none"))
