"""`-m gpu`: the domain-decomposed path on real hardware, as far as ONE GPU allows.

* one rank with a real RCCL communicator: every collective of the step goes through librccl (all-gather, grouped
  send/recv all-to-all, ring exchange -- with itself), on the plugin's stream, and the decomposed kernels (owned-slot pair
  kernel, slab PME with remapped transposes, trailer momentum) must reproduce the ordinary single-GPU run;
* two processes sharing the GPU (the reference's "one device listed twice" trick, platforms/cuda/tests/TestCudaNonbondedForce.cpp:37-96;
  RCCL refuses two ranks on one device, so the collectives go through the host-staged callback transport over gloo).
The 8-GPU run itself is the driver's (bench.py --gpus 8)."""
import os

import numpy as np
import pytest

from openmm_amd import harness as H, testsystems as T, multirank as MR

pytestmark = pytest.mark.gpu


def _run(w, props, steps, seed=7):
    system, nb = w.build()
    integ = H.Integrator(H.LANGEVIN_MIDDLE, 0.002, 300.0, 1.0, seed=seed)
    ctx = H.Context(system, integ, "HIP", props)
    ctx.setPositions(w.positions)
    ctx.applyConstraints(1e-6)
    ctx.setVelocitiesToTemperature(300.0, 3)
    st0 = ctx.getState(getForces=True, getEnergy=True)
    integ.step(steps)
    st1 = ctx.getState(getPositions=True, getVelocities=True, getEnergy=True, getForces=True)
    info = ctx.getPlatformProperty("CommId")
    ctx.close()
    return st0, st1, info


@pytest.mark.parametrize("n_side,grid", [(10, 32), (24, 0)])
def test_one_rank_with_rccl_reproduces_the_single_gpu_run(n_side, grid):
    H.load_hip_platform()
    w = T.water_box(n_side, seed=5)
    if grid:
        w.pme_params = (float(np.sqrt(-np.log(2 * w.ewald_tol)) / w.cutoff), grid, grid, grid)
    w.cm_remover = True
    one0, one1, _ = _run(w, {}, 20)
    dd0, dd1, transport = _run(w, {"Ranks": "1", "Rank": "0", "CommId": MR.new_rccl_id()}, 20)
    assert transport == "rccl"
    rms = np.sqrt((one0.forces ** 2).sum(1).mean())
    err = np.abs(dd0.forces - one0.forces).max() / rms
    print("one rank over RCCL vs plain: force max diff / rms %.3g, E %.6f vs %.6f" % (err, dd0.potentialEnergy, one0.potentialEnergy))
    assert err < 3e-5
    assert abs(dd0.potentialEnergy - one0.potentialEnergy) < 1e-6 * abs(one0.potentialEnergy) + 0.05
    # 20 steps: float32 force noise (different slot order, different summation order) grows, but slowly
    assert np.abs(dd1.positions - one1.positions).max() < 2e-5
    assert abs(dd1.kineticEnergy - one1.kineticEnergy) < 1e-4 * one1.kineticEnergy


def test_two_ranks_sharing_the_gpu_reproduce_the_single_gpu_run(tmp_path):
    from test_multirank_cpu import _run_dd_child
    out = _run_dd_child(tmp_path, False, 0, 10, 29561)
    print(out)
