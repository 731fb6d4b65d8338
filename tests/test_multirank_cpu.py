"""world_size-2 `gloo` tests on CPU.
(1) The N>1 bench path: replicas are independent (no data-path collective); the only cross-rank operations are the barrier
    and the MAX-over-ranks of the timed region.
(2) Force decomposition through the C ABI (first building block of DESIGN.md (e)): two ranks build the neighbour list for one
    half of the i-blocks each (ommhip_neighbor_list.first_block / owned_blocks), run the pair kernel, and all-reduce their
    fixed-point force buffers -- the sum must be bit for bit the single-rank buffer.  Runs on the CPU SIMT emulator build of
    the kernels (tests/emu)."""
import os
import subprocess
import sys

from conftest import ROOT

CHILD = r'''
import os, sys, time
sys.path.insert(0, %r)
import torch, torch.distributed as dist
from openmm_amd.multirank import aggregate_throughput
dist.init_process_group(backend="gloo")
rank = dist.get_rank()
elapsed = 1.0 + rank          # rank 1 is slower
value, ms = aggregate_throughput(elapsed, steps=1000, dt_fs=2.0, group=dist.group.WORLD, device="cpu")
if rank == 0:
    # both ranks did 1000 steps; the job took max(1, 2) = 2 s -> 2 replicas * 2 fs * 1000 / 2 s
    expect = 2 * 2.0e-6 * 1000 / 2.0 * 86400
    assert abs(value - expect) < 1e-9 * expect, (value, expect)
    assert abs(ms - 2.0) < 1e-12
    print("OK", value)
dist.destroy_process_group()
'''


def test_two_rank_aggregate_on_gloo(tmp_path):
    script = tmp_path / "child.py"
    script.write_text(CHILD % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29533", str(script)],
                         capture_output=True, text=True, timeout=600, env=env)
    assert "OK" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


DECOMP_CHILD = r'''
import os, sys
sys.path.insert(0, %r)
sys.path.insert(0, os.path.join(%r, "tests"))
import numpy as np, torch, torch.distributed as dist
from openmm_amd import capi
import kernel_cases as KC
from oracle import nonbonded as ONB
dist.init_process_group(backend="gloo")
rank, world = dist.get_rank(), dist.get_world_size()
K = capi.load(%r)
EXCL = [(i, i + 1) for i in range(0, 600, 3)] + [(i, i + 2) for i in range(0, 600, 3)]
n, cutoff, L = 1500, 0.7, 3.4
blocks = (n + 31) // 32
for compact in (False, True):
    # every rank: the whole evaluation (reference for the bit-for-bit comparison) ...
    full = KC.run_direct_space(K, n, ONB.PME, cutoff, L, EXCL, compact=compact)
    whole = KC.LAST_FIXED_POINT_FORCES.copy()
    # ... and its share of the i-blocks
    first = blocks * rank // world
    count = blocks * (rank + 1) // world - first
    part = KC.run_direct_space(K, n, ONB.PME, cutoff, L, EXCL, compact=compact, block_range=(first, count))
    mine = torch.from_numpy(KC.LAST_FIXED_POINT_FORCES.copy())
    energy = torch.tensor([part[1]], dtype=torch.float64)
    assert part[4][1] < full[4][1]                       # fewer chunks than the full list
    dist.all_reduce(mine, op=dist.ReduceOp.SUM)
    dist.all_reduce(energy, op=dist.ReduceOp.SUM)
    assert np.array_equal(mine.numpy(), whole), "summed fixed-point forces differ from the single-rank buffer"
    assert abs(float(energy) - full[1]) < 1e-9 * abs(full[1]) + 1e-6
if rank == 0:
    print("OK")
dist.destroy_process_group()
'''


def test_two_rank_force_decomposition_is_bit_exact(tmp_path):
    import pytest
    from conftest import EMU_BUILD
    emu_lib = os.path.join(EMU_BUILD, "libopenmm_hip_kernels.so")
    if not os.path.exists(emu_lib):
        pytest.skip("emulated kernel library not built (run __graft_entry__.build())")
    script = tmp_path / "decomp_child.py"
    script.write_text(DECOMP_CHILD % (ROOT, ROOT, emu_lib))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29541", str(script)],
                         capture_output=True, text=True, timeout=900, env=env)
    assert "OK" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]
