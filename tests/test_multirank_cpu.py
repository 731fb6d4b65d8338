"""world_size-2 `gloo` test of the N>1 bench path on CPU: replicas are independent (no data-path collective);
the only cross-rank operations are the barrier and the MAX-over-ranks of the timed region."""
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
