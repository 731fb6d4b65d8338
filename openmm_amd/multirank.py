"""Cross-rank bookkeeping of bench.py (one process per GPU, launched by torch.distributed.run).

The hot path does not shard across GPUs in this round: each rank runs an independent replica of the
workload, so there is no data-path collective.  The only cross-rank operations are the barriers around
the timed region and this MAX-over-ranks reduction of the elapsed time."""


def aggregate_throughput(elapsed_s, steps, dt_fs, group=None, device="cuda"):
    """-> (whole-job ns/day over all ranks, ms per step of the slowest rank)."""
    world = 1
    if group is not None:
        import torch
        import torch.distributed as dist
        t = torch.tensor([elapsed_s], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        elapsed_s = float(t.item())
        world = dist.get_world_size(group)
    ns_per_day_one = dt_fs * 1e-6 * steps / elapsed_s * 86400.0
    return ns_per_day_one * world, 1e3 * elapsed_s / steps
