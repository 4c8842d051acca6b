"""Multi-rank plumbing of bench.py: one process per GPU, sequences sharded across ranks, no data-path collective.
torch.distributed is used for the barrier around the timed region and the MAX of the per-rank elapsed times only
(backend "nccl" = RCCL on the GPU box, "gloo" in the CPU tests)."""
import os


def rank_info():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard_units(n_units, rank, world):
    """Units (sequences / frames) owned by `rank` when `n_units` are dealt round-robin: a fixed global work list can be
    split this way (strong scaling); bench.py gives every rank its own `--frames` units instead (weak scaling)."""
    return list(range(rank, n_units, world))


def init(backend, rank, world, device=None):
    import torch.distributed as dist
    if world <= 1:
        return None
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    kw = {"device_id": device} if device is not None else {}
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return dist


def barrier(dist, sync=None):
    if dist is not None:
        dist.barrier()
    if sync is not None:
        sync()


def max_elapsed(dist, elapsed, device="cpu"):
    """MAX over ranks of the per-rank elapsed time of the timed region."""
    if dist is None:
        return float(elapsed)
    import torch
    t = torch.tensor([elapsed], device=device, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_int(dist, value, device="cpu"):
    """SUM over ranks of an integer (units owned, ranks present)."""
    if dist is None:
        return int(value)
    import torch
    t = torch.tensor([int(value)], device=device, dtype=torch.int64)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())


def count_ranks(dist, device="cpu"):
    """How many ranks took part in the collective (what the bench line reports as ranks_seen)."""
    return sum_int(dist, 1, device)


def job_throughput(units_per_rank_per_step, steps, world, elapsed_max):
    """Whole-job units/s: every rank processed units_per_rank_per_step * steps units in at most elapsed_max seconds."""
    return units_per_rank_per_step * steps * world / elapsed_max
