"""Multi-GPU plumbing of the frame-parallel mode: one process per GPU, frames
sharded round-robin, no data-path collective.  torch.distributed (RCCL on GPUs, gloo in CPU tests)
carries only the barrier and the max-over-ranks of the timed region."""
import os


def env():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def frames_of_rank(n_frames, rank, world):
    """Frame n belongs to GPU n mod world (config C4 of SURVEY §8d)."""
    return list(range(rank, n_frames, world))


def barrier(world):
    if world > 1:
        import torch.distributed as dist
        dist.barrier()


def max_over_ranks(seconds, world, device="cpu"):
    if world == 1:
        return float(seconds)
    import torch
    import torch.distributed as dist
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def job_throughput(units_per_rank_step, steps, seconds_max, world):
    """Whole-job rate: units all ranks processed / slowest rank's time (weak scaling)."""
    return world * units_per_rank_step * steps / seconds_max
