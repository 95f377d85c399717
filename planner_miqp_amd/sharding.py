"""Multi-GPU sharding of independent planning instances: instance b -> rank b mod G,
no data-path collective; results are gathered once (fixed-size records) over torch.distributed."""


def shard_indices(n_total, rank, world):
    return list(range(rank, n_total, world))


def gather_counts(local_vals, device=None):
    """all-gathers a small list of floats from every rank; returns list of lists (rank-major).
    Uses the already initialised default process group (RCCL on GPUs, gloo on CPU)."""
    import torch
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized():
        return [list(local_vals)]
    t = torch.tensor(list(local_vals), dtype=torch.float64, device=device)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [o.cpu().tolist() for o in out]
