"""Multi-GPU use of the solve path, one process per GPU under torch.distributed (backend "nccl" is
RCCL on ROCm, "gloo" on CPU):

  * independent planning instances: instance b -> rank b mod G, no data-path collective (``solve_sharded``), results gathered
    once as fixed-size records;
  * ONE hard instance split over the ranks (C1): ``split_solve`` - the library partitions the tree, the ranks exchange
    {incumbent | owner, bound, done, time-up} by an all-reduce(min) once per round and the owner broadcasts the solution.
    The exchange runs over RCCL inside the library (``init_rccl_comm``) or over any torch.distributed group
    (``torch_exchange``, used by the CPU tests with gloo).
"""
import ctypes as C

_SIGN = 1 << 63


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


def solve_sharded(wrappers, rank=None, world=None):
    """every rank holds the same list of wrappers (one per instance, parameters set); rank r solves the instances
    r, r + G, ... as one batch on its device and the (status, objective, gap) records are gathered on every rank.
    Returns the list of records in instance order."""
    import torch.distributed as dist
    from .wrapper import solve_batch
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
    mine = shard_indices(len(wrappers), rank, world)
    sts = solve_batch([wrappers[b] for b in mine]) if mine else []
    rec = []
    for b, st in zip(mine, sts):
        pr = wrappers[b].getSolutionProperties()
        rec.append((b, int(st), float(pr.objective), float(pr.gap), int(pr.status)))
    if dist.is_initialized() and world > 1:
        allrec = [None] * world
        dist.all_gather_object(allrec, rec)
    else:
        allrec = [rec]
    out = [None] * len(wrappers)
    for part in allrec:
        for r in part:
            out[r[0]] = dict(status=r[1], objective=r[2], gap=r[3], cplex_status=r[4])
    return out


def torch_exchange(group=None, device=None):
    """miqp_exchange_fn over torch.distributed: op 0 = in-place all-reduce(min) over unsigned 64-bit words (the sign bit is
    flipped so that the signed MIN of torch orders them as unsigned), op 1 = broadcast of bytes from ``root``.
    Keep the returned object alive while the library may call it."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from .wrapper import EXCHANGE_FN

    def fn(user, op, buf, count, root):
        try:
            if op == 0:
                a = np.ctypeslib.as_array(C.cast(buf, C.POINTER(C.c_uint64)), shape=(count,))
                if dist.is_initialized() and dist.get_world_size(group) > 1:
                    t = torch.from_numpy((a ^ np.uint64(_SIGN)).view(np.int64).copy())
                    if device is not None:
                        t = t.to(device)
                    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
                    a[:] = t.cpu().numpy().view(np.uint64) ^ np.uint64(_SIGN)
            else:
                a = np.ctypeslib.as_array(C.cast(buf, C.POINTER(C.c_uint8)), shape=(count,))
                if dist.is_initialized() and dist.get_world_size(group) > 1:
                    t = torch.from_numpy(a.copy())
                    if device is not None:
                        t = t.to(device)
                    dist.broadcast(t, src=root, group=group)
                    a[:] = t.cpu().numpy()
            return 0
        except Exception as e:  # the library turns a failed exchange into FAILED_SEG_FAULT
            print("[sharding] exchange failed:", e)
            return -1
    return EXCHANGE_FN(fn)


def init_rccl_comm(device=None):
    """creates the library's RCCL communicator over the ranks of the default process group: rank 0 draws the id, the group
    carries it to the others (any backend), every rank initialises on ``device`` (default: LOCAL device of the process)"""
    import torch
    import torch.distributed as dist
    from .wrapper import load_library
    L = load_library()
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    uid = C.create_string_buffer(128)
    if rank == 0 and L.miqp_comm_unique_id(uid) != 0:
        raise RuntimeError("ncclGetUniqueId failed")
    box = [uid.raw]
    if world > 1:
        dist.broadcast_object_list(box, src=0)
    if device is None:
        device = torch.cuda.current_device() if torch.cuda.is_available() else -1
    rc = L.miqp_comm_init(world, rank, box[0], int(device))
    if rc != 0:
        raise RuntimeError("miqp_comm_init failed: %d" % rc)
    return world, rank


def split_solve(wrapper, exchange=None, timestamp=0.0):
    """``wrapper`` (same parameters on every rank) solved with its tree split over the ranks of the default group"""
    import torch.distributed as dist
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    return wrapper.callCplexSplit(world, rank, exchange, timestamp)
