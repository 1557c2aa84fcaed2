"""Data-parallel glue: one process per GPU, torch.distributed (backend "nccl" == RCCL on ROCm, over xGMI).

The reference is single-process; the only exchange a data-parallel DAE step needs is the sum of the flat
gradient [dW | dbh | dbv].  It is one bucket (20 MB at 10000x500) so a single all-reduce per step
moves it; the optimizer kernel then applies ``grad_scale = 1/world`` (mean of the per-rank mean losses ==
the global-batch mean for equal shards).  Mining is per-rank ("local mining", SURVEY 8e mode ii)."""
from __future__ import annotations

import os

import numpy as np


def is_initialized():
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized()


def world_size():
    import torch.distributed as dist
    return dist.get_world_size() if is_initialized() else 1


def rank():
    import torch.distributed as dist
    return dist.get_rank() if is_initialized() else 0


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT / LOCAL_RANK."""
    import torch
    import torch.distributed as dist
    if is_initialized():
        return
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    if ws <= 1:
        return
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    dist.init_process_group(backend=backend, rank=int(os.environ["RANK"]), world_size=ws)


def allreduce_sum_(flat):
    """In-place sum of the flat gradient over all ranks (RCCL all-reduce on the current stream)."""
    import torch.distributed as dist
    if is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    return flat


def allreduce_max_float(x):
    import torch
    import torch.distributed as dist
    if not (is_initialized() and dist.get_world_size() > 1):
        return float(x)
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([float(x)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def broadcast_array(a, src=0):
    """Broadcast a NumPy array from ``src`` so that every rank starts from identical parameters."""
    import torch
    import torch.distributed as dist
    if not (is_initialized() and dist.get_world_size() > 1):
        return a
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    dist.broadcast(t, src=src)
    return t.cpu().numpy()


def barrier():
    import torch.distributed as dist
    if is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def shard_bounds(start, stop, world, r):
    """Contiguous shard [lo, hi) of the global mini-batch [start, stop) owned by rank r."""
    per = -(-(stop - start) // world)
    lo = min(stop, start + r * per)
    return lo, min(stop, lo + per)
