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


class ShardedExchange:
    """The exchange step of a data-parallel DAE step, built for point-to-point xGMI rather than a switch: instead of one
    fp32 all-reduce of the flat gradient (every rank sends AND receives 2*(R-1)/R * 20.7 MB at 10000x500) the W part is

        reduce-scattered by row chunks   -> rank r receives the rank-summed gradient of ITS rows      ((R-1)/R * 20.7 MB fp32,
                                                                                                         or 10.4 MB as bf16)
        updated by the optimizer there   -> sharded master weights / optimizer slots (dae_plan_apply_rows)
        all-gathered as the bf16 shadow  -> every rank receives W_lo                                  ((R-1)/R * 10.4 MB)

    and Wt_lo is rebuilt locally (dae_plan_refresh_wt).  The bias part (10.6 K floats) is all-reduced and applied on every
    rank.  Per step and rank that is 3/4 (fp32 gradients) or 1/2 (bf16 gradients) of the all-reduce's traffic, in two collectives
    that RCCL runs as direct exchanges over all 7 links.  Only the owner of a row block holds its current fp32 master;
    ``gather_master`` rebuilds the full W where it is needed (get_params / checkpoint).

    ``collective_ms`` accumulates the time of the collectives (events on the current stream when it is a CUDA stream)."""

    def __init__(self, eng, grad_dtype="fp32"):
        import torch
        import torch.distributed as dist
        assert is_initialized(), "torch.distributed is not initialised"
        self.eng, self.torch, self.dist = eng, torch, dist
        self.world, self.rank = dist.get_world_size(), dist.get_rank()
        assert eng.dp_world == self.world, "create the Engine with dp_world = world size (%d != %d)" % (eng.dp_world, self.world)
        assert grad_dtype in ("fp32", "bf16")
        self.grad_dtype = grad_dtype
        c, Hp = eng.chunk_rows, eng.Hp
        self.f0 = min(eng.Fp, self.rank * c)
        self.f1 = min(eng.Fp, (self.rank + 1) * c)
        self.n_w = eng.rows_alloc * Hp
        gdt = torch.float32 if grad_dtype == "fp32" else torch.bfloat16
        self.rs_out = torch.zeros(c * Hp, dtype=gdt, device=eng.device)
        self.rs_f32 = self.rs_out if grad_dtype == "fp32" else torch.zeros(c * Hp, dtype=torch.float32, device=eng.device)
        self.my_lo = torch.zeros((c, Hp), dtype=eng.td, device=eng.device)
        self.collective_ms = 0.0
        self.steps = 0
        self._ev = None
        if eng.device.type == "cuda":
            self._ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True),
                        torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        self._pending = None

    def step(self, grad_scale):
        """Call after eng.train_step(phase=1): exchange + sharded update + shadow rebuild.  grad_scale multiplies the rank-SUMMED
        gradient (1/world for equal shards)."""
        torch, dist, eng = self.torch, self.dist, self.eng
        Hp, c = eng.Hp, eng.chunk_rows
        gw = eng.grad[:self.n_w]
        bias = eng.grad[eng.Fp * Hp:eng.Fp * Hp + Hp + eng.Fp]
        if self._ev:
            self._ev[0].record()
        if self.grad_dtype == "fp32":
            dist.reduce_scatter_tensor(self.rs_out, gw, op=dist.ReduceOp.SUM)
        else:
            dist.reduce_scatter_tensor(self.rs_out, gw.to(torch.bfloat16), op=dist.ReduceOp.SUM)
            self.rs_f32.copy_(self.rs_out)
        dist.all_reduce(bias, op=dist.ReduceOp.SUM)
        if self._ev:
            self._ev[1].record()
        eng.adam_t += 1 if eng.opt == "adam" else 0
        eng.apply_rows(self.rs_f32, self.f0, self.f1, grad_scale=grad_scale, update_bias=True)
        self.my_lo.copy_(eng.W_lo_full[self.rank * c:(self.rank + 1) * c])
        if self._ev:
            self._ev[2].record()
        dist.all_gather_into_tensor(eng.W_lo_full.view(-1), self.my_lo.view(-1))
        if self._ev:
            self._ev[3].record()
        eng.refresh_wt()
        self.steps += 1
        if self._ev:
            self._pending = True

    def collect_time(self):
        """Add the last step's collective time (ms) to collective_ms; synchronises on the recorded events."""
        if self._ev and self._pending:
            self._ev[3].synchronize()
            self.collective_ms += self._ev[0].elapsed_time(self._ev[1]) + self._ev[2].elapsed_time(self._ev[3])
            self._pending = None

    def gather_master(self):
        """Full fp32 W on every rank (each rank contributes the rows it owns)."""
        torch, dist, eng = self.torch, self.dist, self.eng
        c, Hp = eng.chunk_rows, eng.Hp
        mine = torch.zeros((c, Hp), dtype=torch.float32, device=eng.device)
        if self.f1 > self.f0:
            mine[:self.f1 - self.f0] = eng.W[self.f0:self.f1]
        full = torch.empty((eng.rows_alloc, Hp), dtype=torch.float32, device=eng.device)
        dist.all_gather_into_tensor(full.view(-1), mine.view(-1))
        eng.W.copy_(full[:eng.Fp])


def shard_bounds(start, stop, world, r):
    """Contiguous shard [lo, hi) of the global mini-batch [start, stop) owned by rank r."""
    per = -(-(stop - start) // world)
    lo = min(stop, start + r * per)
    return lo, min(stop, lo + per)
