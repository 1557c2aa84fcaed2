"""Data-parallel glue: one process per GPU, torch.distributed (backend "nccl" == RCCL on ROCm, over xGMI).

The reference is single-process.  A data-parallel DAE step exchanges the W gradient, and xGMI is point-to-point (no switch), so
the exchange is a reduce-scatter by row chunks + a sharded optimizer + an all-gather of the low-precision shadow
(``ShardedExchange``) rather than one ring all-reduce of the flat gradient; the bias gradients (10.6 K floats) are all-reduced.
Mining is per rank ("local", SURVEY 8e mode ii) or over the all-gathered global batch (``GlobalMiner``, mode i)."""
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
    quiet_first_collective()


def quiet_first_collective():
    """RCCL creates its communicator at the first collective and prints a version banner to STDOUT while doing so (C stdio,
    flushed at exit -- i.e. after anything Python printed).  Callers like bench.py promise ONE JSON line on stdout, so the first
    collective is issued here with file descriptor 1 pointed at stderr, and the C buffers are flushed before it is restored."""
    import ctypes
    import sys
    import torch
    import torch.distributed as dist
    if not is_initialized():
        return
    sys.stdout.flush()
    libc = ctypes.CDLL(None)
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        t = torch.zeros(1, device=dev)
        dist.all_reduce(t)
        if dev == "cuda":
            torch.cuda.synchronize()
        libc.fflush(None)
    finally:
        os.dup2(saved, 1)
        os.close(saved)


class _stdout_to_stderr:
    """File descriptor 1 pointed at stderr for the duration (C stdio of loaded libraries included; flushed before it is restored)."""

    def __enter__(self):
        import sys
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        import ctypes
        ctypes.CDLL(None).fflush(None)
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


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


def allreduce_sum_float(x):
    import torch
    import torch.distributed as dist
    if not (is_initialized() and dist.get_world_size() > 1):
        return float(x)
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([float(x)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
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
        updated by the optimizer there   -> sharded master weights / optimizer slots; the low-precision rows are written
                                            straight into the all-gather send buffer (dae_plan_apply_rows_packed)
        all-gathered as the bf16 shadow  -> every rank receives all chunks                             ((R-1)/R * 10.4 MB)
        unpacked                         -> W_lo, Wt_lo and the biases in one kernel (dae_plan_dp_unpack)

    TWO collectives per step: the bias gradients (10.6 K floats per rank) do not get an all-reduce of their own -- every rank
    appends its LOCAL bias gradients to its all-gather chunk and all ranks sum the gathered pieces in rank order (identical
    result everywhere).  Only the owner of a row block holds its current fp32 master; ``gather_master`` / ``gather_slots`` rebuild
    the full state where it is needed (get_params / checkpoint).  ``packed=False`` keeps the former three-collective form and
    ``overlap=True`` issues the reduce-scatter on a side stream right behind the dW GEMM, beside the step's tail kernel -- measured
    SLOWER with torch.distributed's RCCL process group (every collective already hops to the group's own stream and back; a second
    hop costs more than the 10 us of tail it hides: profiles/r03_dp_step_breakdown.txt), so it is off by default.

    ``collective_ms`` accumulates the time of the collectives (events on the streams they run on, CUDA only)."""

    def __init__(self, eng, grad_dtype="fp32", packed=True, overlap=False):
        import torch
        import torch.distributed as dist
        assert is_initialized(), "torch.distributed is not initialised"
        self.eng, self.torch, self.dist = eng, torch, dist
        self.world, self.rank = dist.get_world_size(), dist.get_rank()
        assert eng.dp_world == self.world, "create the Engine with dp_world = world size (%d != %d)" % (eng.dp_world, self.world)
        assert grad_dtype in ("fp32", "bf16")
        # split-bf16 mode (precision='bf16x3'): the 1e-4 curve needs fp32 gradients in the exchange and fp32-accurate weights in the
        # encode, so this mode reduce-scatters fp32, updates its rows of the fp32 master and all-gathers the MASTER rows (the same
        # bytes as the hi + lo bf16 images would be); every rank then rebuilds its four low-precision images locally
        self.x3 = bool(getattr(eng, "x3", False))
        if self.x3:
            grad_dtype, packed, overlap = "fp32", False, False
        self.grad_dtype = grad_dtype
        self.packed = bool(packed)
        # only the bf16 shadow W_lo is all-gathered: the fp32 masters of the rows another rank owns are stale here, so the sparse
        # encode must read W_lo (single-GPU bf16 steps read the fp32 master, option encode_w32)
        if not self.x3:
            eng.set_option("encode_w32", 0)
        c, Hp = eng.chunk_rows, eng.Hp
        self.f0 = min(eng.Fp, self.rank * c)
        self.f1 = min(eng.Fp, (self.rank + 1) * c)
        self.n_w = eng.rows_alloc * Hp
        lo16 = eng.td if eng.td != torch.float32 else torch.bfloat16      # 16-bit storage format of the loaded library build (bf16 / fp16)
        gdt = torch.float32 if grad_dtype == "fp32" else lo16
        self.rs_out = torch.zeros(c * Hp, dtype=gdt, device=eng.device)
        self.rs_f32 = self.rs_out if grad_dtype == "fp32" else torch.zeros(c * Hp, dtype=torch.float32, device=eng.device)
        self.my_lo = torch.zeros((c, Hp), dtype=eng.td, device=eng.device)
        # packed chunk: [c x Hp low-precision rows | pad to 16 B | Hp + Fp fp32 bias gradients | pad to 16 B]
        es = 2 if eng.td != torch.float32 else 4
        self.bias_off = -(-(c * Hp * es) // 16) * 16
        self.chunk_stride = self.bias_off + -(-((Hp + eng.Fp) * 4) // 16) * 16
        self.send = torch.zeros(self.chunk_stride, dtype=torch.uint8, device=eng.device)
        self.recv = torch.zeros(self.world * self.chunk_stride, dtype=torch.uint8, device=eng.device)
        self.collective_ms = 0.0
        self.steps = 0
        self._ev = None
        self._side = None
        if eng.device.type == "cuda":
            self._ev = tuple(torch.cuda.Event(enable_timing=True) for _ in range(4))
            if overlap and self.packed:
                self._side = torch.cuda.Stream(device=eng.device)
                self._ev_dw = torch.cuda.Event(); self._ev_rs = torch.cuda.Event()
        self._pending = None
        self._my_w = None

    def _reduce_scatter(self, gw):
        torch, dist, eng = self.torch, self.dist, self.eng
        if self.grad_dtype == "fp32":
            dist.reduce_scatter_tensor(self.rs_out, gw, op=dist.ReduceOp.SUM)
        else:
            # bf16 exchange: the dW kernel's epilogue wrote the gradient as bf16 (Engine(grad_lo=True)); otherwise cast here
            src = eng.grad_lo.view(-1) if getattr(eng, "grad_lo", None) is not None else gw.to(self.rs_out.dtype)
            dist.reduce_scatter_tensor(self.rs_out, src, op=dist.ReduceOp.SUM)
            self.rs_f32.copy_(self.rs_out)

    def step(self, grad_scale, grad_ready_after_dw=False):
        """Call after eng.train_step(phase=1): exchange + sharded update + shadow rebuild.  grad_scale multiplies the rank-SUMMED
        gradient (1/world for equal shards).  grad_ready_after_dw=True: the W gradient is exactly what the step's dW kernel wrote
        (nothing was enqueued behind train_step that touches it), so the reduce-scatter may start behind that kernel, beside the
        step's tail; otherwise it starts behind everything enqueued so far."""
        torch, dist, eng = self.torch, self.dist, self.eng
        Hp, c = eng.Hp, eng.chunk_rows
        gw = eng.grad[:self.n_w]
        if not self.packed:
            return self._step_three_collectives(grad_scale)
        if self._side is not None:
            # reduce-scatter on the side stream: it waits for the W gradient only (the last mark, else everything enqueued so far)
            cur = torch.cuda.current_stream()
            if not (grad_ready_after_dw and eng.stream_wait_dw(self._side)):
                self._ev_dw.record(cur)
                self._side.wait_event(self._ev_dw)
            with torch.cuda.stream(self._side):
                self._ev[0].record()
                self._reduce_scatter(gw)
                self._ev[1].record()
                self._ev_rs.record()
            cur.wait_event(self._ev_rs)
        else:
            if self._ev:
                self._ev[0].record()
            self._reduce_scatter(gw)
            if self._ev:
                self._ev[1].record()
        eng.adam_t += 1 if eng.opt == "adam" else 0
        eng.apply_rows_packed(self.rs_f32, self.f0, self.f1, self.send, self.bias_off, grad_scale=grad_scale)
        if self._ev:
            self._ev[2].record()
        dist.all_gather_into_tensor(self.recv, self.send)
        if self._ev:
            self._ev[3].record()
        eng.dp_unpack(self.recv, self.world, self.chunk_stride, self.bias_off, grad_scale=grad_scale)
        self.steps += 1
        if self._ev:
            self._pending = True

    def _step_three_collectives(self, grad_scale):
        torch, dist, eng = self.torch, self.dist, self.eng
        Hp, c = eng.Hp, eng.chunk_rows
        gw = eng.grad[:self.n_w]
        bias = eng.grad[eng.Fp * Hp:eng.Fp * Hp + Hp + eng.Fp]
        if self._ev:
            self._ev[0].record()
        self._reduce_scatter(gw)
        dist.all_reduce(bias, op=dist.ReduceOp.SUM)
        if self._ev:
            self._ev[1].record()
        eng.adam_t += 1 if eng.opt == "adam" else 0
        eng.apply_rows(self.rs_f32, self.f0, self.f1, grad_scale=grad_scale, update_bias=True)
        if self.x3:
            if self._my_w is None:
                self._my_w = torch.zeros((c, Hp), dtype=torch.float32, device=eng.device)
            self._my_w.copy_(eng.W_full[self.rank * c:(self.rank + 1) * c])
            if self._ev:
                self._ev[2].record()
            dist.all_gather_into_tensor(eng.W_full.view(-1), self._my_w.view(-1))
            if self._ev:
                self._ev[3].record()
            eng.sync_shadows()
        else:
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
        if self.x3:
            return                      # every step all-gathers the master rows: W is already current everywhere
        c, Hp = eng.chunk_rows, eng.Hp
        mine = torch.zeros((c, Hp), dtype=torch.float32, device=eng.device)
        if self.f1 > self.f0:
            mine[:self.f1 - self.f0] = eng.W[self.f0:self.f1]
        full = torch.empty((eng.rows_alloc, Hp), dtype=torch.float32, device=eng.device)
        dist.all_gather_into_tensor(full.view(-1), mine.view(-1))
        eng.W.copy_(full[:eng.Fp])

    def gather_slots(self):
        """Full optimizer slots on every rank.  The sharded optimizer updates the W rows of s1 / s2 (momentum, Adagrad, Adam
        state) only where the row chunk is owned, so a checkpoint written from one rank's buffers would pair current weights with
        initial slots for (world-1)/world of W.  The bias slots are updated identically on every rank."""
        torch, dist, eng = self.torch, self.dist, self.eng
        c, Hp, n = eng.chunk_rows, eng.Hp, eng.Fp * eng.Hp
        for t in (eng.s1, eng.s2):
            if t is None:
                continue
            part = t[:n].view(eng.Fp, Hp)
            mine = torch.zeros((c, Hp), dtype=torch.float32, device=eng.device)
            if self.f1 > self.f0:
                mine[:self.f1 - self.f0] = part[self.f0:self.f1]
            full = torch.empty((eng.rows_alloc, Hp), dtype=torch.float32, device=eng.device)
            dist.all_gather_into_tensor(full.view(-1), mine.view(-1))
            part.copy_(full[:eng.Fp])


class AllReduceExchange:
    """The exchange of the split-bf16 mode (precision='bf16x3', what precision='auto' resolves to) under data parallel: ONE collective.

        all-reduce of the flat fp32 gradient [dW | dbh | dbv]    (2 (R-1)/R * 20.7 MB per rank at 10000 x 500 -- exactly the bytes of
                                                                  ShardedExchange's fp32 reduce-scatter + fp32 master all-gather)
        dae_plan_apply on every rank                               (optimizer on the whole W and the biases + all four low-precision images,
                                                                  one opt_w_kernel pass: what the sharded form spends on dae_plan_sync_shadows)

    The split mode cannot shrink the exchange below fp32 gradients in and fp32-accurate weights out, so sharding the optimizer buys
    no bytes here -- it only adds two collective launches (reduce-scatter + all-gather instead of one all-reduce), the bias all-reduce, a copy of
    the master rows and torch's stream hops around each of them: without communication the sharded step form costs 93.6 us on top of the phase-1
    step, this one ~35 (tools/dp_step_breakdown.py --precision bf16x3).  Every rank holds the full master weights and optimizer slots at all
    times: gather_master / gather_slots are no-ops.  Same interface as ShardedExchange.

    buckets > 1 (an option; default 1): the flat gradient is cut into `buckets` row bands of W (whole 64-row blocks; the
    bias gradients ride at the end of the last band).  All bands are handed to the collective library at once (async_op: RCCL runs them back to back
    on its own stream) and the step's stream applies band k (dae_plan_apply_band: master rows, slots and every 16-bit image of those rows) as soon as
    band k has arrived -- the optimizer pass over band k runs while band k + 1 is on the wire, so of the ~20 us optimizer pass only the last band's
    share stays exposed behind the collective.  Element-wise the same sums and the same update as the single-bucket form (bit-identical for two
    ranks; for more ranks the reduction order inside the collective library may differ by bucket size).  Measured without communication (one-rank RCCL
    group, tools/dp_step_breakdown.py, profiles/r05_dp_step_breakdown.txt): every torch.distributed call costs ~25 us of HOST time, so four buckets
    make the exchange 117 us of host work per step against 30 us for one -- the bands only pay when the all-reduce on the wire is longer than that,
    which a one-GPU box cannot show; the default therefore stays ONE bucket until a multi-GPU measurement says otherwise (bench.py --buckets N)."""

    def __init__(self, eng, buckets=None):
        import torch
        import torch.distributed as dist
        assert is_initialized(), "torch.distributed is not initialised"
        self.eng, self.torch, self.dist = eng, torch, dist
        self.world, self.rank = dist.get_world_size(), dist.get_rank()
        assert eng.dp_world == self.world, "create the Engine with dp_world = world size (%d != %d)" % (eng.dp_world, self.world)
        # the flat fp32 gradient is what gets all-reduced: an engine built with a 16-bit exchange image (grad_lo) writes the W gradient THERE in
        # phase 1 / 5 steps and leaves eng.grad's W part stale -- refuse the combination instead of training on a stale gradient
        if getattr(eng, "grad_lo", None) is not None:
            raise ValueError("AllReduceExchange all-reduces the flat fp32 gradient: build the Engine with grad_lo=False (dp_grad_dtype='fp32')")
        self.x3 = bool(getattr(eng, "x3", False))
        self.grad_dtype, self.packed = "fp32", False
        self.flat = eng.grad[:eng.n_flat]                       # [dW (Fp*Hp) | dbh (Hp) | dbv (Fp)]
        self.f0, self.f1 = 0, eng.Fp                            # (interface parity: this rank "owns" every row)
        if buckets is None:
            buckets = 1
        nblk = eng.Fp // 64
        self.buckets = max(1, min(int(buckets), nblk))
        # band k = rows [bounds[k], bounds[k + 1]) of W; its slice of the flat buffer ends at the band's last row -- or at the end of the buffer (biases)
        self.bounds = [64 * ((nblk * k) // self.buckets) for k in range(self.buckets)] + [eng.Fp]
        self.slices = [self.flat[self.bounds[k] * eng.Hp:(self.bounds[k + 1] * eng.Hp if k + 1 < self.buckets else eng.n_flat)] for k in range(self.buckets)]
        self.collective_ms = 0.0
        self.steps = 0
        self._ev = tuple(torch.cuda.Event(enable_timing=True) for _ in range(2)) if eng.device.type == "cuda" else None
        self._pending = None

    def step(self, grad_scale, grad_ready_after_dw=False):
        """Call after eng.train_step(phase=1): all-reduce + full optimizer step.  grad_scale multiplies the rank-SUMMED gradient."""
        if self._ev:
            self._ev[0].record()
        if self.buckets == 1:
            self.dist.all_reduce(self.flat, op=self.dist.ReduceOp.SUM)
            if self._ev:
                self._ev[1].record()
                self._pending = True
            self.eng.apply(grad_scale=grad_scale)               # (counts the Adam step itself)
        else:
            works = [self.dist.all_reduce(sl, op=self.dist.ReduceOp.SUM, async_op=True) for sl in self.slices]
            self.eng.begin_apply()                              # one optimizer step (Adam's t), applied band by band
            for k, w in enumerate(works):
                w.wait()                                        # the step's stream waits for band k only
                if k == self.buckets - 1 and self._ev:
                    self._ev[1].record()
                    self._pending = True
                self.eng.apply_band(self.bounds[k], self.bounds[k + 1], grad_scale=grad_scale)
        self.steps += 1

    def collect_time(self):
        if self._ev and self._pending:
            self._ev[1].synchronize()
            self.collective_ms += self._ev[0].elapsed_time(self._ev[1])
            self._pending = None

    def gather_master(self):
        return

    def gather_slots(self):
        return


class Comm:
    """One RCCL communicator of the library's own (include/dae_hip.h: dae_comm_*): the collectives are issued by the C ABI on the step's stream
    instead of through torch.distributed's process group (its own stream: two cross-stream hops and ~25 us of host time per call,
    profiles/r05_dp_step_breakdown.txt).  The 128-byte id travels over whatever channel the host has -- here torch.distributed's broadcast (any
    backend) when a process group exists; a one-rank communicator needs none.  `ranks_seen` = an all-reduced token: what the communicator really spans."""

    def __init__(self, lib, device, rank=None, world=None):
        import ctypes as C
        import torch
        from . import _lib as L
        self.lib, self.L, self.C, self.torch = lib, L, C, torch
        self.rank = rank if rank is not None else (globals()["rank"]())
        self.world = world if world is not None else world_size()
        self.device = torch.device(device)
        idb = (C.c_uint8 * L.COMM_ID_BYTES)()
        with torch.cuda.device(self.device), _stdout_to_stderr():      # RCCL prints its version banner to STDOUT when a communicator comes up
            err = None
            if self.rank == 0:
                try:
                    L.check(lib.dae_comm_unique_id(idb), "dae_comm_unique_id", lib)
                except RuntimeError as e:      # (RCCL missing ...): the other ranks are waiting in the broadcast below -- send them an all-zero id so
                    err = e                    # that EVERY rank raises and falls back together instead of rank 0 alone leaving the rendezvous
                    idb = (C.c_uint8 * L.COMM_ID_BYTES)()
            if self.world > 1:
                raw = broadcast_array(np.frombuffer(bytes(idb), np.uint8).copy(), src=0)
                idb = (C.c_uint8 * L.COMM_ID_BYTES)(*[int(v) for v in raw])
            if not any(bytes(idb)):
                raise RuntimeError("dae_comm: rank 0 could not create a communicator id (%s)" % (err if err is not None else "all-zero id received"))
            h = C.c_void_p()
            L.check(lib.dae_comm_init(idb, self.rank, self.world, C.byref(h)), "dae_comm_init", lib)
            self.handle = h
            tok = torch.ones(1, dtype=torch.float32, device=self.device)
            self.allreduce_(tok)
            torch.cuda.synchronize(self.device)
        self.ranks_seen = int(tok.item())
        if self.ranks_seen != self.world:
            raise RuntimeError("dae_comm spans %d ranks, expected %d" % (self.ranks_seen, self.world))
        self.library = lib.dae_comm_library().decode()

    def allreduce_(self, t, op="sum"):
        """In-place all-reduce of a float32 device tensor on the current stream."""
        L = self.L
        assert t.dtype == self.torch.float32 and t.is_contiguous()
        L.check(self.lib.dae_comm_allreduce_f32(self.handle, L.ptr(t), t.numel(), 0 if op == "sum" else 1, L.current_stream()), "dae_comm_allreduce_f32", self.lib)
        return t

    def close(self):
        if getattr(self, "handle", None):
            self.lib.dae_comm_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:        # noqa: BLE001
            pass


class NativeAllReduceExchange:
    """AllReduceExchange with the collective in the C ABI (dae_dp_exchange, csrc/dae_api.hip + dae_comm.hip): ONE library call per step enqueues the
    all-reduce of the flat fp32 gradient [dW | dbh | dbv] and the optimizer behind it.  buckets = 1: both on the step's stream, no hop.  buckets > 1: the
    W-only row bands are reduced on the communicator's wire stream from the moment the dW GEMM has finished (beside the step's tail kernel), the last
    band (bias gradients) behind the tail, and the step's stream applies band k while band k + 1 is on the wire -- event waits cost microseconds here,
    so the bands are affordable (through torch.distributed four buckets cost 117 us of host time per step).  Same interface and the same arithmetic
    as AllReduceExchange (the band boundaries are dae_dp_bands == AllReduceExchange.bounds: tests/test_dp_exchange_gloo.py)."""

    def __init__(self, eng, buckets=None, comm=None):
        import torch
        self.eng, self.torch = eng, torch
        if getattr(eng, "grad_lo", None) is not None:
            raise ValueError("the all-reduce exchange moves the flat fp32 gradient: build the Engine with grad_lo=False (dp_grad_dtype='fp32')")
        self.comm = comm if comm is not None else Comm(eng.lib, eng.device)
        self.world, self.rank = self.comm.world, self.comm.rank
        assert eng.dp_world == self.world, "create the Engine with dp_world = world size (%d != %d)" % (eng.dp_world, self.world)
        self.x3 = bool(getattr(eng, "x3", False))
        self.grad_dtype, self.packed = "fp32", False
        self.f0, self.f1 = 0, eng.Fp
        self.bounds = native_bands(eng.lib, eng.Fp, 1 if buckets is None else buckets)
        self.buckets = len(self.bounds) - 1
        self.collective_ms = 0.0
        self.steps = 0
        self._ev = tuple(torch.cuda.Event(enable_timing=True) for _ in range(2))
        self._pending = None

    def step(self, grad_scale, grad_ready_after_dw=False):
        """Call after eng.train_step(phase=1).  grad_scale multiplies the rank-SUMMED gradient."""
        L, eng = self.comm.L, self.eng
        self._ev[0].record()
        eng.begin_apply()
        L.check(eng.lib.dae_dp_exchange(eng.plan, self.comm.handle, eng.adam_t, float(grad_scale), self.buckets, L.current_stream()), "dae_dp_exchange", eng.lib)
        self._ev[1].record()          # (brackets all-reduce + optimizer: the whole second half of the step)
        self._pending = True
        self.steps += 1

    def collect_time(self):
        if self._pending:
            self._ev[1].synchronize()
            self.collective_ms += self._ev[0].elapsed_time(self._ev[1])
            self._pending = None

    def gather_master(self):
        return

    def gather_slots(self):
        return


def native_bands(lib, Fp, buckets):
    """Row-band boundaries of the bucketed exchange as the C ABI computes them (dae_dp_bands; host arithmetic, no GPU)."""
    import ctypes as C
    b = (C.c_int32 * 16)()
    n = lib.dae_dp_bands(int(Fp), int(buckets), b)
    return [int(b[k]) for k in range(n + 1)]


def native_collective_ok(eng):
    """Can the library's own RCCL communicator carry this process group?  One rank per device over the nccl backend (or no group at all: one rank)."""
    import torch.distributed as dist
    if eng.device.type != "cuda":
        return False
    return (not is_initialized()) or dist.get_world_size() == 1 or dist.get_backend() == "nccl"


def make_exchange(eng, grad_dtype="fp32", kind="auto", buckets=None, impl="auto"):
    """The data-parallel exchange for `eng`: kind 'auto' = the all-reduce form in the split modes (nothing to gain from sharding there),
    ShardedExchange otherwise (bf16 gradients / bf16 shadow rows halve its bytes); 'sharded' / 'allreduce' force one form.
    impl: who issues the all-reduce -- 'native' = the C ABI's own RCCL communicator on the step's stream (NativeAllReduceExchange), 'torch' =
    torch.distributed's process group (AllReduceExchange: any backend, the form the gloo tests drive), 'auto' = native whenever the group is one rank
    per device over RCCL; should the native communicator fail to come up there, the torch form takes over with a message on stderr (both are RCCL)."""
    assert kind in ("auto", "sharded", "allreduce"), kind
    assert impl in ("auto", "native", "torch"), impl
    if kind == "allreduce" or (kind == "auto" and getattr(eng, "x3", False)):
        if impl == "native" or (impl == "auto" and native_collective_ok(eng)):
            try:
                return NativeAllReduceExchange(eng, buckets=buckets)
            except Exception as e:        # noqa: BLE001
                if impl == "native" or not is_initialized():
                    raise
                import sys
                print("[dae dp] native RCCL communicator unavailable (%s): using torch.distributed's process group" % (repr(e)[:200],), file=sys.stderr, flush=True)
        return AllReduceExchange(eng, buckets=buckets)
    return ShardedExchange(eng, grad_dtype=grad_dtype)


class GlobalMiner:
    """Global-batch triplet mining under data parallel (SURVEY 8e mode i): the reference objective at the GLOBAL batch size.

    Per global mini-batch every rank
      1. encodes ITS rows (Engine.train_step phase 4) and all-gathers the embeddings h (equal blocks of `per` rows; the
         shards are contiguous, so the gathered rows [0, B_glob) are the global batch in order);
      2. computes the rows of D = h h^T that belong to its anchors (exact-fp32 MFMA GEMM, dae_gemm_nt) and runs the miner on
         them against ALL columns (dae_triplet_batch_{all,hard}_rows);
      3. normalisers: N_valid and data_weight of batch_all are closed forms of the global label histogram (host integers);
         batch_hard all-reduces its count and its data_weight vector; the loss sum is all-reduced;
      4. dL/dh of its rows = scale * (G_loc h_all + reduce_scatter(G_loc^T h_loc)): G = dL/dD restricted to the rank's anchor
         rows, (G + G^T) h split into the local product and the one exchanged by reduce-scatter (12.8 MB at 8 x 800 x 500);
      5. writes cw / tri_scalars / dh_extra into the plan and resumes the step (phase 5).
    The flat gradients of the ranks then SUM to the gradient of the global-batch cost (grad_scale = 1 in the exchange)."""

    def __init__(self, eng, strategy, alpha, max_global_batch):
        import ctypes as C
        import torch
        import torch.distributed as dist
        from . import _lib as L
        assert strategy in ("batch_all", "batch_hard")
        self.eng, self.torch, self.dist, self.L, self.C = eng, torch, dist, L, C
        self.strategy, self.alpha = strategy, float(alpha)
        self.world, self.rank = dist.get_world_size(), dist.get_rank()
        self.per_max = -(-int(max_global_batch) // self.world)
        self.Mp = L.pad(self.per_max)
        self.Gp = L.pad(self.per_max * self.world)
        # the miners hold one D row of the GLOBAL batch per workgroup in LDS: dae_triplet_batch_all_rows up to 4096 padded columns,
        # dae_triplet_batch_hard_rows up to 16384 -- fail here, not at the first step of the fit
        limit = 4096 if strategy == "batch_all" else 16384
        if self.Gp > limit:
            raise ValueError("dp_mining='global' with %s supports a padded global batch of at most %d rows (got %d = %d ranks x %d); "
                             "use dp_mining='local' or a smaller batch" % (strategy, limit, self.Gp, self.world, self.per_max))
        dev, Hp = eng.device, eng.Hp
        z = lambda *shape, dtype=torch.float32: torch.zeros(shape, dtype=dtype, device=dev)
        self.h_blocks = z(self.world * self.per_max, Hp)
        self.h_all = z(self.Gp, Hp); self.h_allT = z(Hp, self.Gp)
        self.h_loc = z(self.Mp, Hp); self.h_locT = z(Hp, self.Mp)
        self.D = z(self.Mp, self.Gp); self.G = z(self.Mp, self.Gp); self.GT = z(self.Gp, self.Mp)
        self.T1 = z(self.Mp, Hp); self.T2 = z(self.Gp, Hp); self.T2blk = z(self.world * self.per_max, Hp); self.rs = z(self.per_max, Hp)
        self.loss_part = z(self.Mp); self.cnt_part = z(self.Mp, dtype=torch.int32); self.dw = z(self.Gp, dtype=torch.int32)
        self.role_cnt = None
        self.cw = eng.buffer("cw", (eng.Bpm,), torch.float32)
        self.dh_extra = eng.buffer("dh_extra", (eng.Bpm, Hp), torch.float32)
        self.tri = eng.buffer("tri_scalars", (64,), torch.float32)
        self.h_f32 = eng.buffer("h_f32", (eng.Bpm, Hp), torch.float32)

    def _gemm(self, M, N, A, lda, Bt, ldb, K, Cm):
        L = self.L
        L.call("dae_gemm_nt", L.F32, M, N, L.ptr(A), lda, L.ptr(Bt), ldb, K, None, 0, None, 0, 0, L.ptr(Cm), N, 1, 0, L.current_stream())

    def _transpose(self, src, rows, cols, dst):
        L = self.L
        L.call("dae_transpose_shadow", L.ptr(src), rows, cols, L.F32, L.ptr(dst), L.current_stream())

    def mine(self, labels_host, start, stop):
        """labels_host: int ids of the global batch rows [start, stop) (NumPy, identical on every rank).  Returns the global
        (triplet_loss, fraction, num) after writing cw / dh_extra / tri_scalars for this rank's rows."""
        torch, dist, L, eng = self.torch, self.dist, self.L, self.eng
        Bg = stop - start
        per = -(-Bg // self.world)
        lo = min(Bg, self.rank * per); hi = min(Bg, lo + per)
        nA = hi - lo
        Hp, Gp, Mp = eng.Hp, self.Gp, self.Mp
        labels = torch.from_numpy(np.ascontiguousarray(labels_host, dtype=np.int32)).to(eng.device)
        # 1. all-gather the embeddings (equal blocks of `per_max` rows; only the first `per` of a block are used)
        self.h_loc.zero_()
        if nA:
            self.h_loc[:nA] = self.h_f32[:nA]
        dist.all_gather_into_tensor(self.h_blocks.view(-1), self.h_loc[:self.per_max].contiguous().view(-1))
        self.h_all.zero_()
        hb = self.h_blocks.view(self.world, self.per_max, Hp)
        for r in range(self.world):
            r0 = min(Bg, r * per); r1 = min(Bg, r0 + per)
            if r1 > r0:
                self.h_all[r0:r1] = hb[r, :r1 - r0]
        # 2. D rows of my anchors, miner on them
        self.G.zero_(); self.loss_part.zero_(); self.cnt_part.zero_()
        loss_sum = torch.zeros(2, dtype=torch.float64, device=eng.device)
        if nA:
            self._gemm(Mp, Gp, self.h_loc, Hp, self.h_all, Hp, Hp, self.D)
            if self.strategy == "batch_all":
                L.call("dae_triplet_batch_all_rows", L.ptr(self.D), 1, 0, Gp, L.ptr(labels), Bg, Gp, lo, nA, 0, L.ptr(self.loss_part),
                       L.ptr(self.cnt_part), L.ptr(self.G), None, L.current_stream())
            else:
                L.call("dae_triplet_batch_hard_rows", L.ptr(self.D), 1, 0, Gp, L.ptr(labels), Bg, Gp, lo, nA, L.ptr(self.loss_part),
                       L.ptr(self.cnt_part), L.ptr(self.dw), L.ptr(self.G), L.current_stream())
            loss_sum[0] = self.loss_part[:nA].double().sum(); loss_sum[1] = self.cnt_part[:nA].double().sum()
        elif self.strategy == "batch_hard":
            self.dw.zero_()
        dist.all_reduce(loss_sum)
        # 3. normalisers and row weights
        lab = np.asarray(labels_host).astype(np.int64)
        if self.strategy == "batch_all":
            _, inv, cnt = np.unique(lab, return_inverse=True, return_counts=True)
            n = cnt[inv].astype(np.int64)
            S = int((cnt * (cnt - 1)).sum())
            NV = int((cnt * (cnt - 1) * (Bg - cnt)).sum())                       # triplet_loss_utils.py:110-111 as a closed form
            dw = 2 * (n - 1) * (Bg - n) + (S - n * (n - 1))                      # :129
            N = float(NV)
            cw = dw[lo:hi].astype(np.float64) / (3.0 * NV + 1e-16)
            cw_t = torch.from_numpy(cw.astype(np.float32)).to(eng.device)
            num = float(loss_sum[1].item()); frac = num / (NV + 1e-16)
        else:
            dist.all_reduce(self.dw)
            dwf = self.dw[:Bg].float()
            N = float(loss_sum[1].item())
            cw_t = dwf[lo:hi] / (dwf.sum() + 1e-16)
            num = N; frac = N / float(Bg)
        tl = float(loss_sum[0].item()) / (N + 1e-16)
        scale = self.alpha / (N + 1e-16)
        # 4. dL/dh of my rows
        self._transpose(self.h_all, Gp, Hp, self.h_allT)
        self._transpose(self.h_loc, Mp, Hp, self.h_locT)
        self._transpose(self.G, Mp, Gp, self.GT)
        self._gemm(Mp, Hp, self.G, Gp, self.h_allT, Gp, Gp, self.T1)
        self._gemm(Gp, Hp, self.GT, Mp, self.h_locT, Mp, Mp, self.T2)
        self.T2blk.zero_()
        tb = self.T2blk.view(self.world, self.per_max, Hp)
        for r in range(self.world):
            r0 = min(Bg, r * per); r1 = min(Bg, r0 + per)
            if r1 > r0:
                tb[r, :r1 - r0] = self.T2[r0:r1]
        dist.reduce_scatter_tensor(self.rs.view(-1), self.T2blk.view(-1), op=dist.ReduceOp.SUM)
        # 5. hand over to the resumed step
        self.cw.zero_(); self.dh_extra.zero_()
        if nA:
            self.cw[:nA] = cw_t
            self.dh_extra[:nA] = scale * (self.T1[:nA] + self.rs[:nA])
        self.tri[1] = tl; self.tri[2] = frac; self.tri[3] = num
        return tl, frac, num


def shard_bounds(start, stop, world, r):
    """Contiguous shard [lo, hi) of the global mini-batch [start, stop) owned by rank r."""
    per = -(-(stop - start) // world)
    lo = min(stop, start + r * per)
    return lo, min(stop, lo + per)
