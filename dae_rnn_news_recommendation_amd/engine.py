"""Device-side engine: owns the HBM-resident train set, the padded parameters / optimizer slots and the
workspace, and drives libdae_hip's whole-step entry point (`dae_train_step`).

HBM layout: the train set stays resident as CSR (int64 indptr, int32 sorted column
ids, fp32 values or none for binary) or as a dense fp32 matrix; parameters are fp32 masters padded to
multiples of 128 (W [Fp x Hp], bh [Hp], bv [Fp]) with bf16 (or fp32) shadows W_lo and W^T_lo that the
optimizer kernel refreshes; the gradient is one flat fp32 buffer [dW | dbh | dbv] so data parallelism is
a single all-reduce.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib as L


class Engine:
    def __init__(self, n_features, n_components, max_batch, *, dtype="bf16", enc_act="sigmoid", dec_act="sigmoid",
                 loss_func="cross_entropy", opt="gradient_descent", learning_rate=0.1, momentum=0.5, alpha=1.0,
                 triplet="none", pos_triplets_only=False, device=None, encode_splits=0, dh_splits=0, gram_splits=0, dp_world=1,
                 grad_lo=False):
        if not torch.cuda.is_available():
            raise RuntimeError("dae_rnn_news_recommendation_amd.Engine needs a ROCm GPU (MI355X): no CPU fallback exists")
        # precision name -> library build (16-bit storage format), dae_config.dtype and the split mode's lo terms (L.PRECISIONS)
        if isinstance(dtype, str):
            assert dtype in L.PRECISIONS, dtype
            self.fmt, cfg_dtype, terms = L.PRECISIONS[dtype]
        else:
            self.fmt, cfg_dtype, terms = "bf16", dtype, None
        self.precision = dtype
        self.lib = L.load(self.fmt)
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        self.F, self.H, self.Bmax = int(n_features), int(n_components), int(max_batch)
        assert cfg_dtype in (L.BF16, L.F32, L.BF16X3), dtype
        # the split modes store 16-bit images everywhere (self.dtype = the element type of the images); only the plan's config says x3
        self.x3 = cfg_dtype == L.BF16X3
        self.dtype = L.BF16 if self.x3 else cfg_dtype
        self.td = L.torch_lo_dtype(self.fmt) if self.dtype == L.BF16 else torch.float32
        self.opt = opt
        self.cfg = L.dae_config(self.F, self.H, self.Bmax, cfg_dtype, L.ACT[enc_act], L.ACT[dec_act], L.LOSS[loss_func],
                                L.OPT[opt], L.TRIPLET[triplet], int(pos_triplets_only), encode_splits, dh_splits,
                                gram_splits, float(learning_rate), float(momentum), float(alpha))
        plan = C.c_void_p()
        self._chk(self.lib.dae_plan_create(C.byref(self.cfg), C.byref(plan)), "dae_plan_create")
        self.plan = plan
        if terms is not None:
            self._chk(self.lib.dae_plan_set_option(self.plan, b"x3_terms", int(terms)), "dae_plan_set_option")
        self.Fp, self.Hp, self.Bpm = L.pad(self.F), L.pad(self.H), L.pad(self.Bmax)
        dev = self.device
        n_flat = self.Fp * self.Hp + self.Hp + self.Fp
        # data parallel with a sharded optimizer (dp.ShardedExchange): W is cut into `dp_world` equal chunks of whole 64-row
        # blocks; the gradient buffer and W_lo are over-allocated so that every chunk exists.  NOTE: the flat gradient is
        # [dW (Fp*Hp) | dbh | dbv], so "rows" >= Fp of that over-allocated view alias the bias gradients: the exchange may sum them
        # as if they were W rows, and dae_plan_apply_rows clamps its update to rows < Fp
        self.dp_world = int(dp_world)
        self.chunk_rows = -(-self.Fp // (64 * self.dp_world)) * 64
        self.rows_alloc = self.chunk_rows * self.dp_world
        self.n_flat = n_flat
        # (split-bf16 mode under data parallel all-gathers the fp32 MASTER rows: W is over-allocated to whole chunks like W_lo)
        self.W_full = torch.zeros((self.rows_alloc if self.x3 else self.Fp, self.Hp), dtype=torch.float32, device=dev)
        self.W = self.W_full[:self.Fp]
        self.bh = torch.zeros(self.Hp, dtype=torch.float32, device=dev)
        self.bv = torch.zeros(self.Fp, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(max(n_flat, self.rows_alloc * self.Hp), dtype=torch.float32, device=dev)
        # data parallel with a bf16 exchange (dp_grad_dtype='bf16'): the dW kernel writes the W gradient as bf16 straight into
        # this image (dae_buffers.grad_lo) in phase 1 / 5 steps; the bias gradients stay in `grad`
        self.grad_lo = None
        if grad_lo and self.dtype == L.BF16 and not self.x3:       # split-bf16 mode exchanges fp32 gradients
            self.grad_lo = torch.zeros((self.rows_alloc, self.Hp), dtype=self.td, device=dev)
        self.s1 = self.s2 = None
        if opt == "ada_grad":
            self.s1 = torch.full((n_flat,), 0.1, dtype=torch.float32, device=dev)   # TF initial_accumulator_value
        elif opt == "momentum":
            self.s1 = torch.zeros(n_flat, dtype=torch.float32, device=dev)
        elif opt == "adam":
            self.s1 = torch.zeros(n_flat, dtype=torch.float32, device=dev)
            self.s2 = torch.zeros(n_flat, dtype=torch.float32, device=dev)
        self.W_lo_full = torch.zeros((self.rows_alloc, self.Hp), dtype=self.td, device=dev)
        self.W_lo = self.W_lo_full[:self.Fp]
        self.Wt_lo = torch.zeros((self.Hp, self.Fp), dtype=self.td, device=dev)
        ws_bytes = int(self.lib.dae_plan_workspace_bytes(self.plan))
        self.workspace = torch.zeros(ws_bytes, dtype=torch.uint8, device=dev)
        self.csr = None
        self.dense = None
        self.adam_t = 0
        self._bound = False

    def _chk(self, rc, what=""):
        L.check(rc, what, self.lib)

    # ------------------------------------------------------------------ data
    def upload_csr(self, m):
        """Make a scipy CSR matrix resident in HBM (canonical form: sorted, no duplicates)."""
        from scipy import sparse
        m = sparse.csr_matrix(m)
        if not m.has_canonical_format:
            m = m.copy(); m.sum_duplicates(); m.sort_indices()
        assert m.shape[1] == self.F, (m.shape, self.F)
        dev = self.device
        binary = bool(m.nnz == 0 or np.all(m.data == 1))
        self.csr = dict(
            indptr=torch.from_numpy(m.indptr.astype(np.int64)).to(dev),
            indices=torch.from_numpy(m.indices.astype(np.int32)).to(dev),
            values=None if binary else torch.from_numpy(m.data.astype(np.float32)).to(dev),
            n_rows=m.shape[0], nnz=int(m.nnz))
        self._max_row_nnz = int(np.diff(m.indptr).max()) if m.shape[0] else 0
        self._sp = None
        self.dense = None
        self._bind()
        return self.csr

    @staticmethod
    def supports_x3(data, scale=1.0):
        """Can precision='bf16x3' run this train set?  Every input kind: binary CSR is exact in bf16; valued CSR (tf-idf, decay noise's scale
        factor, salt-and-pepper copies) and dense ndarrays get lo images of x, x~ and x~^T as well."""
        return data is not None

    def upload_dense(self, a):
        a = np.ascontiguousarray(a, dtype=np.float32)
        assert a.shape[1] == self.F
        self.dense = torch.from_numpy(a).to(self.device)
        self.csr = None
        self._bind()

    @staticmethod
    def to_device_csr(m, device):
        from scipy import sparse
        m = sparse.csr_matrix(m)
        if not m.has_canonical_format:
            m = m.copy(); m.sum_duplicates(); m.sort_indices()
        return dict(indptr=torch.from_numpy(m.indptr.astype(np.int64)).to(device),
                    indices=torch.from_numpy(m.indices.astype(np.int32)).to(device),
                    values=torch.from_numpy(m.data.astype(np.float32)).to(device), n_rows=m.shape[0], nnz=int(m.nnz))

    def _bind(self):
        b = L.dae_buffers()
        if self.csr is not None:
            b.indptr = self.csr["indptr"].data_ptr(); b.indices = self.csr["indices"].data_ptr()
            b.values = None if self.csr["values"] is None else self.csr["values"].data_ptr()
            b.n_rows = self.csr["n_rows"]; b.nnz = self.csr["nnz"]
        if self.dense is not None:
            b.dense = self.dense.data_ptr(); b.ld_dense = self.dense.stride(0); b.n_rows = self.dense.shape[0]
        b.W = self.W.data_ptr(); b.bh = self.bh.data_ptr(); b.bv = self.bv.data_ptr(); b.grad = self.grad.data_ptr()
        b.opt_s1 = None if self.s1 is None else self.s1.data_ptr()
        b.opt_s2 = None if self.s2 is None else self.s2.data_ptr()
        b.W_lo = self.W_lo.data_ptr(); b.Wt_lo = self.Wt_lo.data_ptr()
        b.workspace = self.workspace.data_ptr(); b.workspace_bytes = self.workspace.numel()
        b.grad_lo = None if self.grad_lo is None else self.grad_lo.data_ptr()
        self._chk(self.lib.dae_plan_bind(self.plan, C.byref(b)), "dae_plan_bind")
        self._bound = True

    # ------------------------------------------------------------------ params
    def set_params(self, W, bh=None, bv=None):
        """Inject (unpadded) parameters; padding stays exactly zero."""
        if not self._bound:
            self._bind()
        self.W.zero_(); self.bh.zero_(); self.bv.zero_()
        self.W[:self.F, :self.H] = torch.as_tensor(np.asarray(W, np.float32)).to(self.device)
        if bh is not None:
            self.bh[:self.H] = torch.as_tensor(np.asarray(bh, np.float32)).to(self.device)
        if bv is not None:
            self.bv[:self.F] = torch.as_tensor(np.asarray(bv, np.float32)).to(self.device)
        self._chk(self.lib.dae_plan_sync_shadows(self.plan, L.current_stream()), "dae_plan_sync_shadows")

    def get_params(self):
        return (self.W[:self.F, :self.H].cpu().numpy(), self.bh[:self.H].cpu().numpy(), self.bv[:self.F].cpu().numpy())

    def grads(self):
        """(dW, dbh, dbv) views of the flat gradient buffer, unpadded copies on the host (the W part from the bf16 exchange
        image when the engine was created with grad_lo=True: phase 1 / 5 steps write it there)."""
        n = self.Fp * self.Hp
        if self.grad_lo is not None:
            dW = self.grad_lo[:self.F, :self.H].float().cpu().numpy()
        else:
            dW = self.grad[:n].view(self.Fp, self.Hp)[:self.F, :self.H].cpu().numpy()
        dbh = self.grad[n:n + self.Hp][:self.H].cpu().numpy()
        dbv = self.grad[n + self.Hp:n + self.Hp + self.Fp][:self.F].cpu().numpy()
        return dW, dbh, dbv

    def optimizer_state(self):
        """Optimizer slots in the flat layout [W | bh | bv].  Under dp.ShardedExchange only the W rows this rank owns are current:
        call ShardedExchange.gather_slots() first (fit() does, before it saves)."""
        return {"s1": self.s1, "s2": self.s2, "adam_t": self.adam_t}

    # ------------------------------------------------------------------ steps
    def train_step(self, row_idx, labels, stats, *, corr_mode=L.CORR_NONE, keep_bits=None, seed=0, rng_stream=0,
                   corr_frac=0.0, scale=1.0, corrupted_csr=None, phase=0, grad_scale=1.0):
        # corrupted_csr may carry "row_idx": the rows of the corrupted CSR to take (batch-local CSR of salt_pepper_batch)
        """Enqueue one mini-batch step.  row_idx / labels: device int32 tensors; stats: device float32[8].
        phase: 0 step + update (grads() stays readable), 1 gradients only (DP), 2 forward only, 3 step + update
        with the optimizer fused into the dW GEMM and no W-gradient image (the training loops use this)."""
        s = L.dae_step()
        s.row_idx = row_idx.data_ptr(); s.labels = None if labels is None else labels.data_ptr()
        s.B = int(row_idx.numel())
        s.corr_mode = corr_mode; s.keep_bits = None if keep_bits is None else keep_bits.data_ptr()
        s.seed = int(seed); s.rng_stream = int(rng_stream); s.corr_frac = float(corr_frac); s.scale = float(scale)
        if corrupted_csr is not None:
            s.c_indptr = corrupted_csr["indptr"].data_ptr(); s.c_indices = corrupted_csr["indices"].data_ptr()
            s.c_values = None if corrupted_csr["values"] is None else corrupted_csr["values"].data_ptr()
            if corrupted_csr.get("row_idx") is not None:
                s.c_row_idx = corrupted_csr["row_idx"].data_ptr()
        s.stats = stats.data_ptr(); s.phase = int(phase)
        if phase in (0, 3) and self.opt == "adam":   # 3 = update without materialising the W gradient
            self.adam_t += 1
        s.adam_t = self.adam_t; s.grad_scale = float(grad_scale)
        self._chk(self.lib.dae_train_step(self.plan, C.byref(s), L.current_stream()), "dae_train_step")

    def salt_pepper_batch(self, row_idx, v, lo, hi, seed, rng_stream):
        """Salt-and-pepper corruption of the batch rows on the device (dae_salt_pepper_batch): returns the batch-local corrupted CSR
        as the ``corrupted_csr`` argument of train_step."""
        assert self.csr is not None, "salt_pepper_batch needs a CSR train set"
        B = int(row_idx.numel())
        if getattr(self, "_sp", None) is None:
            cap = int(self._max_row_nnz + v)
            dev = self.device
            self._sp = dict(cap=cap, span=torch.zeros(2 * self.Bmax, dtype=torch.int64, device=dev),
                            indices=torch.zeros(self.Bmax * cap, dtype=torch.int32, device=dev),
                            values=torch.zeros(self.Bmax * cap, dtype=torch.float32, device=dev),
                            rows=torch.arange(0, 2 * self.Bmax, 2, dtype=torch.int32, device=dev))
        sp = self._sp
        self._chk(self.lib.dae_salt_pepper_batch(L.ptr(self.csr["indptr"]), L.ptr(self.csr["indices"]), L.ptr(self.csr["values"]), L.ptr(row_idx),
                                               B, self.F, int(v), float(lo), float(hi), int(seed), int(rng_stream), L.ptr(sp["span"]),
                                               L.ptr(sp["indices"]), L.ptr(sp["values"]), sp["cap"], L.current_stream()), "dae_salt_pepper_batch")
        return dict(indptr=sp["span"], indices=sp["indices"], values=sp["values"], row_idx=sp["rows"][:B])

    def apply_rows(self, grad_rows, f0, f1, grad_scale=1.0, update_bias=True):
        """Sharded-optimizer step on the rows [f0, f1) this rank owns (dp.ShardedExchange); W_lo rows refreshed, Wt_lo not."""
        self._chk(self.lib.dae_plan_apply_rows(self.plan, self.adam_t, float(grad_scale), L.ptr(grad_rows), int(f0), int(f1),
                                             int(bool(update_bias)), L.current_stream()), "dae_plan_apply_rows")

    def stream_wait_dw(self, stream):
        """Make the torch stream `stream` wait for the W gradient of the last enqueued step (event between the dW GEMM and the step tail).
        Returns False when the event did not exist yet (first call): the caller then waits for the whole step."""
        rc = self.lib.dae_plan_stream_wait_dw(self.plan, C.c_void_p(stream.cuda_stream))
        if rc == L.WAIT_DW_CREATED:
            return False
        self._chk(rc, "dae_plan_stream_wait_dw")        # a genuine HIP error must not read as "event not ready"
        return True

    def apply_rows_packed(self, grad_rows, f0, f1, send, bias_off, grad_scale=1.0):
        """Sharded-optimizer step on the rows [f0, f1); their low-precision image goes straight into the all-gather send buffer
        `send` (uint8 tensor) and this rank's bias gradients are copied to send[bias_off:] (dp.ShardedExchange, packed form)."""
        self._chk(self.lib.dae_plan_apply_rows_packed(self.plan, self.adam_t, float(grad_scale), L.ptr(grad_rows), int(f0), int(f1),
                                                    L.ptr(send), int(bias_off), L.current_stream()), "dae_plan_apply_rows_packed")

    def dp_unpack(self, recv, world, chunk_stride, bias_off, grad_scale=1.0):
        """After the all-gather of the packed chunks: W_lo, Wt_lo and the biases (rank-ordered sum of the gathered bias gradients)."""
        self._chk(self.lib.dae_plan_dp_unpack(self.plan, L.ptr(recv), int(world), int(self.chunk_rows), int(chunk_stride), int(bias_off),
                                            self.adam_t, float(grad_scale), L.current_stream()), "dae_plan_dp_unpack")

    def sync_shadows(self):
        """Rebuild every low-precision image of W (W_lo, Wt_lo and, in split-bf16 mode, their lo parts) from the fp32 master."""
        self._chk(self.lib.dae_plan_sync_shadows(self.plan, L.current_stream()), "dae_plan_sync_shadows")

    def zero_grads(self):
        """An empty shard contributes nothing to the exchange: clear the flat gradient AND its bf16 exchange image."""
        self.grad.zero_()
        if self.grad_lo is not None:
            self.grad_lo.zero_()

    def refresh_wt(self):
        self._chk(self.lib.dae_plan_refresh_wt(self.plan, L.current_stream()), "dae_plan_refresh_wt")

    def apply(self, grad_scale=1.0):
        """Optimizer step on the (all-reduced) flat gradient -- the second half of a DP step."""
        if self.opt == "adam":
            self.adam_t += 1
        self._chk(self.lib.dae_plan_apply(self.plan, self.adam_t, float(grad_scale), L.current_stream()), "dae_plan_apply")

    def begin_apply(self):
        """Count one optimizer step (Adam's t) for a sequence of apply_band calls that together cover W."""
        if self.opt == "adam":
            self.adam_t += 1

    def apply_band(self, f0, f1, grad_scale=1.0):
        """Optimizer step on the rows [f0, f1) of W from the (all-reduced) flat gradient; the band ending at Fp also updates the biases."""
        self._chk(self.lib.dae_plan_apply_band(self.plan, self.adam_t, float(grad_scale), int(f0), int(f1), L.current_stream()), "dae_plan_apply_band")

    def encode_rows(self, row_idx, out, *, scale=1.0, csr=None, dense=None):
        """out[B x H] (device fp32) = encode(scale * rows) -- transform() (autoencoder.py:479-505)."""
        csr = self.csr if (csr is None and dense is None) else csr
        dense = self.dense if (csr is None and dense is None) else dense
        self._chk(self.lib.dae_encode_rows(
            self.plan, L.ptr(row_idx), int(row_idx.numel()), float(scale),
            None if csr is None else L.ptr(csr["indptr"]), None if csr is None else L.ptr(csr["indices"]),
            None if csr is None else L.ptr(csr["values"]),
            None if dense is None else L.ptr(dense), 0 if dense is None else dense.stride(0),
            L.ptr(out), out.stride(0), L.current_stream()), "dae_encode_rows")

    def buffer(self, name, shape, dtype):
        """Debug/test view of a workspace buffer as a torch tensor (no copy)."""
        p = self.lib.dae_plan_buffer(self.plan, name.encode())
        if not p:
            raise KeyError(name)
        off = p - self.workspace.data_ptr()
        n = int(np.prod(shape)) * torch.tensor([], dtype=dtype).element_size()
        return self.workspace[off:off + n].view(dtype).view(*shape)

    def set_option(self, name, value):
        """Code-path choice of the plan (A/B measurements, equivalence tests): see dae_plan_set_option in include/dae_hip.h."""
        self._chk(self.lib.dae_plan_set_option(self.plan, name.encode(), int(value)), "dae_plan_set_option")
        # options that change the split-K plan (x3_dec_wlo, x3_dh_hlo, gram_fp32) also change the workspace size; they are refused once bound
        ws_bytes = int(self.lib.dae_plan_workspace_bytes(self.plan))
        if ws_bytes > self.workspace.numel():
            self.workspace = torch.zeros(ws_bytes, dtype=torch.uint8, device=self.device)

    def profile(self, enable, queued=False, stamps=False):
        """Per-launch HIP-event timing of train_step (dae_plan_profile in include/dae_hip.h).  Default: the host waits behind every launch.
        queued=True: one event pair per launch, read when the pool fills / by profile_read (launches and steps run back to back).
        stamps=True: queued pairs stamped by the dispatch itself (hipExtLaunchKernelGGL) -- the kernel's own duration, as rocprofv3 reports it."""
        self._chk(self.lib.dae_plan_profile(self.plan, (3 if stamps else 2 if queued else 1) if enable else 0), "dae_plan_profile")

    def profile_read(self):
        """{kernel slot name: (total ms, launches)} accumulated since profile(True)."""
        n = int(self.lib.dae_plan_profile_slots())
        ms = (C.c_double * n)(); cnt = (C.c_int32 * n)()
        self._chk(self.lib.dae_plan_profile_read(self.plan, n, ms, cnt), "dae_plan_profile_read")
        return {self.lib.dae_plan_profile_name(i).decode(): (float(ms[i]), int(cnt[i])) for i in range(n)}

    def info(self):
        out = (C.c_int32 * 8)()
        self._chk(self.lib.dae_plan_info(self.plan, out), "dae_plan_info")
        w = int(out[7]) & 0xffffffff
        return dict(Fp=out[0], Hp=out[1], Bpm=out[2], encode_splits=out[3], dh_splits=out[4], gram_splits=out[5], es=out[6],
                    split=bool(w & 1), x3_terms=(w >> 1) & L.X3T_ALL, op_scale=float(2 ** ((w >> 16) & 0xff)), storage=self.fmt)

    def __del__(self):
        try:
            if getattr(self, "plan", None):
                self.lib.dae_plan_destroy(self.plan)
                self.plan = None
        except Exception:
            pass
