"""dp.ShardedExchange itself on CPU (gloo, world 2 and 3): reduce-scatter by row chunks -> row-sharded optimizer -> (packed)
all-gather -> unpack, with a torch-CPU stand-in for the engine calls the exchange makes (dae_plan_apply_rows[_packed],
dae_plan_dp_unpack, dae_plan_refresh_wt, dae_plan_sync_shadows -- restated from include/dae_hip.h for plain SGD).  What is under test
is the choreography the GPU ranks run around `dae_train_step(phase=1)`: buffer sizes and offsets, ragged and EMPTY row chunks, the bias
gradients riding on the all-gather, the bf16 gradient image, the split-bf16 mode's fp32 master exchange, gather_master.  The GPU tests
(tests/test_hip_dp.py) check the same exchange with the real kernels."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LR = 0.1


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


class FakeEngine:
    """The attributes and calls dp.ShardedExchange uses, on CPU tensors (plain SGD, lr = LR)."""

    def __init__(self, Fp, Hp, world, td, x3=False, grad_lo=False):
        self.device = torch.device("cpu")
        self.Fp, self.Hp, self.dp_world, self.td, self.x3 = Fp, Hp, world, td, x3
        self.opt, self.adam_t = "gradient_descent", 0
        self.chunk_rows = -(-Fp // (64 * world)) * 64
        self.rows_alloc = self.chunk_rows * world
        n_flat = Fp * Hp + Hp + Fp
        self.grad = torch.zeros(max(n_flat, self.rows_alloc * Hp))
        self.grad_lo = torch.zeros((self.rows_alloc, Hp), dtype=torch.bfloat16) if grad_lo else None
        self.W_full = torch.zeros((self.rows_alloc if x3 else Fp, Hp)); self.W = self.W_full[:Fp]
        self.W_lo_full = torch.zeros((self.rows_alloc, Hp), dtype=td); self.W_lo = self.W_lo_full[:Fp]
        self.Wt_lo = torch.zeros((Hp, Fp), dtype=td)
        self.bh = torch.zeros(Hp); self.bv = torch.zeros(Fp)
        self.options, self.shadow_syncs = {}, 0
        self.n_flat = n_flat

    def set_option(self, name, value):
        self.options[name] = value

    def stream_wait_dw(self, stream):
        return False

    def _bias_grads(self):
        n = self.Fp * self.Hp
        return self.grad[n:n + self.Hp + self.Fp]

    def apply_rows(self, grad_rows, f0, f1, grad_scale=1.0, update_bias=True):           # dae_plan_apply_rows
        assert 0 <= f0 <= f1 <= self.Fp and f0 % 64 == 0 and f1 % 64 == 0
        g = grad_rows.view(-1, self.Hp)[:f1 - f0]
        self.W[f0:f1] -= LR * grad_scale * g
        self.W_lo[f0:f1] = self.W[f0:f1].to(self.td)
        if update_bias:
            b = self._bias_grads()
            self.bh -= LR * grad_scale * b[:self.Hp]; self.bv -= LR * grad_scale * b[self.Hp:]

    def apply_rows_packed(self, grad_rows, f0, f1, send, bias_off, grad_scale=1.0):     # dae_plan_apply_rows_packed
        assert 0 <= f0 <= f1 <= self.Fp and f0 % 64 == 0 and f1 % 64 == 0
        g = grad_rows.view(-1, self.Hp)[:f1 - f0]
        self.W[f0:f1] -= LR * grad_scale * g
        es = 2 if self.td == torch.bfloat16 else 4
        rows = send[:self.chunk_rows * self.Hp * es].view(self.td).view(self.chunk_rows, self.Hp)
        rows[:f1 - f0] = self.W[f0:f1].to(self.td)
        send[bias_off:bias_off + (self.Hp + self.Fp) * 4].view(torch.float32).copy_(self._bias_grads())

    def dp_unpack(self, recv, world, chunk_stride, bias_off, grad_scale=1.0):            # dae_plan_dp_unpack
        es = 2 if self.td == torch.bfloat16 else 4
        c = self.chunk_rows
        bsum = torch.zeros(self.Hp + self.Fp)
        for r in range(world):
            chunk = recv[r * chunk_stride:(r + 1) * chunk_stride]
            rows = chunk[:c * self.Hp * es].view(self.td).view(c, self.Hp)
            lo, hi = min(self.Fp, r * c), min(self.Fp, (r + 1) * c)
            self.W_lo[lo:hi] = rows[:hi - lo]
            bsum += chunk[bias_off:bias_off + (self.Hp + self.Fp) * 4].view(torch.float32)      # rank order: identical everywhere
        self.Wt_lo.copy_(self.W_lo.T)
        self._bias_grads().copy_(bsum)
        self.bh -= LR * grad_scale * bsum[:self.Hp]; self.bv -= LR * grad_scale * bsum[self.Hp:]

    def refresh_wt(self):                                                                 # dae_plan_refresh_wt
        self.Wt_lo.copy_(self.W_lo.T)

    def apply(self, grad_scale=1.0):                                                      # dae_plan_apply: whole W + biases + shadows
        n = self.Fp * self.Hp
        self.W -= LR * grad_scale * self.grad[:n].view(self.Fp, self.Hp)
        b = self._bias_grads()
        self.bh -= LR * grad_scale * b[:self.Hp]; self.bv -= LR * grad_scale * b[self.Hp:]
        self.W_lo.copy_(self.W.to(self.td)); self.Wt_lo.copy_(self.W_lo.T)
        self.n_flat_seen = n + self.Hp + self.Fp

    def begin_apply(self):                                                                # one optimizer step for a sequence of apply_band calls
        self.bands_seen = []

    def apply_band(self, f0, f1, grad_scale=1.0):                                         # dae_plan_apply_band: rows [f0, f1) of W (+ biases with the last band)
        assert 0 <= f0 < f1 <= self.Fp and f0 % 64 == 0 and f1 % 64 == 0
        n = self.Fp * self.Hp
        self.W[f0:f1] -= LR * grad_scale * self.grad[:n].view(self.Fp, self.Hp)[f0:f1]
        self.W_lo[f0:f1] = self.W[f0:f1].to(self.td); self.Wt_lo[:, f0:f1] = self.W_lo[f0:f1].T
        if f1 == self.Fp:
            b = self._bias_grads()
            self.bh -= LR * grad_scale * b[:self.Hp]; self.bv -= LR * grad_scale * b[self.Hp:]
        self.bands_seen.append((f0, f1))

    def sync_shadows(self):                                                               # dae_plan_sync_shadows
        self.W_lo.copy_(self.W.to(self.td)); self.Wt_lo.copy_(self.W_lo.T)
        self.shadow_syncs += 1


def _local_grads(Fp, Hp, rank, step):
    rng = np.random.default_rng(1000 * step + rank)
    return (torch.from_numpy(rng.standard_normal((Fp, Hp)).astype(np.float32)),
            torch.from_numpy(rng.standard_normal(Hp + Fp).astype(np.float32)))


def _worker(rank, world, port, mode, Fp, Hp, out):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LOCAL_RANK=str(rank))
    from dae_rnn_news_recommendation_amd import dp
    dp.init_from_env("gloo")
    td = torch.float32 if mode == "fp32_shadow" else torch.bfloat16
    eng = FakeEngine(Fp, Hp, world, td, x3=mode.startswith("x3"), grad_lo=(mode == "bf16_grad_image"))
    W0 = torch.from_numpy(np.random.default_rng(7).uniform(-0.3, 0.3, (Fp, Hp)).astype(np.float32))
    eng.W.copy_(W0); eng.sync_shadows()
    if mode.startswith("x3_allreduce"):        # what dp.make_exchange(kind='auto') picks for the split mode: all-reduce + full optimizer step per rank
        ex = dp.make_exchange(eng, buckets=None if mode == "x3_allreduce_1" else 4)      # default: one bucket; 4 = row bands applied as they arrive
        assert type(ex).__name__ == "AllReduceExchange" and type(dp.make_exchange(eng, kind="sharded")).__name__ == "ShardedExchange"
        assert ex.buckets == (1 if mode == "x3_allreduce_1" else min(4, Fp // 64)) and ex.bounds[0] == 0 and ex.bounds[-1] == Fp
        lo_eng = FakeEngine(Fp, Hp, world, td, x3=True, grad_lo=True)                    # a 16-bit exchange image would leave the flat gradient stale
        with pytest.raises(ValueError, match="grad_lo"):
            dp.AllReduceExchange(lo_eng)
    else:
        ex = dp.ShardedExchange(eng, grad_dtype="bf16" if mode.startswith("bf16_grad") else "fp32", packed=(mode != "three"))
        assert (ex.packed, ex.grad_dtype) == ((False, "fp32") if mode == "x3" else (mode != "three", ex.grad_dtype))
    assert ("encode_w32" in eng.options) == (not mode.startswith("x3"))          # only the split mode keeps the fp32-master encode
    for step in range(3):
        dW, db = _local_grads(Fp, Hp, rank, step)
        eng.grad.zero_()
        eng.grad[:Fp * Hp] = dW.reshape(-1); eng.grad[Fp * Hp:Fp * Hp + Hp + Fp] = db
        if eng.grad_lo is not None:
            eng.grad_lo.zero_(); eng.grad_lo[:Fp] = dW.to(torch.bfloat16)
        ex.step(grad_scale=1.0 / world)
    shadow = eng.W_lo.float().clone(); shadow_t = eng.Wt_lo.float().clone()
    own = (ex.f0, ex.f1, eng.W[ex.f0:ex.f1].clone())
    ex.gather_master()
    dp.barrier()
    out[rank] = dict(W_lo=shadow.numpy(), Wt_lo=shadow_t.numpy(), own=(own[0], own[1], own[2].numpy()), W=eng.W.clone().numpy(),
                     bh=eng.bh.numpy().copy(), bv=eng.bv.numpy().copy(), syncs=eng.shadow_syncs, bands=getattr(eng, "bands_seen", None))


@pytest.mark.parametrize("mode", ["packed", "fp32_shadow", "bf16_grad_image", "bf16_grad_cast", "three", "x3", "x3_allreduce", "x3_allreduce_1"])
@pytest.mark.parametrize("world,Fp", [(2, 640), (3, 640), (3, 128)])       # 3 x 256 rows > 640: ragged last chunk; 3 x 64 > 128: EMPTY last chunk
def test_sharded_exchange_equals_single_process(mode, world, Fp):
    Hp = 128
    port = _free_port()
    mgr = mp.Manager(); out = mgr.dict()
    mp.spawn(_worker, args=(world, port, mode, Fp, Hp, out), nprocs=world, join=True)
    W = np.random.default_rng(7).uniform(-0.3, 0.3, (Fp, Hp)).astype(np.float32).astype(np.float64)
    b = np.zeros(Hp + Fp)
    bf16_grads = mode.startswith("bf16_grad")
    for step in range(3):
        gs = [_local_grads(Fp, Hp, r, step) for r in range(world)]
        W -= LR / world * sum(g[0].double().numpy() for g in gs)
        b -= LR / world * sum(g[1].double().numpy() for g in gs)
    td = torch.float32 if mode == "fp32_shadow" else torch.bfloat16
    tolW = 3e-2 if bf16_grads else 1e-5                             # bf16 gradients: 2^-9 per rank gradient entry, |g| ~ 1, lr 0.1, 3 steps
    want_lo = torch.from_numpy(W).float().to(td).float().numpy()
    for rk in range(world):
        o = out[rk]
        assert np.abs(o["W"] - W).max() <= tolW, (rk, np.abs(o["W"] - W).max())              # gather_master: the full fp32 master everywhere
        f0, f1, rows = o["own"]
        assert np.abs(rows - W[f0:f1]).max() <= tolW if f1 > f0 else rows.size == 0         # before it: the owner's rows were current
        lo_tol = tolW + (2.0 ** -8 * np.abs(W).max() if td == torch.bfloat16 else 0.0)
        assert np.abs(o["W_lo"] - want_lo).max() <= lo_tol
        assert np.array_equal(o["Wt_lo"], o["W_lo"].T)
        assert np.abs(o["bh"] - b[:Hp]).max() <= 1e-5 and np.abs(o["bv"] - b[Hp:]).max() <= 1e-5
        assert np.array_equal(o["W_lo"], out[0]["W_lo"]) and np.array_equal(o["bh"], out[0]["bh"])   # every rank holds IDENTICAL shadows / biases
        assert o["syncs"] == (1 + 3 if mode == "x3" else 1)                                         # (the all-reduce form rebuilds the images inside dae_plan_apply)
        if mode == "x3_allreduce":                                                                  # bucketed: the bands tile [0, Fp) in order
            nb = min(4, Fp // 64)
            assert len(o["bands"]) == nb and o["bands"][0][0] == 0 and o["bands"][-1][1] == Fp
            assert all(o["bands"][i][1] == o["bands"][i + 1][0] for i in range(nb - 1))


def test_bucketed_allreduce_is_bitwise_the_single_bucket_form():
    """Two ranks: a sum of two floats does not depend on how the buffer is cut, so the bucketed exchange must reproduce the one-collective form bit
    for bit -- master weights, both shadows and biases."""
    Fp, Hp, world = 640, 128, 2
    res = {}
    for mode in ("x3_allreduce", "x3_allreduce_1"):
        mgr = mp.Manager(); out = mgr.dict()
        mp.spawn(_worker, args=(world, _free_port(), mode, Fp, Hp, out), nprocs=world, join=True)
        res[mode] = {r: dict(out[r]) for r in range(world)}
    for r in range(world):
        for key in ("W", "W_lo", "Wt_lo", "bh", "bv"):
            assert np.array_equal(res["x3_allreduce"][r][key], res["x3_allreduce_1"][r][key]), (r, key)


def test_native_band_boundaries_equal_the_python_exchange():
    """dae_dp_bands (the row bands dae_dp_exchange reduces and applies, C ABI) == AllReduceExchange.bounds (the torch.distributed form whose bucketed
    arithmetic the gloo tests above pin) for every shape / bucket count in use; pure host arithmetic, no GPU."""
    from dae_rnn_news_recommendation_amd import _lib as L, dp
    for fmt in ("bf16", "f16"):
        lib = L.load(fmt)
        for Fp in (128, 640, 768, 10112, 50048):
            for buckets in (1, 2, 3, 4, 7, 8, 11):
                nblk = Fp // 64
                nb = max(1, min(buckets, nblk, L.COMM_MAX_BUCKETS))
                want = [64 * ((nblk * k) // nb) for k in range(nb)] + [Fp]
                got = dp.native_bands(lib, Fp, buckets)
                assert got == want, (Fp, buckets, got, want)
                assert all(b % 64 == 0 for b in got) and got == sorted(set(got))
