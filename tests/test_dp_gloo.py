"""Data-parallel exchange logic on CPU (gloo, world_size 2): sharding of every global mini-batch, flat-gradient
all-reduce and 1/world scaling reproduce the single-process gradient of the global batch (strategy none, where
the per-rank mean of row losses averages to the global mean).  The compute leg here is the CPU oracle -- the
test exercises dae_rnn_news_recommendation_amd.dp (what the GPU ranks call around dae_train_step phase 1)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LOCAL_RANK=str(rank))
    import oracle as O
    from dae_rnn_news_recommendation_amd import dp
    dp.init_from_env("gloo")
    assert dp.world_size() == world and dp.rank() == rank
    rng = np.random.default_rng(0)                       # identical data on every rank
    N, F, H, B = 64, 40, 8, 16
    x = (rng.random((N, F)) < 0.2).astype(np.float64)
    W = rng.uniform(-0.3, 0.3, (F, H)); bh = np.zeros(H); bv = np.zeros(F)
    W = dp.broadcast_array(W + rank)                     # rank 1 starts different: broadcast must repair it
    order = np.arange(N); np.random.seed(5); np.random.shuffle(order)
    flat_sizes = (F * H, H, F)
    for start in range(0, N, B):
        lo, hi = dp.shard_bounds(start, start + B, world, rank)
        r = O.forward_backward(W, bh, bv, x[order[lo:hi]], x[order[lo:hi]], None, triplet_strategy="none", dt=np.float64)
        flat = torch.from_numpy(np.concatenate([r["dW"].ravel(), r["dbh"], r["dbv"]]))
        dp.allreduce_sum_(flat)
        g = flat.numpy() / world
        dW = g[:flat_sizes[0]].reshape(F, H); dbh = g[flat_sizes[0]:flat_sizes[0] + H]; dbv = g[flat_sizes[0] + H:]
        W -= 0.1 * dW; bh -= 0.1 * dbh; bv -= 0.1 * dbv
    t = dp.allreduce_max_float(float(rank))
    dp.barrier()
    out[rank] = (W.copy(), bh.copy(), bv.copy(), t)


def test_dp_two_ranks_equal_single_process():
    sys.path.insert(0, ROOT)
    import oracle as O
    world = 2
    port = _free_port()
    mgr = mp.Manager(); out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    rng = np.random.default_rng(0)
    N, F, H, B = 64, 40, 8, 16
    x = (rng.random((N, F)) < 0.2).astype(np.float64)
    W = rng.uniform(-0.3, 0.3, (F, H)); bh = np.zeros(H); bv = np.zeros(F)
    order = np.arange(N); np.random.seed(5); np.random.shuffle(order)
    for start in range(0, N, B):
        idx = order[start:start + B]
        r = O.forward_backward(W, bh, bv, x[idx], x[idx], None, triplet_strategy="none", dt=np.float64)
        W -= 0.1 * r["dW"]; bh -= 0.1 * r["dbh"]; bv -= 0.1 * r["dbv"]
    for rk in range(world):
        Wr, bhr, bvr, t = out[rk]
        assert np.allclose(Wr, W, rtol=1e-12, atol=1e-14) and np.allclose(bhr, bh, atol=1e-14) and np.allclose(bvr, bv, atol=1e-14)
        assert t == world - 1


def test_shard_bounds_cover_batch():
    from dae_rnn_news_recommendation_amd import dp
    for n, w in ((800, 8), (37, 4), (5, 8), (16, 2)):
        spans = [dp.shard_bounds(100, 100 + n, w, r) for r in range(w)]
        assert spans[0][0] == 100 and spans[-1][1] == 100 + n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
