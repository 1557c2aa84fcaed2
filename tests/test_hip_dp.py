"""Data parallel THROUGH THE ESTIMATOR on the GPU: two ranks (two processes sharing this box's one GPU, gloo collectives --
RCCL needs a GPU per rank) run ``DenoisingAutoencoder.fit(data_parallel=True)`` -- phase-1 HIP step on the rank's shard of
every global mini-batch, reduce-scatter of the W gradient, sharded optimizer (dae_plan_apply_rows), all-gather of W_lo,
local rebuild of Wt_lo -- and must reproduce ONE rank training on the global batches: same per-batch statistics, same
parameters.  Ragged shards (37 rows on 2 ranks = 19 / 18; a 2-row tail batch) are included."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _data():
    from scipy import sparse
    rng = np.random.default_rng(3)
    N, F = 150, 700
    m = sparse.random(N, F, density=0.04, random_state=np.random.RandomState(3), format="csr", dtype=np.float32)
    m.data = np.ones_like(m.data); m.sort_indices()
    lab = rng.integers(0, 3, N)
    W0 = rng.uniform(-0.2, 0.2, (F, F // 10)).astype(np.float32)
    return m, lab, W0


def _fit(tmp, strategy, opt, dp_flag, grad_dtype="fp32", seed=11, mining="local", precision="fp32", exchange="auto"):
    from dae_rnn_news_recommendation_amd.autoencoder import DenoisingAutoencoder
    m, lab, W0 = _data()
    model = DenoisingAutoencoder(model_name="dp", main_dir="dp%d" % os.getpid(), compress_factor=10, enc_act_func="sigmoid",
                                 dec_act_func="sigmoid", loss_func="cross_entropy", num_epochs=2, batch_size=37, opt=opt,
                                 learning_rate=0.05, momentum=0.5, corr_type="masking", corr_frac=0.3, verbose=0, verbose_step=1,
                                 seed=seed, alpha=1, triplet_strategy=strategy, precision=precision, rng="numpy", init_weights=W0,
                                 data_parallel=dp_flag, dp_grad_dtype=grad_dtype, dp_mining=mining, dp_exchange=exchange, results_root=tmp + "/")
    model.fit(m, train_set_label=lab if strategy != "none" else None)
    if dp_flag:
        want = "AllReduceExchange" if (exchange == "allreduce" or (exchange == "auto" and precision in ("auto", "bf16x3", "f16x2"))) else "ShardedExchange"
        assert type(model._exchange).__name__ == want, (type(model._exchange).__name__, want)
    stats = np.stack([model.epoch_stats(e + 1)["per_batch"] for e in range(2)])
    return stats, model.engine.get_params()


def _fit_save_restore(tmp, opt, dp_flag, precision="fp32"):
    """fit 2 epochs (checkpoint written by rank 0) -> a NEW estimator restores it and trains 2 more epochs.  Returns the checkpoint's
    optimizer slots and the final parameters."""
    from dae_rnn_news_recommendation_amd.autoencoder import DenoisingAutoencoder
    m, lab, W0 = _data()
    kw = dict(model_name="ck", main_dir="ck", compress_factor=10, enc_act_func="sigmoid", dec_act_func="sigmoid", loss_func="cross_entropy",
              num_epochs=2, batch_size=37, opt=opt, learning_rate=0.05, momentum=0.5, corr_type="masking", corr_frac=0.3, verbose=0,
              verbose_step=1, seed=11, alpha=1, triplet_strategy="none", precision=precision, rng="numpy", data_parallel=dp_flag,
              results_root=tmp + "/")
    a = DenoisingAutoencoder(init_weights=W0, **kw)
    a.fit(m)
    if dp_flag:
        from dae_rnn_news_recommendation_amd import dp
        dp.barrier()
    z = np.load(a.model_path + ".npz")
    slots = {k: z[k].copy() for k in z.files if k.startswith("opt-")}
    b = DenoisingAutoencoder(init_weights=W0, **kw)
    b.fit(m, restore_previous_model=True)
    return slots, b.engine.get_params()


def _worker_ck(rank, world, port, tmp, opt, out):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LOCAL_RANK="0")
    import torch
    from dae_rnn_news_recommendation_amd import dp
    torch.cuda.set_device(0)
    dp.init_from_env("gloo")
    out[rank] = _fit_save_restore(tmp, opt, True)
    dp.barrier()
    torch.distributed.destroy_process_group()


def _worker(rank, world, port, tmp, strategy, opt, grad_dtype, seed, mining, out, precision="fp32", exchange="auto"):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LOCAL_RANK="0")
    import torch
    from dae_rnn_news_recommendation_amd import dp
    torch.cuda.set_device(0)
    dp.init_from_env("gloo")
    stats, params = _fit(tmp, strategy, opt, True, grad_dtype, seed, mining, precision, exchange)
    out[rank] = (stats, params)
    dp.barrier()
    torch.distributed.destroy_process_group()


def _run_dp(tmp, strategy, opt, grad_dtype="fp32", seed=11, mining="local", precision="fp32", exchange="auto"):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    mgr = ctx.Manager(); out = mgr.dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, tmp, strategy, opt, grad_dtype, seed, mining, out, precision, exchange)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0, "rank exited with %r" % (p.exitcode,)
    return dict(out)


def _rel(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() / (np.abs(b).max() + 1e-30))


@pytest.mark.parametrize("opt", ["gradient_descent", "adam"])
def test_fit_two_ranks_equal_one_rank_strategy_none(tmp_path, opt):
    ref_stats, ref_p = _fit(str(tmp_path), "none", opt, False)
    out = _run_dp(str(tmp_path), "none", opt)
    for r in range(2):
        stats, p = out[r]
        # global-batch cost on every rank.  Adam's first steps are lr * g / (|g| + 1e-8): weights whose gradient is ~0 amplify the
        # rounding difference between "one pass over 37 rows" and "19 rows + 18 rows" -- the SGD case is the exactness check
        assert _rel(stats[..., 0], ref_stats[..., 0]) < (1e-5 if opt == "gradient_descent" else 1e-4), (r, stats[..., 0], ref_stats[..., 0])
        for a, b in zip(p, ref_p):
            assert _rel(a, b) < (1e-5 if opt == "gradient_descent" else 5e-3)
    assert np.array_equal(out[0][1][0], out[1][1][0])                      # both ranks end with the same weights, bit for bit


@pytest.mark.parametrize("strategy,mining,exchange,precision", [("none", "local", "auto", "bf16x3"), ("batch_all", "global", "auto", "bf16x3"),
                                                                ("batch_all", "global", "sharded", "bf16x3"), ("none", "local", "sharded", "bf16x3"),
                                                                ("none", "local", "auto", "f16x2"), ("batch_all", "global", "auto", "f16x2"),
                                                                ("batch_all", "local", "sharded", "f16x2")])
def test_fit_two_ranks_split_bf16_equals_one_rank(tmp_path, strategy, mining, exchange, precision):
    """precision='bf16x3' under data parallel.  Default exchange (dp.AllReduceExchange): ONE all-reduce of the flat fp32 gradient, every rank
    runs the optimizer on the whole W and rebuilds its four bf16 images in the same kernel.  exchange='sharded' (dp.ShardedExchange): fp32
    gradients reduce-scattered, each rank updates its rows of the fp32 master, the MASTER rows all-gathered, images rebuilt.  Either way two
    ranks reproduce one rank at the global batch (and, with dp_mining='global', its triplet leg)."""
    # (f16x2, the fp16 build's split mode = what precision='auto' resolves to: the two half batches round their fp16 images differently from the
    #  whole batch -- the ranks agree with one rank to the mode's own gradient accuracy)
    ref_stats, ref_p = _fit(str(tmp_path), strategy, "gradient_descent", False, precision=precision)
    out = _run_dp(str(tmp_path), strategy, "gradient_descent", mining=mining, precision=precision, exchange=exchange)
    for r in range(2):
        stats, p = out[r]
        if mining == "global" or strategy == "none":
            assert _rel(stats[..., 0], ref_stats[..., 0]) < (2e-5 if precision == "bf16x3" else 1e-4), (r, stats[..., 0], ref_stats[..., 0])
            for a, b in zip(p, ref_p):
                assert _rel(a, b) < (1e-4 if precision == "bf16x3" else 2e-3)
    assert np.array_equal(out[0][1][0], out[1][1][0])


def test_fit_two_ranks_seed_is_shared_when_unseeded(tmp_path):
    """seed < 0: rank 0's entropy is broadcast, so both ranks draw the same masks / shuffles: the shards partition
    every global batch and the ranks stay bit-identical."""
    out = _run_dp(str(tmp_path), "none", "gradient_descent", seed=-1)
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1][0], out[1][1][0])
    assert np.isfinite(out[0][0]).all()


def test_fit_two_ranks_bf16_gradients(tmp_path):
    """perf mode of the exchange: the reduce-scatter carries bf16 gradients (half the bytes); the update stays within bf16
    rounding of the fp32 exchange."""
    ref_stats, ref_p = _fit(str(tmp_path), "none", "gradient_descent", False)
    out = _run_dp(str(tmp_path), "none", "gradient_descent", grad_dtype="bf16")
    stats, p = out[0]
    assert _rel(stats[..., 0], ref_stats[..., 0]) < 2e-3
    assert _rel(p[0], ref_p[0]) < 2e-3


def test_fit_two_ranks_local_mining_runs(tmp_path):
    """triplet strategies under data parallel mine within each rank's shard (documented, warned): finite, decreasing cost and
    identical weights on both ranks."""
    out = _run_dp(str(tmp_path), "batch_all", "gradient_descent")
    assert np.isfinite(out[0][0]).all()
    assert np.array_equal(out[0][1][0], out[1][1][0])
    assert out[0][0][1, :, 0].mean() < out[0][0][0, :, 0].mean()


@pytest.mark.parametrize("strategy", ["batch_all", "batch_hard"])
def test_fit_two_ranks_global_mining_equals_one_rank(tmp_path, strategy):
    """dp_mining='global' (SURVEY 8e mode i): every rank mines ITS anchors against the all-gathered embeddings, the (G + G^T) h
    cross term is exchanged by reduce-scatter, normalisers come from the global batch -- two ranks reproduce ONE rank at the
    global batch size: triplet loss, AE loss (globally normalised data weights), counts and parameters."""
    ref_stats, ref_p = _fit(str(tmp_path), strategy, "gradient_descent", False)
    out = _run_dp(str(tmp_path), strategy, "gradient_descent", mining="global")
    for r in range(2):
        stats, p = out[r]
        assert _rel(stats[..., 2], ref_stats[..., 2]) < 2e-5, (stats[..., 2], ref_stats[..., 2])      # triplet loss
        assert _rel(stats[..., 1], ref_stats[..., 1]) < 2e-5 and _rel(stats[..., 0], ref_stats[..., 0]) < 2e-5
        assert np.abs(stats[..., 4] - ref_stats[..., 4]).max() <= 2                                 # num (near-tie flips of the fp32 Gram)
        for a, b in zip(p, ref_p):
            assert _rel(a, b) < 5e-5
    assert np.array_equal(out[0][1][0], out[1][1][0])


@pytest.mark.parametrize("opt", ["adam", "momentum"])
def test_dp_checkpoint_carries_every_ranks_optimizer_slots(tmp_path, opt):
    """the sharded optimizer updates the slots of W only on the owner of a row chunk; fit() gathers them before rank 0
    saves, so save -> restore -> continue under data parallel equals the same sequence on one rank (slots and parameters)."""
    import torch.multiprocessing as mp
    one = str(tmp_path / "one"); two = str(tmp_path / "two")
    os.makedirs(one); os.makedirs(two)
    ref_slots, ref_p = _fit_save_restore(one, opt, False)
    ctx = mp.get_context("spawn")
    mgr = ctx.Manager(); out = mgr.dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker_ck, args=(r, 2, port, two, opt, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0, "rank exited with %r" % (p.exitcode,)
    slots, params = out[0]
    assert set(slots) == set(ref_slots) and len(slots) >= 1
    tol = 5e-3 if opt == "adam" else 2e-5
    for k in slots:
        n = ref_slots[k].size
        assert _rel(slots[k], ref_slots[k]) < tol, (k, _rel(slots[k], ref_slots[k]))
        # the rows of the OTHER rank's chunk are not the initial value any more
        assert np.count_nonzero(slots[k]) > 0.5 * np.count_nonzero(ref_slots[k])
    for a, b in zip(params, ref_p):
        assert _rel(a, b) < tol


# ---- the collective in the C ABI (include/dae_hip.h: dae_comm_*, dae_allreduce_grads, dae_dp_exchange) on a ONE-rank RCCL communicator ----
@pytest.mark.parametrize("dtype,opt,buckets", [("f16x2", "adam", 1), ("f16x2", "gradient_descent", 4), ("bf16x3", "momentum", 3), ("fp32", "ada_grad", 8)])
def test_native_exchange_on_one_rank_equals_plan_apply(dtype, opt, buckets):
    """dae_dp_exchange over a one-rank communicator (the sum over one rank is the gradient itself) == dae_plan_apply: master weights, biases, slots and
    every 16-bit image bit for bit, for one bucket (all-reduce + apply on the step's stream) and for row bands reduced on the wire stream while the step's
    stream applies them (event-ordered: a missing wait would apply a band before its all-reduce or all-reduce it before the dW GEMM finished)."""
    import numpy as np
    import torch
    from scipy import sparse
    from dae_rnn_news_recommendation_amd import dp
    from dae_rnn_news_recommendation_amd.engine import Engine
    rng = np.random.default_rng(43)
    N, F, H, B = 300, 700, 90, 128
    m = sparse.random(N, F, density=0.06, random_state=np.random.RandomState(3), format="csr", dtype=np.float32); m.data[:] = 1.0; m.sort_indices()
    lab = rng.integers(0, 4, N)
    W0 = rng.uniform(-0.3, 0.3, (F, H)).astype(np.float32)
    idx = torch.from_numpy(rng.permutation(N)[:B].astype(np.int32)).cuda()
    labs = torch.from_numpy(lab[idx.cpu().numpy()].astype(np.int32)).cuda()
    res = []
    comm = None
    for native in (False, True):
        eng = Engine(F, H, B, dtype=dtype, opt=opt, learning_rate=0.05, triplet="batch_all")
        eng.upload_csr(m); eng.set_params(W0)
        stats = torch.zeros(8, device="cuda")
        ex = None
        if native:
            ex = dp.NativeAllReduceExchange(eng, buckets=buckets)
            comm = ex.comm
            assert comm.ranks_seen == 1 and comm.world == 1 and "rccl" in comm.library
            assert ex.buckets == min(buckets, eng.Fp // 64) and ex.bounds[0] == 0 and ex.bounds[-1] == eng.Fp
        for _ in range(3):
            eng.train_step(idx, labs, stats, phase=1)
            if native:
                ex.step(grad_scale=1.0)
            else:
                eng.apply()
        torch.cuda.synchronize()
        it = torch.int16 if eng.td != torch.float32 else torch.int32
        imgs = [eng.W.clone(), eng.bh.clone(), eng.bv.clone(), eng.W_lo.clone().view(it), eng.Wt_lo.clone().view(it), stats.clone()]
        if eng.x3:
            imgs.append(eng.buffer("Wt_lo2", (eng.Hp, eng.Fp), torch.int16).clone())
        if eng.s1 is not None:
            imgs.append(eng.s1.clone())
        res.append(imgs)
    for a, b in zip(*res):
        assert torch.equal(a, b)
    ex.collect_time()
    assert ex.collective_ms > 0 and ex.steps == 3
    comm.close()


def test_native_allreduce_entry_points_on_one_rank():
    """dae_comm_allreduce_f32 (sum / max) and dae_allreduce_grads leave a one-rank buffer unchanged; dae_comm_info reports the communicator; errors are
    reported, not swallowed (a plan with a 16-bit exchange image refuses the flat all-reduce)."""
    import ctypes as C
    import numpy as np
    import torch
    from scipy import sparse
    from dae_rnn_news_recommendation_amd import _lib as L, dp
    from dae_rnn_news_recommendation_amd.engine import Engine
    eng = Engine(600, 64, 128, dtype="bf16", triplet="none", grad_lo=True)
    comm = dp.Comm(eng.lib, eng.device, rank=0, world=1)
    info = (C.c_int32 * 4)()
    L.check(eng.lib.dae_comm_info(comm.handle, info), "dae_comm_info", eng.lib)
    assert list(info)[:2] == [0, 1] and info[2] > 20000 and info[3] == L.COMM_MAX_BUCKETS
    t = torch.arange(1000, dtype=torch.float32, device="cuda") - 17.5
    want = t.clone()
    comm.allreduce_(t); comm.allreduce_(t, op="max")
    torch.cuda.synchronize()
    assert torch.equal(t, want)
    m = sparse.random(200, 600, density=0.05, random_state=np.random.RandomState(1), format="csr", dtype=np.float32); m.data[:] = 1.0
    eng.upload_csr(m)
    rc = eng.lib.dae_allreduce_grads(eng.plan, comm.handle, L.current_stream())
    assert rc == 1 and b"grad_lo" in eng.lib.dae_last_error()
    with pytest.raises(ValueError):
        dp.NativeAllReduceExchange(eng, comm=comm)
    comm.close()
