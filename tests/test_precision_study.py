"""CPU check of tools/precision_study.py, the host replay that decided which GEMM operands the split-bf16 mode stores as hi + lo:
on a small problem the all-split replay must sit orders of magnitude closer to the fp32 replay than the all-bf16 one, and the
split product itself must carry ~2^-16 operands."""
import importlib.util
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def _load():
    spec = importlib.util.spec_from_file_location("precision_study", os.path.join(HERE, "..", "tools", "precision_study.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_split_product_is_two_orders_closer_than_bf16():
    ps = _load()
    g = torch.Generator().manual_seed(3)
    a, b = torch.randn(64, 512, generator=g), torch.randn(512, 48, generator=g)
    ref = a.double() @ b.double()
    e_bf = (ps.mm(a, b, "bf16", "bf16").double() - ref).abs().max() / ref.abs().max()
    e_sp = (ps.mm(a, b, "split", "split").double() - ref).abs().max() / ref.abs().max()
    e_32 = (ps.mm(a, b, "f32", "f32").double() - ref).abs().max() / ref.abs().max()
    assert e_32 < 1e-6 and e_sp < 5e-5 and e_bf > 50 * e_sp, (float(e_32), float(e_sp), float(e_bf))


def test_curve_replay_orders_the_modes():
    ps = _load()
    from dae_rnn_news_recommendation_amd.synthetic import synthetic_csr, synthetic_labels, xavier_uniform
    torch.set_num_threads(4)
    N, F, B, steps = 600, 800, 150, 4
    data = synthetic_csr(N, F, nnz_per_row=40, seed=5).tocsr()
    labels = synthetic_labels(N, seed=5)
    W0 = xavier_uniform(F, F // 20, seed=1).astype(np.float32)
    ops = ("h", "W", "d2", "d1", "Gs")
    ref = ps.run(dict.fromkeys(ops, "f32"), data, labels, W0, steps, B)
    bf = ps.run(dict.fromkeys(ops, "bf16"), data, labels, W0, steps, B)
    sp = ps.run(dict.fromkeys(ops, "split"), data, labels, W0, steps, B)
    d_bf = (np.abs(bf - ref) / np.abs(ref)).max()
    d_sp = (np.abs(sp - ref) / np.abs(ref)).max()
    assert np.isfinite(ref).all() and ref[-1, 0] < ref[0, 0]            # the replay trains
    assert d_sp < 1e-5 and d_bf > 10 * d_sp, (d_bf, d_sp)


def test_fp16_storage_modes_and_golden_order():
    """The fp16 storage modes of the replay (round 4: the scheme study behind DESIGN 11.0) and its golden batch order."""
    ps = _load()
    g = torch.Generator().manual_seed(5)
    a, b = torch.randn(64, 512, generator=g) * 1e-5, torch.randn(512, 48, generator=g)      # delta2-sized entries: far below fp16's normal range
    ref = a.double() @ b.double()
    err = lambda ma, mb: float((ps.mm(a, b, ma, mb).double() - ref).abs().max() / ref.abs().max())
    e16, e16s, ebf, eraw = err("f16", "f16"), err("f16split", "f16split"), err("bf16", "bf16"), err("f16raw", "f16")
    assert e16 < ebf / 4 and e16s < 1e-5 and e16s < e16 / 50, (e16, e16s, ebf)           # 11 bits vs 8; the split carries ~22
    assert eraw > 5 * e16, (eraw, e16)                                                    # without the power-of-two scale the small operand sits in the subnormals
    # golden order: per epoch the keep decisions of the whole set, then the shuffle, both from NumPy's legacy global stream seeded with 0
    from dae_rnn_news_recommendation_amd.synthetic import synthetic_csr, synthetic_labels, xavier_uniform
    torch.set_num_threads(4)
    N, F, B = 300, 400, 100
    data = synthetic_csr(N, F, nnz_per_row=30, seed=7).tocsr(); labels = synthetic_labels(N, seed=7); W0 = xavier_uniform(F, F // 20, seed=1).astype(np.float32)
    ops = ("h", "W", "d2", "d1", "Gs")
    r1 = ps.run(dict.fromkeys(ops, "f32"), data, labels, W0, 4, B, golden=True)
    r2 = ps.run(dict.fromkeys(ops, "f32"), data, labels, W0, 4, B, golden=True)
    assert np.array_equal(r1, r2) and np.isfinite(r1).all()                               # the global stream is re-seeded per run: reproducible
