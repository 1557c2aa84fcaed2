"""GPU parity against vectors produced by the REFERENCE'S OWN MODEL CLASSES (tests/golden/make_golden_graph.py executes
autoencoder/autoencoder.py and autoencoder_triplet.py of the reference over a torch-backed ``tensorflow`` stand-in):
the HIP step, ``DenoisingAutoencoder.fit()/transform()`` and ``DenoisingAutoencoderTriplet.fit()`` of this package are
compared with the reference's outputs DIRECTLY (no oracle in between)."""
import json
import os

import numpy as np
import pytest
import torch
from scipy import sparse

pytestmark = pytest.mark.gpu

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_graph_vectors.npz"))


def _rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def _cfg(tag):
    return json.loads(str(G[tag + "_cfg"]))


def _bits(keep):
    b = np.packbits(np.asarray(keep, bool).ravel(), bitorder="little")
    b = np.concatenate([b, np.zeros((-len(b)) % 4, np.uint8)]).view(np.int32)
    return torch.from_numpy(b.copy()).cuda()


def _step(ci, dtype):
    from dae_rnn_news_recommendation_amd import _lib as L
    from dae_rnn_news_recommendation_amd.engine import Engine
    k = f"S{ci}_"; c = _cfg(f"S{ci}")
    x, xc = G[k + "x"], G[k + "xc"]
    B, F = x.shape; H = G[k + "W0"].shape[1]
    eng = Engine(F, H, B, dtype=dtype, enc_act=c["enc"], dec_act=c["dec"], loss_func=c["loss"], opt="gradient_descent",
                 learning_rate=0.1, alpha=c["alpha"], triplet=c["strategy"])
    kw = {}
    if c["kind"] == "sparse":
        eng.upload_csr(sparse.csr_matrix(x))
        kw["corrupted_csr"] = Engine.to_device_csr(sparse.csr_matrix(xc), eng.device)
    else:
        eng.upload_dense(x)
        kw.update(corr_mode=L.CORR_KEEPBITS, keep_bits=_bits(~((x != 0) & (xc == 0))))
    eng.set_params(G[k + "W0"], G[k + "bh0"], G[k + "bv0"])
    stats = torch.zeros(8, device="cuda")
    lab = torch.from_numpy(G[k + "labels"].astype(np.int32)).cuda() if c["strategy"] != "none" else None
    eng.train_step(torch.arange(B, dtype=torch.int32, device="cuda"), lab, stats, phase=0, **kw)
    torch.cuda.synchronize()
    return c, k, stats.cpu().numpy(), eng.grads()


@pytest.mark.parametrize("ci", range(int(G["S_n"])))
def test_hip_step_fp32_vs_reference_graph(ci):
    """cost legs and tied-weight gradients of one step == the reference graph's (autoencoder.py:371-442, tf.gradients)."""
    c, k, st, (dW, dbh, dbv) = _step(ci, "fp32")
    if c["strategy"] == "none":
        assert abs(st[0] - G[k + "cost"]) <= 2e-5 * abs(G[k + "cost"]), (c, st, G[k + "cost"])
    else:
        ae, tl = float(G[k + "ae"]), float(G[k + "triplet"])
        assert abs(st[1] - ae) <= 2e-5 * abs(ae), (c, st, ae)
        assert abs(st[2] - tl) <= 2e-5 * abs(tl) + 1e-9, (c, st, tl)
        assert abs(st[0] - G[k + "cost"]) <= 2e-5 * (abs(ae) + abs(c["alpha"] * tl))      # the legs may cancel (cosine)
        assert st[4] == G[k + "num"] and abs(st[3] - G[k + "fraction"]) <= 1e-6
    assert _rel(dW, G[k + "dW"]) < 5e-5 and _rel(dbh, G[k + "dbh"]) < 5e-5 and _rel(dbv, G[k + "dbv"]) < 5e-5


@pytest.mark.parametrize("ci", [i for i in range(int(G["S_n"])) if _cfg(f"S{i}")["loss"] == "cross_entropy"])
def test_hip_step_bf16_vs_reference_graph(ci):
    """bf16 MFMA operands (W, x~, h, delta rounded to 8 bits of mantissa): the north star's 1e-4 loss gate; gradients to
    bf16 operand precision."""
    c, k, st, (dW, dbh, dbv) = _step(ci, "bf16")
    if c["strategy"] == "none":
        assert abs(st[0] - G[k + "cost"]) <= 1e-3 * abs(G[k + "cost"])       # W0 here is NOT bf16-representable (|W| up to 0.6)
    else:
        assert abs(st[1] - G[k + "ae"]) <= 1e-3 * abs(G[k + "ae"])
    assert _rel(dW, G[k + "dW"]) < 2e-2 and _rel(dbv, G[k + "dbv"]) < 2e-2


def _fit_model(tag, precision, tmp_path):
    from dae_rnn_news_recommendation_amd.autoencoder import DenoisingAutoencoder
    c = _cfg(tag)
    X = G[tag + "_X"]
    Xin = X if c["kind"] == "dense" else sparse.csr_matrix(X)
    m = DenoisingAutoencoder(model_name=tag, main_dir=tag, compress_factor=c["compress_factor"], enc_act_func=c["enc"],
                             dec_act_func=c["dec"], loss_func=c["loss"], num_epochs=c["epochs"], batch_size=c["bs"], opt=c["opt"],
                             learning_rate=c["lr"], momentum=0.5, corr_type=c["corr"], corr_frac=c["frac"], verbose=0, verbose_step=1,
                             seed=c["seed"], alpha=c["alpha"], triplet_strategy=c["strategy"], precision=precision, rng="numpy",
                             init_weights=G[tag + "_W0"].astype(np.float32), results_root=str(tmp_path) + "/")
    m.fit(Xin, train_set_label=G[tag + "_labels"] if c["strategy"] != "none" else None)
    return c, m, Xin


@pytest.mark.parametrize("tag", json.loads(str(G["F_tags"])))
def test_fit_fp32_vs_reference_fit(tag, tmp_path):
    """DenoisingAutoencoder.fit() here == DenoisingAutoencoder.fit() of the reference, batch by batch: same legacy-RNG
    corruption and shuffle stream (seed), same W0 -> per-batch cost/ae/triplet within 1e-4 (fp32: 2e-5), same final
    parameters, same transform()."""
    c, m, Xin = _fit_model(tag, "fp32", tmp_path)
    for e in range(c["epochs"]):
        pb = m.epoch_stats(e + 1)["per_batch"]
        assert _rel(pb[:, 0], G[tag + "_cost"][e]) < 2e-5, (tag, e, pb[:, 0], G[tag + "_cost"][e])
        if c["strategy"] != "none":
            assert _rel(pb[:, 1], G[tag + "_ae"][e]) < 2e-5 and _rel(pb[:, 2], G[tag + "_triplet"][e]) < 2e-5
            assert np.array_equal(pb[:, 4], G[tag + "_num"][e])
    W, bh, bv = m.engine.get_params()
    assert _rel(W, G[tag + "_W"]) < 2e-5 and _rel(bv, G[tag + "_bv"]) < 2e-5 and _rel(bh, G[tag + "_bh"]) < 1e-4
    assert _rel(m.transform(Xin), G[tag + "_transform"]) < 2e-5
    p = m.get_model_parameters()
    assert np.array_equal(p["enc_w"], W) and np.array_equal(p["enc_b"], bh) and np.array_equal(p["dec_b"], bv)


@pytest.mark.parametrize("tag", ["F0", "F1"])
def test_fit_bf16_vs_reference_fit(tag, tmp_path):
    """bench precision (bf16 operands): the epoch means the reference prints stay within the 1e-4 gate (strategy none and
    batch_all; batch_hard selects ONE hardest pair per anchor, so bf16 rounding of h flips selections on these 12-row batches
    and the curve follows a different -- equally valid -- sequence of hard pairs: checked at 1e-3 in the next test)."""
    c, m, _ = _fit_model(tag, "bf16", tmp_path)
    for e in range(c["epochs"]):
        got = m.epoch_stats(e + 1)["cost"]; want = float(np.mean(G[tag + "_cost"][e]))
        assert abs(got - want) <= 1e-4 * abs(want), (tag, e, got, want)


def test_fit_bf16_batch_hard_vs_reference_fit(tmp_path):
    c, m, _ = _fit_model("F2", "bf16", tmp_path)
    for e in range(c["epochs"]):
        got = m.epoch_stats(e + 1)["cost"]; want = float(np.mean(G["F2_cost"][e]))
        assert abs(got - want) <= 1e-3 * abs(want), (e, got, want)


@pytest.mark.parametrize("tag", json.loads(str(G["T_tags"])))
def test_explicit_triplet_fit_vs_reference(tag, tmp_path):
    """DenoisingAutoencoderTriplet.fit() == the reference class' fit (autoencoder_triplet.py:40-146), batch by batch."""
    from dae_rnn_news_recommendation_amd.autoencoder import DenoisingAutoencoderTriplet
    c = _cfg(tag)
    data = {k: sparse.csr_matrix(G[tag + "_X_" + k]) for k in ("org", "pos", "neg")}
    m = DenoisingAutoencoderTriplet(model_name=tag, main_dir=tag, compress_factor=c["compress_factor"], enc_act_func=c["enc"],
                                    dec_act_func=c["dec"], loss_func=c["loss"], num_epochs=c["epochs"], batch_size=c["bs"],
                                    learning_rate=c["lr"], corr_type=c["corr"], corr_frac=c["frac"], verbose=0, verbose_step=1,
                                    seed=c["seed"], alpha=c["alpha"], precision="fp32", rng="numpy",
                                    init_weights=G[tag + "_W0"].astype(np.float32), results_root=str(tmp_path) + "/")
    m.fit(data)
    for e in range(c["epochs"]):
        pb = m.epoch_stats(e + 1)["per_batch"]
        assert _rel(pb[:, 0], G[tag + "_cost"][e]) < 2e-5, (tag, e, pb[:, 0], G[tag + "_cost"][e])
        assert _rel(pb[:, 1], G[tag + "_ae"][e]) < 2e-5 and _rel(pb[:, 2], G[tag + "_triplet"][e]) < 2e-5
    W, bh, bv = m.engine.get_params()
    assert _rel(W, G[tag + "_W"]) < 2e-5 and _rel(bv, G[tag + "_bv"]) < 2e-5 and _rel(bh, G[tag + "_bh"]) < 1e-4
