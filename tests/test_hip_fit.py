"""GPU parity of the estimator surface: DenoisingAutoencoder.fit()/transform() vs the oracle's restated
fit loop on identical seeded inputs, injected W0 and the reference's legacy-RNG corruption/shuffle order."""
import numpy as np
import pytest
from scipy import sparse

import oracle as O

pytestmark = pytest.mark.gpu


def _data(N=600, F=900, seed=0, binary=True):
    rng = np.random.default_rng(seed)
    m = sparse.random(N, F, density=0.04, random_state=np.random.RandomState(seed), format="csr", dtype=np.float32)
    m.data = np.ones_like(m.data) if binary else (m.data * 0.9 + 0.1).astype(np.float32)
    m.sort_indices()
    lab = rng.integers(0, 4, N)
    return m, lab


@pytest.mark.parametrize("precision,strategy,opt", [("fp32", "batch_all", "gradient_descent"), ("bf16", "batch_all", "gradient_descent"),
                                                    ("fp32", "batch_hard", "momentum"), ("fp32", "none", "ada_grad")])
def test_fit_loss_curve_matches_oracle(tmp_path, precision, strategy, opt):
    from dae_rnn_news_recommendation_amd.autoencoder import DenoisingAutoencoder
    m, lab = _data()
    F = m.shape[1]; H = F // 10
    W0 = np.random.default_rng(1).uniform(-0.15, 0.15, (F, H)).astype(np.float32)
    kw = dict(compress_factor=10, enc_act_func="sigmoid", dec_act_func="sigmoid", loss_func="cross_entropy", num_epochs=4,
              batch_size=0.25, opt=opt, learning_rate=0.05, momentum=0.5, corr_type="masking", corr_frac=0.3, seed=7, alpha=1,
              triplet_strategy=strategy)
    model = DenoisingAutoencoder(model_name="t", main_dir="t", verbose=False, verbose_step=1, precision=precision, rng="numpy",
                                 init_weights=W0, results_root=str(tmp_path) + "/", **kw)
    model.fit(m, None, lab if strategy != "none" else None)
    ref = O.fit_reference(m, lab if strategy != "none" else None, W0, enc_act="sigmoid", dec_act="sigmoid",
                          loss_func="cross_entropy", num_epochs=4, batch_size=0.25, opt=opt, learning_rate=0.05, momentum=0.5,
                          corr_type="masking", corr_frac=0.3, seed=7, alpha=1.0, triplet_strategy=strategy, dt=np.float64)
    tol = 1e-4            # the loss gate BASELINE.json's north_star states (relative)
    for e in range(4):
        got = model.epoch_stats(e + 1)
        want = ref["history"][e]
        assert abs(got["cost"] - np.mean(want["cost"])) <= tol * abs(np.mean(want["cost"])), (e, got["cost"], np.mean(want["cost"]))
        assert abs(got["ae"] - np.mean(want["ae"])) <= tol * abs(np.mean(want["ae"]))
        if strategy != "none":
            assert abs(got["triplet"] - np.mean(want["triplet"])) <= tol * abs(np.mean(want["triplet"])) + 1e-9
    W, bh, bv = model.engine.get_params()
    wt = 2e-5 if precision == "fp32" else 2e-3
    assert np.abs(W - ref["W"]).max() <= wt * np.abs(ref["W"]).max()
    # transform(): decay-compensated encode of unseen rows (main_autoencoder.py:289)
    emb = model.transform(m[:100] * 0.7)
    want, _ = O.encode((m[:100] * 0.7).toarray(), ref["W"], ref["bh"], "sigmoid", np.float64)
    assert np.abs(emb - want).max() <= (1e-4 if precision == "fp32" else 5e-3)
    # checkpoint round trip through a fresh object (load_model, autoencoder.py:507)
    m2 = DenoisingAutoencoder(model_name="t", main_dir="t", verbose=False, precision=precision, results_root=str(tmp_path) + "/", **kw)
    m2.load_model((F, H), model.model_path)
    assert np.array_equal(m2.get_model_parameters()["enc_w"], W)
    assert np.allclose(m2.transform(m[:100] * 0.7), emb, atol=1e-6)


def test_fit_philox_rng_is_statistically_equivalent(tmp_path):
    """Device Philox masking: same algorithm, different (counter-based) stream -> close but not equal losses."""
    from dae_rnn_news_recommendation_amd.autoencoder import DenoisingAutoencoder
    m, lab = _data(seed=3)
    F = m.shape[1]; H = F // 10
    W0 = np.random.default_rng(1).uniform(-0.15, 0.15, (F, H)).astype(np.float32)
    res = {}
    for rng in ("numpy", "philox"):
        model = DenoisingAutoencoder(model_name=rng, main_dir=rng, verbose=False, verbose_step=1, compress_factor=10,
                                     enc_act_func="sigmoid", dec_act_func="sigmoid", loss_func="cross_entropy", num_epochs=3,
                                     batch_size=0.25, learning_rate=0.05, corr_type="masking", corr_frac=0.3, seed=7,
                                     triplet_strategy="batch_all", precision="fp32", rng=rng, init_weights=W0,
                                     results_root=str(tmp_path) + "/")
        model.fit(m, None, lab)
        res[rng] = [model.epoch_stats(e + 1)["cost"] for e in range(3)]
    for a, b in zip(res["numpy"], res["philox"]):
        assert a != b and abs(a - b) < 0.02 * abs(a)


@pytest.mark.parametrize("corr_type", ["none", "decay", "salt_and_pepper"])
def test_fit_other_corruptions_match_oracle(tmp_path, corr_type):
    from dae_rnn_news_recommendation_amd.autoencoder import DenoisingAutoencoder
    m, lab = _data(N=200, F=300, seed=5, binary=False)
    F = m.shape[1]; H = F // 10
    W0 = np.random.default_rng(1).uniform(-0.2, 0.2, (F, H)).astype(np.float32)
    kw = dict(enc_act="sigmoid", dec_act="sigmoid", loss_func="mean_squared", num_epochs=2, batch_size=50, learning_rate=0.05,
              corr_type=corr_type, corr_frac=0.1, seed=11, alpha=1.0, triplet_strategy="batch_hard")
    model = DenoisingAutoencoder(model_name="c", main_dir="c", verbose=False, verbose_step=1, compress_factor=10,
                                 enc_act_func="sigmoid", dec_act_func="sigmoid", loss_func="mean_squared", num_epochs=2,
                                 batch_size=50, learning_rate=0.05, corr_type=corr_type, corr_frac=0.1, seed=11,
                                 triplet_strategy="batch_hard", precision="fp32", init_weights=W0, results_root=str(tmp_path) + "/")
    model.fit(m, None, lab)
    ref = O.fit_reference(m, lab, W0, dt=np.float64, **kw)
    for e in range(2):
        got = model.epoch_stats(e + 1)["cost"]; want = np.mean(ref["history"][e]["cost"])
        assert abs(got - want) <= 1e-4 * abs(want), (corr_type, e, got, want)


def test_fit_device_salt_and_pepper_matches_oracle(tmp_path):
    """corr_type='salt_and_pepper' with rng='philox': the flips are drawn per batch row on the device (dae_salt_pepper_batch);
    the oracle applies the same Philox flips (oracle.salt_and_pepper_philox) to the whole set, epoch by epoch."""
    from dae_rnn_news_recommendation_amd.autoencoder import DenoisingAutoencoder
    m, lab = _data(N=160, F=400, seed=9, binary=True)
    F = m.shape[1]; H = F // 10
    W0 = np.random.default_rng(2).uniform(-0.2, 0.2, (F, H)).astype(np.float32)
    model = DenoisingAutoencoder(model_name="sp", main_dir="sp", verbose=False, verbose_step=1, compress_factor=10,
                                 enc_act_func="sigmoid", dec_act_func="sigmoid", loss_func="cross_entropy", num_epochs=2,
                                 batch_size=40, learning_rate=0.05, corr_type="salt_and_pepper", corr_frac=0.1, seed=13,
                                 triplet_strategy="batch_all", precision="fp32", rng="philox", init_weights=W0,
                                 results_root=str(tmp_path) + "/")
    model.fit(m, None, lab)
    np.random.seed(13)
    v = int(np.round(0.1 * F))
    plans = []
    for e in range(2):
        xc = sparse.csr_matrix(O.salt_and_pepper_philox(m, np.arange(m.shape[0]), v, 13, e, lo=0.0, hi=1.0))
        plans.append((xc, O.gen_batches_index(m.shape[0], 40)))
    ref = O.fit_reference(m, lab, W0, enc_act="sigmoid", dec_act="sigmoid", loss_func="cross_entropy", num_epochs=2, batch_size=40,
                          learning_rate=0.05, corr_type="salt_and_pepper", corr_frac=0.1, seed=-1, alpha=1.0,
                          triplet_strategy="batch_all", dt=np.float64, plans=plans)
    for e in range(2):
        got = model.epoch_stats(e + 1)["per_batch"][:, 0]
        want = np.array(ref["history"][e]["cost"])
        assert np.abs(got - want).max() <= 2e-5 * np.abs(want).max(), (e, got, want)


def test_function_surface_matches_golden():
    """triplet_loss_utils function names / return tuples, evaluated on the GPU, vs the reference's own outputs."""
    import os
    from dae_rnn_news_recommendation_amd.autoencoder import triplet_loss_utils as T
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_vectors.npz"))
    for case in range(int(G["n_miner_cases"])):
        k = f"miner{case}_"
        lab, h = G[k + "labels"], G[k + "encode"]
        l, dw, fr, num = T.batch_all_triplet_loss(False, lab, h)
        assert np.allclose(l, G[k + "ba_all_loss"], rtol=1e-5, atol=1e-7) and np.array_equal(dw, G[k + "ba_all_dw"])
        assert np.allclose(fr, G[k + "ba_all_frac"], rtol=1e-6) and num == G[k + "ba_all_num"]
        l, dw, fr, num = T.batch_all_triplet_loss(False, lab, h, True)
        assert np.allclose(l, G[k + "ba_pos_loss"], rtol=1e-5, atol=1e-7) and np.array_equal(dw, G[k + "ba_pos_dw"])
        l, dw, fr, num = T.batch_hard_triplet_loss(False, lab, h)
        assert np.allclose(l, G[k + "bh_loss"], rtol=1e-4, atol=1e-6)
        assert (T._get_triplet_mask(lab) == G[k + "mask3"]).all()
        assert (T._get_anchor_positive_triplet_mask(lab) == G[k + "mask_ap"]).all()
        assert (T._get_anchor_negative_triplet_mask(lab) == G[k + "mask_an"]).all()
    for lf in ("cross_entropy", "mean_squared", "cosine_proximity"):
        x = G["wl_xb"] if lf == "cross_entropy" else G["wl_xt"]
        assert np.allclose(T.weighted_loss(False, x, G["wl_y"], lf), G[f"wl_{lf}_unw"], rtol=1e-5)
        assert np.allclose(T.weighted_loss(False, x, G["wl_y"], lf, G["wl_w"]), G[f"wl_{lf}_w"], rtol=1e-5)
        assert np.allclose(T.weighted_loss(True, sparse.csr_matrix(x), G["wl_y"], lf, G["wl_w"]), G[f"wl_{lf}_sparse_w"], rtol=1e-5)
