"""CPU tests of the host-side mirror of autoencoder/utils.py: golden vectors produced by the reference's own
utils.py (tests/golden/make_golden.py) + the property tests of the reference's autoencoder/tests/test_utils.py."""
import os

import numpy as np
import pandas as pd
import pytest
from scipy import sparse

from dae_rnn_news_recommendation_amd.autoencoder import utils as U

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_vectors.npz"))


def test_noise_functions_match_reference_streams():
    Xd = G["u_X"]; Xs = sparse.csr_matrix(Xd)
    np.random.seed(123)
    m = U.masking_noise(Xs, 0.3)
    assert sparse.issparse(m) and (m.toarray() == G["u_mask_sparse_seed123"]).all()
    np.random.seed(123)
    assert (U.masking_noise(Xd, 0.3) == G["u_mask_dense_seed123"]).all()
    np.random.seed(7)
    assert (U.salt_and_pepper_noise(Xs, 5).toarray() == G["u_sp_sparse_seed7_v5"]).all()
    np.random.seed(7)
    assert (U.salt_and_pepper_noise(Xd, 5) == G["u_sp_dense_seed7_v5"]).all()
    assert (U.decay_noise(Xs, 0.3).toarray() == G["u_decay_sparse"]).all()
    assert (U.decay_noise(Xd, 0.3) == G["u_decay_dense"]).all()
    ind, val, shp = U.get_sparse_ind_val_shape(sparse.coo_matrix(Xd))
    assert (ind == G["u_feed_indices"]).all() and (val == G["u_feed_values"]).all() and tuple(shp) == tuple(G["u_feed_shape"])


def test_keep_bits_are_the_reference_masking_decisions():
    """fit()'s epoch plan: keep bits == np.random.rand(nnz) >= v in CSR storage order, then the shuffle."""
    Xd = G["u_X"]; Xs = sparse.csr_matrix(Xd)
    for bs, tag in ((4, "bs4"), (0.3, "bs0p3")):
        np.random.seed(42)
        keep = U.masking_keep(Xs.nnz, 0.3)
        order = U.epoch_permutation(Xs.shape[0])
        xc = Xs.copy(); xc.data = xc.data * keep
        assert (xc.toarray() == G[f"u_epoch_{tag}_xc"]).all()
        assert order.tolist() == G[f"u_epoch_{tag}_order"].tolist()
        bits = U.pack_keep_bits(keep)
        unpacked = np.unpackbits(bits.view(np.uint8), bitorder="little")[:Xs.nnz].astype(bool)
        assert (unpacked == keep).all()


def test_ndarray_shuffle_equals_list_shuffle():
    for n in (1, 2, 30, 8000):
        np.random.seed(n)
        a = list(range(n)); np.random.shuffle(a)
        np.random.seed(n)
        assert U.epoch_permutation(n).tolist() == a


def test_gen_batches_golden_order():
    ident = np.arange(30, dtype=np.float32).reshape(-1, 1)
    np.random.seed(42)
    U.masking_keep(sparse.csr_matrix(G["u_X"]).nnz, 0.3)            # same draws as the golden epoch
    order = []
    for b in U.gen_batches(sparse.csr_matrix(ident), sparse.csr_matrix(ident), 4, data_label=np.arange(30)):
        order.extend(b[2].tolist())
    assert order == G["u_epoch_bs4_order"].tolist()
    np.random.seed(9)
    d3 = {k: ident.copy() for k in ("org", "pos", "neg")}
    order = []
    for a, _ in U.gen_batches_triplet(d3, d3, 4):
        order.extend(a[0][:, 0].astype(int).tolist())
    assert order == G["u_triplet_bs4_seed9_order"].tolist()


def test_gen_batches_properties():
    # reference autoencoder/tests/test_utils.py:11-61
    num_data = 30
    data = np.arange(num_data).reshape((-1, 1)).astype(np.float32)
    data_corrupted = np.random.randint(0, 2, (num_data, 10)).astype(np.float32)
    data_label = np.random.randint(0, 10, num_data).astype(np.float32)
    for label in [None, data_label, data_label.reshape((-1, 1)), pd.Series(data_label), pd.DataFrame(data_label)]:
        for func in [lambda x: x, sparse.csr_matrix, pd.DataFrame]:
            in_data = func(data); in_corr = func(data_corrupted)
            if isinstance(in_data, pd.DataFrame):
                in_data.index = np.random.choice(num_data * 2, num_data, replace=False)
                in_corr.index = in_data.index
            if isinstance(label, (pd.DataFrame, pd.Series)):
                label.index = np.random.choice(num_data * 2, num_data, replace=False)
            for batch_size in [4, 0.3]:
                seen = np.zeros(num_data)
                for res in U.gen_batches(in_data, in_corr, batch_size=batch_size, data_label=label):
                    a, b = res[0], res[1]
                    if sparse.issparse(a):
                        a, b = a.toarray(), b.toarray()
                    if isinstance(in_data, pd.DataFrame):
                        idx = a.loc[:, 0].astype(int).tolist()
                        assert (data_corrupted[idx, :] == b.values).all()
                    else:
                        idx = list(a[:, 0].astype(int))
                        assert (data_corrupted[idx, :] == b).all()
                    if label is not None:
                        got = res[2].values if isinstance(label, (pd.DataFrame, pd.Series)) else res[2]
                        want = label.iloc[idx].values if isinstance(label, (pd.DataFrame, pd.Series)) else label[idx]
                        assert (np.asarray(got) == np.asarray(want)).all()
                    seen[idx] += 1
                assert (seen == 1).all()


def test_masking_noise_properties():
    # reference autoencoder/tests/test_utils.py:108-125
    X = sparse.csr_matrix(np.random.rand(10, 10000).astype(np.float32))
    for in_X in [X, X.toarray()]:
        for prob in [0., 0.3, 1.]:
            Xm = sparse.csr_matrix(U.masking_noise(in_X, prob))
            if prob == 0.:
                assert (X != Xm).nnz == 0
            elif prob == 1.:
                assert Xm.nnz == 0
            else:
                assert abs(Xm.nnz / X.nnz - (1. - prob)) <= 1e-2
                assert (Xm.multiply(X != 0) != Xm).nnz == 0


def test_gen_batches_triplet_accepts_fractional_size():
    """The reference crashes here (no int() cast, utils.py:86-90); the mirror must not."""
    d = {k: np.arange(20, dtype=np.float32).reshape(-1, 1) for k in ("org", "pos", "neg")}
    sizes = [a[0].shape[0] for a, _ in U.gen_batches_triplet(d, d, 0.25)]
    assert sizes == [5, 5, 5, 5]


def test_xavier_bound():
    w = U.xavier_init(1000, 50, 2)
    b = 2 * np.sqrt(6.0 / 1050)
    assert w.shape == (1000, 50) and w.dtype == np.float32 and np.abs(w).max() <= b and np.abs(w).max() > 0.9 * b


def test_label_masks_match_golden():
    from dae_rnn_news_recommendation_amd.autoencoder import triplet_loss_utils as T
    for case in range(int(G["n_miner_cases"])):
        lab = G[f"miner{case}_labels"]
        assert (T._get_triplet_mask(lab) == G[f"miner{case}_mask3"]).all()
        assert (T._get_anchor_positive_triplet_mask(lab) == G[f"miner{case}_mask_ap"]).all()
        assert (T._get_anchor_negative_triplet_mask(lab) == G[f"miner{case}_mask_an"]).all()


def test_similar_articles_matches_reference_golden():
    """datasets.articles.similar_articles (reference datasets/articles.py:83-128) vs vectors produced by the reference's own
    function (tests/golden/make_golden.py::golden_similar_articles): positives, negatives (same global-RNG draws) and the
    valid flag, including min_cate / max_cate filtering."""
    import os
    pd = pytest.importorskip("pandas")
    from dae_rnn_news_recommendation_amd.datasets import similar_articles
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "similar_articles.npz"))
    for k in range(int(g["n_cases"])):
        n, _, seed, min_cate, max_cate = [int(v) for v in g[f"c{k}_cfg"]]
        df = pd.DataFrame({"article_id": np.arange(1, n + 1), "label": g[f"c{k}_label"]})
        np.random.seed(seed)
        res = similar_articles(df, id_colname="article_id", cate_colname="label", min_cate=min_cate,
                               max_cate=None if max_cate < 0 else max_cate)
        assert np.array_equal(res["article_id_pos"].to_numpy(), g[f"c{k}_pos"])
        assert np.array_equal(res["article_id_neg"].to_numpy(), g[f"c{k}_neg"])
        assert np.array_equal(res["valid_triplet_data"].to_numpy(), g[f"c{k}_valid"])
        lab = g[f"c{k}_label"]
        v = res["valid_triplet_data"].to_numpy() == 1
        assert v.any()
        assert (lab[res["article_id_pos"].to_numpy()[v] - 1] == lab[v]).all()       # positives share the label
        assert (lab[res["article_id_neg"].to_numpy()[v] - 1] != lab[v]).all()       # negatives do not


def test_save_file_read_file_round_trips(tmp_path):
    """helpers.save_file / read_file (reference helpers.py:138-264): format from the extension, reader / writer from the
    container type, the reference's assert on unsupported combinations."""
    from dae_rnn_news_recommendation_amd import helpers as H
    d = str(tmp_path) + "/"
    a = np.arange(12.).reshape(3, 4)
    for ext in ("npy", "csv", "tsv"):
        H.save_file(a, d + "a." + ext)
        assert np.allclose(H.read_file(d + "a." + ext, data_type="numpy"), a)
    m = sparse.random(5, 7, density=0.3, format="csr", dtype=np.float32, random_state=np.random.RandomState(0))
    H.save_file(m, d + "m.npz")
    assert (H.read_file(d + "m.npz") != m).nnz == 0
    H.save_file(m, d + "m.tsv")                                       # text formats densify the sparse matrix first
    assert np.allclose(H.read_file(d + "m.tsv", data_type="scipy").toarray(), m.toarray())
    s = pd.Series([3, 1, 2], name="label_story")
    H.save_file(s, d + "s.pkl")
    assert H.read_file(d + "s.pkl", data_type="pandas_series").equals(s)
    df = pd.DataFrame({"a": [1, 2], "b": [3.5, 4.5]})
    H.save_file(df, d + "d.pkl")
    assert H.read_file(d + "d.pkl").equals(df)
    H.save_file(df, d + "d.tsv")
    assert np.allclose(H.read_file(d + "d.tsv").to_numpy(), df.to_numpy())
    with pytest.raises(AssertionError):
        H.save_file(m, d + "m.npy")                                   # scipy matrices only go to .npz
    with pytest.raises(AssertionError):
        H.read_file(d + "missing.npy")


def test_class_sort_batches_keeps_every_batch_as_a_set():
    """fit() hands every mini-batch over sorted by label: the rows of a batch are exactly those of the reference's shuffle."""
    from dae_rnn_news_recommendation_amd.autoencoder import utils
    rng = np.random.default_rng(0)
    order = rng.permutation(1003); lab = rng.integers(0, 5, 1003)
    out = utils.class_sort_batches(order, lab, 250)
    assert sorted(out) == sorted(order)
    for s in range(0, 1003, 250):
        assert sorted(out[s:s + 250]) == sorted(order[s:s + 250])
        assert (np.diff(lab[out[s:s + 250]]) >= 0).all()
    # stable: rows of one class keep their shuffled order
    first = out[:250]; want = [r for c in range(5) for r in order[:250] if lab[r] == c]
    assert list(first) == want


def test_auto_precision_resolution_per_strategy_and_fp16_range(tmp_path):
    """precision='auto' resolves PER TRIPLET STRATEGY (_lib.AUTO_BY_STRATEGY: the cheapest mode measured to hold the reference's curve over 100 steps -- none
    f16x2d, batch_all / batch_hard f16x2h, explicit triplets f16x2d), with or without a train set to look at; data beyond the fp16 range takes the
    split-bf16 mode; every input container (ndarray, sparse, list / dict of matrices, validation sets) goes through the fp16 range check."""
    from scipy import sparse
    from dae_rnn_news_recommendation_amd import _lib as L
    from dae_rnn_news_recommendation_amd.autoencoder import DenoisingAutoencoder, DenoisingAutoencoderTriplet
    kw = dict(results_root=str(tmp_path) + "/", verbose=False)
    assert L.AUTO_BY_STRATEGY == {"none": "f16x2d", "batch_all": "f16x2h", "batch_hard": "f16x2h", "explicit": "f16x2d"}
    assert all(v in L.PRECISIONS for v in L.AUTO_BY_STRATEGY.values()) and L.AUTO_PRECISION == "f16x2h"
    assert L.PRECISIONS["f16x2h"] == ("f16", 2, 1 | 2 | 4 | 32 | 64) and L.PRECISIONS["f16x2d"] == ("f16", 2, 1 | 4 | 8 | 128)
    x = sparse.random(20, 30, density=0.2, format="csr", dtype=np.float32, random_state=np.random.RandomState(0))
    big = x.copy(); big.data[:] = 3.0e4
    huge = x.copy(); huge.data[:] = 7.0e4
    for strategy, want in (("none", "f16x2d"), ("batch_all", "f16x2h"), ("batch_hard", "f16x2h")):
        m = DenoisingAutoencoder(model_name="p" + strategy, main_dir="p" + strategy, triplet_strategy=strategy, **kw)
        assert m._resolve_precision(None) == want and m._resolve_precision(x) == want and m._resolve_precision(x.toarray()) == want, strategy
        assert m._resolve_precision(big) == "bf16x3" and m._resolve_precision(big.toarray()) == "bf16x3"
    mt = DenoisingAutoencoderTriplet(model_name="pt", main_dir="pt", **kw)
    assert mt._resolve_precision(None) == "f16x2d" and mt._resolve_precision({"org": x, "pos": x, "neg": big}) == "bf16x3"
    m = DenoisingAutoencoder(model_name="p", main_dir="p", **kw)
    m2 = DenoisingAutoencoder(model_name="p2", main_dir="p2", precision="fp32", **kw)
    assert m2._resolve_precision(big) == "fp32"
    # a model that stores fp16 images (load_model() resolves 'auto' with no data to look at) refuses values fp16 cannot hold, loudly -- whatever the container
    m._check_storage_range(x); m._check_storage_range(x.toarray()); m._check_storage_range(big); m._check_storage_range([x, big]); m._check_storage_range({"a": x})
    for bad in (huge.toarray(), huge, [x, huge], {"org": x, "neg": huge.toarray()}):
        with pytest.raises(ValueError, match="bf16x3"):
            m._check_storage_range(bad)
    assert m._forward_precision(huge) == "bf16x3"          # nothing trained yet: 'auto' still looks at the data
    m.precision_used = "f16x2h"                            # ... after fit() the training precision is fixed: a validation set it cannot hold is refused
    with pytest.raises(ValueError, match="validation"):
        m._forward_precision(huge)
    assert m._forward_precision(x) == "f16x2h"
    m2._check_storage_range(huge)
    DenoisingAutoencoder(model_name="p3", main_dir="p3", precision="bf16x3", **kw)._check_storage_range(huge)
