"""Pin the oracle against golden vectors produced by executing the reference's own
autoencoder/triplet_loss_utils.py and autoencoder/utils.py (tests/golden/make_golden.py)."""
import os

import numpy as np
import pytest
from scipy import sparse

import oracle as O

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_vectors.npz"))


@pytest.mark.parametrize("case", range(int(G["n_miner_cases"])))
def test_miners_vs_reference(case):
    k = f"miner{case}_"
    lab, h = G[k + "labels"], G[k + "encode"]
    assert (O.get_triplet_mask(lab) == G[k + "mask3"]).all()
    assert (O.get_anchor_positive_triplet_mask(lab) == G[k + "mask_ap"]).all()
    assert (O.get_anchor_negative_triplet_mask(lab) == G[k + "mask_an"]).all()
    for pos_only, s in ((False, "all"), (True, "pos")):
        l, dw, fr, num = O.batch_all_triplet_loss(lab, h, pos_only, np.float32)
        assert np.allclose(l, G[k + f"ba_{s}_loss"], rtol=2e-6, atol=1e-7)
        assert (dw == G[k + f"ba_{s}_dw"]).all()          # integer-valued: bit exact
        assert np.allclose(fr, G[k + f"ba_{s}_frac"], rtol=1e-6)
        assert num == G[k + f"ba_{s}_num"]
    nv, dwc = O.batch_all_closed_form(lab)
    assert (dwc == G[k + "ba_all_dw"]).all()
    l, dw, fr, num = O.batch_hard_triplet_loss(lab, h, np.float32)
    assert np.allclose(l, G[k + "bh_loss"], rtol=2e-6, atol=1e-7)
    assert (dw == G[k + "bh_dw"]).all()
    assert np.allclose(fr, G[k + "bh_frac"]) and num == G[k + "bh_num"]


@pytest.mark.parametrize("lf", ["cross_entropy", "mean_squared", "cosine_proximity"])
def test_weighted_loss_vs_reference(lf):
    x = G["wl_xb"] if lf == "cross_entropy" else G["wl_xt"]
    y, w = G["wl_y"], G["wl_w"]
    assert np.allclose(O.weighted_loss(x, y, lf), G[f"wl_{lf}_unw"], rtol=2e-6)
    assert np.allclose(O.weighted_loss(x, y, lf, w), G[f"wl_{lf}_w"], rtol=2e-6)
    assert np.allclose(O.weighted_loss(sparse.csr_matrix(x), y, lf, w), G[f"wl_{lf}_sparse_w"], rtol=2e-6)


def test_noise_and_batching_vs_reference():
    Xd = G["u_X"]; Xs = sparse.csr_matrix(Xd)
    np.random.seed(123)
    assert (O.masking_noise(Xs, 0.3).toarray() == G["u_mask_sparse_seed123"]).all()
    np.random.seed(123)
    assert (O.masking_noise(Xd, 0.3) == G["u_mask_dense_seed123"]).all()
    np.random.seed(7)
    assert (O.salt_and_pepper_noise(Xs, 5).toarray() == G["u_sp_sparse_seed7_v5"]).all()
    np.random.seed(7)
    assert (O.salt_and_pepper_noise(Xd, 5) == G["u_sp_dense_seed7_v5"]).all()
    assert np.allclose(O.decay_noise(Xs, 0.3).toarray(), G["u_decay_sparse"], rtol=0, atol=0)
    assert (O.decay_noise(Xd, 0.3) == G["u_decay_dense"]).all()
    ind, val, shp = O.get_sparse_ind_val_shape(sparse.coo_matrix(Xd))
    assert (ind == G["u_feed_indices"]).all() and (val == G["u_feed_values"]).all()
    assert tuple(shp) == tuple(G["u_feed_shape"])
    for bs, tag in ((4, "bs4"), (0.3, "bs0p3")):
        np.random.seed(42)
        xc, batches = O.epoch_plan(Xs, "masking", 0.3, bs)
        assert (xc.toarray() == G[f"u_epoch_{tag}_xc"]).all()
        assert [len(b) for b in batches] == G[f"u_epoch_{tag}_sizes"].tolist()
        assert sum(batches, []) == G[f"u_epoch_{tag}_order"].tolist()
    np.random.seed(9)
    assert sum(O.gen_batches_index(30, 4), []) == G["u_triplet_bs4_seed9_order"].tolist()
