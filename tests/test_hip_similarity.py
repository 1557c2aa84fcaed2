"""GPU parity of the evaluation step after the path: dae_pairwise_similarity through the mirror of
helpers.pairwise_similarity vs the oracle (which test_oracle.py pins against scikit-learn)."""
import numpy as np
import pytest
import torch
from scipy import sparse

import oracle as O

pytestmark = pytest.mark.gpu


def _close(got, want, tol=1e-5):
    """fp32 product vs the fp64 oracle: error relative to the largest entry (entries near zero are sums that cancel)."""
    want = np.asarray(want, np.float64)
    return float(np.max(np.abs(np.asarray(got, np.float64) - want))) <= tol * float(np.max(np.abs(want)) + 1e-30)


@pytest.mark.parametrize("norm", ["", "l1", "l2", "max"])
@pytest.mark.parametrize("metric", ["cosine", "linear kernel"])
def test_pairwise_similarity_dense(norm, metric):
    from dae_rnn_news_recommendation_amd import helpers
    rng = np.random.default_rng(2)
    X = rng.standard_normal((300, 70)).astype(np.float32)            # ragged against every tile size
    X[11] = 0.0
    got = helpers.pairwise_similarity(X, norm=norm, metric=metric)
    want = O.pairwise_similarity(X, norm=norm, metric=metric)
    assert got.shape == (300, 300) and got.dtype == np.float32
    assert _close(got, want)
    assert (np.diag(got) == 0).all()
    keep = helpers.pairwise_similarity(X, norm=norm, metric=metric, set_diagonal_zero=False)
    assert _close(keep, O.pairwise_similarity(X, norm=norm, metric=metric, set_diagonal_zero=False))
    assert _close(got, got.T, 1e-6)                                   # symmetric, as X X^T must be


def test_pairwise_similarity_sparse_bow_and_embeddings():
    """The two shapes main_autoencoder.py:307-317 feeds: a sparse binary BoW matrix and a dense embedding matrix."""
    from dae_rnn_news_recommendation_amd import helpers
    rng = np.random.default_rng(3)
    bow = sparse.random(500, 3000, density=0.02, format="csr", dtype=np.float32, random_state=np.random.RandomState(1))
    bow.data[:] = 1.0
    got = helpers.pairwise_similarity(bow, metric="cosine")
    assert _close(got, O.pairwise_similarity(bow, metric="cosine"))
    emb = rng.random((500, 500)).astype(np.float32)
    t = helpers.pairwise_similarity(torch.from_numpy(emb).cuda(), metric="cosine", return_tensor=True)
    assert t.is_cuda and t.shape == (500, 500)
    assert _close(t.cpu().numpy(), O.pairwise_similarity(emb, metric="cosine"))


def test_pairwise_similarity_rejects_other_metrics():
    from dae_rnn_news_recommendation_amd import helpers
    with pytest.raises(AssertionError):
        helpers.pairwise_similarity(np.eye(4, dtype=np.float32), metric="euclidean")
    with pytest.raises(ValueError):
        helpers.pairwise_similarity(np.eye(4, dtype=np.float32), norm="l3")


@pytest.mark.parametrize("n,classes,missing", [(300, 5, True), (129, 2, False), (64, 64, False)])
def test_pair_stats_matches_oracle(n, classes, missing):
    """dae_pair_stats (AUROC + box-plot numbers of related vs unrelated pairs) vs the oracle that test_oracle.py pins against
    scikit-learn; tied scores (coarse embeddings) and missing labels included."""
    from dae_rnn_news_recommendation_amd import helpers
    rng = np.random.default_rng(n)
    lab = rng.integers(0, classes, n)
    if missing:
        lab[rng.choice(n, n // 10, replace=False)] = -1
    X = np.round(rng.standard_normal((n, 8)), 1).astype(np.float32)
    S = helpers.pairwise_similarity(X, metric="linear kernel", return_tensor=True)
    got = helpers.visualize_pairwise_similarity(lab, S)
    want = O.pair_stats(lab, S.cpu().numpy())
    assert got["n_related"] == want["n_related"] and got["n_unrelated"] == want["n_unrelated"]
    if want["n_related"] and want["n_unrelated"]:
        assert abs(got["auroc"] - want["auroc"]) < 1e-12             # integer counting: exact up to the final division
    for pop in ("related", "unrelated"):
        if want["n_" + pop]:
            for k in ("min", "q1", "median", "q3", "max"):
                assert abs(got[pop][k] - want[pop][k]) <= 1e-6 * (1 + abs(want[pop][k]))
            assert abs(got["mean_" + pop] - want["mean_" + pop]) <= 1e-6 * (1 + abs(want["mean_" + pop]))
        else:
            assert np.isnan(got["auroc"])
