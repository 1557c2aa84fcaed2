"""Seeded synthetic inputs of the shapes BASELINE.json names (the UCI-news blob is missing from the
reference checkout: .MISSING_LARGE_BLOBS:2).  Recipe from SURVEY.md section 8(d): row nnz ~ clipped
Poisson(lambda), columns Zipf(s~1.1) without replacement per row, values 1.0 (binary,
main_autoencoder.py:235) or L2-normalised tf-idf-like weights; 4-class labels with the UCI category
skew (b/t/e/m ~ .27/.26/.36/.11) or power-law "story" ids."""
from __future__ import annotations

import numpy as np
from scipy import sparse


def synthetic_csr(n_rows, n_features, *, nnz_per_row=200, seed=1234, tfidf=False, zipf_s=1.1):
    rng = np.random.default_rng(seed)
    p = 1.0 / np.arange(1, n_features + 1) ** zipf_s
    p /= p.sum()
    counts = np.clip(rng.poisson(nnz_per_row, n_rows), 1, min(n_features, 4 * nnz_per_row))
    # Zipf columns without replacement per row: Gumbel top-k on log p (vectorised in row blocks)
    logp = np.log(p)
    indptr = np.zeros(n_rows + 1, np.int64)
    indptr[1:] = np.cumsum(counts)
    indices = np.empty(indptr[-1], np.int32)
    blk = max(1, (1 << 24) // n_features)
    for r0 in range(0, n_rows, blk):
        r1 = min(n_rows, r0 + blk)
        g = logp[None, :] + rng.gumbel(size=(r1 - r0, n_features))
        kmax = int(counts[r0:r1].max())
        top = np.argpartition(-g, kmax - 1, axis=1)[:, :kmax]
        # order the kmax candidates by score so that the first counts[i] are the true top-k
        order = np.argsort(-np.take_along_axis(g, top, axis=1), axis=1)
        top = np.take_along_axis(top, order, axis=1)
        for i in range(r0, r1):
            c = np.sort(top[i - r0, :counts[i]])
            indices[indptr[i]:indptr[i + 1]] = c
    if tfidf:
        data = rng.random(indptr[-1]).astype(np.float32) * 0.9 + 0.1
        m = sparse.csr_matrix((data, indices, indptr), shape=(n_rows, n_features))
        norms = np.sqrt(np.asarray(m.multiply(m).sum(axis=1))).ravel()
        m = sparse.diags(1.0 / np.maximum(norms, 1e-12)).dot(m).tocsr().astype(np.float32)
        m.sort_indices()
        return m
    data = np.ones(indptr[-1], np.float32)
    return sparse.csr_matrix((data, indices, indptr), shape=(n_rows, n_features))


def synthetic_labels(n_rows, *, kind="category", seed=1234):
    rng = np.random.default_rng(seed + 1)
    if kind == "category":
        return rng.choice(4, size=n_rows, p=[0.27, 0.26, 0.36, 0.11]).astype(np.int64)
    n_story = max(2, n_rows // 6)
    w = 1.0 / np.arange(1, n_story + 1) ** 0.8
    return rng.choice(n_story, size=n_rows, p=w / w.sum()).astype(np.int64)


def xavier_uniform(n_features, n_components, const=1, seed=42):
    """U(+-const*sqrt(6/(F+H))) (autoencoder/utils.py:16-26) from a NumPy Generator: the reference's
    tf.random_uniform stream is not reproducible without TensorFlow, so parity runs inject this W0."""
    b = const * np.sqrt(6.0 / (n_features + n_components))
    return np.random.default_rng(seed).uniform(-b, b, (n_features, n_components)).astype(np.float32)
