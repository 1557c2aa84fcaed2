"""Host-side utilities with the names and argument meaning of the reference's ``autoencoder/utils.py``.

These are the pieces of the training path that are *host logic in the reference too*: the NumPy
legacy-RNG noise functions and the shuffled mini-batch generators (``utils.py:29-180``).  They are kept
so code written against the reference imports and runs unchanged, and so the reference-exact RNG stream
(``rng='numpy'`` in ``DenoisingAutoencoder``) can be produced bit-for-bit.  The per-batch *compute*
(corrupt + gather + matmul + losses) does not go through them: ``fit()`` keeps the train set resident in
HBM and only ships keep-bits / a permutation per epoch (see ``epoch_keep_bits`` / ``epoch_permutation``).
"""
from __future__ import annotations

import numpy as np
from scipy import sparse

try:  # pandas is optional for the hot path; gen_batches accepts DataFrames like the reference
    import pandas as pd
except Exception:  # pragma: no cover
    pd = None


_XAVIER_RNG = np.random.RandomState()      # private stream: the weight draw must not consume the global legacy stream


def xavier_init(fan_in, fan_out, const=1, rng=None):
    """U(-c*sqrt(6/(fan_in+fan_out)), +c*sqrt(...)) as a float32 ndarray  (reference utils.py:16-26).

    The reference draws from ``tf.random_uniform`` under the TF graph seed: a stream of its own that cannot be
    reproduced without TensorFlow and that does NOT touch NumPy's global RandomState.  The draw here therefore comes
    from a private ``RandomState`` (``rng``, or a module-level one), so that with the same ``seed`` the global legacy
    stream -- masking decisions and shuffles of every epoch -- stays exactly the reference's (autoencoder.py:72-73,
    218-220) whether or not ``init_weights`` is injected."""
    bound = const * np.sqrt(6.0 / (fan_in + fan_out))
    draw = (_XAVIER_RNG if rng is None else rng).uniform
    return draw(-bound, bound, (fan_in, fan_out)).astype(np.float32)


def _resolve_batch_size(n_rows, batch_size):
    assert batch_size > 0.
    if batch_size < 1.:
        batch_size = max(round(n_rows * batch_size), 1)          # fraction of the set (utils.py:47)
    return int(batch_size)


def _is_frame(x):
    return pd is not None and isinstance(x, (pd.DataFrame, pd.Series))


def _take(container, rows):
    return container.iloc[rows] if _is_frame(container) else container[rows]


def epoch_permutation(n_rows, random=True):
    """The row order ``gen_batches`` uses for one epoch: ``np.random.shuffle(list(range(N)))``
    (utils.py:50-51).  Shuffling an int64 ndarray consumes the legacy stream identically to shuffling
    the Python list (same Fisher-Yates draws) and is ~50x faster; tests pin the equality."""
    index = np.arange(n_rows, dtype=np.int64)
    if random:
        np.random.shuffle(index)
    return index


def gen_batches(data, data_corrupted, batch_size, data_label=None, random=True):
    """Yield ``(batch, corrupted_batch[, label_batch])`` slices in shuffled order (utils.py:29-70).

    ``data`` / ``data_corrupted``: ndarray, scipy sparse matrix or DataFrame of the same type;
    ``batch_size``: int >= 1 or a fraction in (0,1); ``data_label``: 1-d / [N,1] array, Series or
    DataFrame."""
    assert data.shape[0] == data_corrupted.shape[0]
    assert type(data) == type(data_corrupted), (type(data), type(data_corrupted))
    if _is_frame(data):
        assert (data.index == data_corrupted.index).all()
    if data_label is not None:
        assert data_label.ndim == 1 or data_label.shape[1] == 1
    n = data.shape[0]
    bs = _resolve_batch_size(n, batch_size)
    order = epoch_permutation(n, random).tolist()
    for start in range(0, n, bs):
        rows = order[start:start + bs]
        if data_label is None:
            yield _take(data, rows), _take(data_corrupted, rows)
        else:
            yield _take(data, rows), _take(data_corrupted, rows), _take(data_label, rows)


def gen_batches_triplet(data, data_corrupted, batch_size, random=True):
    """Shared-shuffle batches over a dict of same-shape matrices, e.g. {'org','pos','neg'}
    (utils.py:73-91).  Unlike the reference (which forgets the ``int()`` and crashes for float sizes
    >= 1) the batch size is always an integer."""
    key = None
    for key in data:
        assert data[key].shape[0] == data_corrupted[key].shape[0]
    n = data[key].shape[0]
    bs = _resolve_batch_size(n, batch_size)
    order = epoch_permutation(n, random).tolist()
    for start in range(0, n, bs):
        rows = order[start:start + bs]
        yield [data[k][rows, :] for k in data], [data_corrupted[k][rows, :] for k in data]


def masking_keep(n_stored, v):
    """Keep decisions of sparse masking noise: ``np.random.rand(nnz) >= v`` in storage order
    (utils.py:111).  This bool vector IS the epoch's corruption for the HBM-resident CSR."""
    assert 0. <= v <= 1.
    return np.random.rand(n_stored) >= v


def masking_noise(X, v):
    """Force a fraction ``v`` of the entries of X to zero (utils.py:94-115).
    sparse input: every *stored* entry is dropped with probability v; dense input: every element."""
    assert 0. <= v <= 1.
    if isinstance(X, np.ndarray):
        keep = np.random.choice(a=[0, 1], size=X.shape, p=[v, 1 - v])
        return keep * X
    coo = X.tocoo(True)
    keep = masking_keep(coo.nnz, v)
    out = sparse.coo_matrix((coo.data[keep], (coo.row[keep], coo.col[keep])), shape=coo.shape)
    return out.tocsr()


def salt_and_pepper_noise(X, v):
    """Per row, ``v`` random column ids (with replacement) are set to the global min or max of X by a
    fair coin (utils.py:118-144).  Draw order per row: ``randint(0,F,v)`` then v uniform draws; when an
    id repeats the last coin wins -- resolved here without the reference's per-element Python loop."""
    dense = isinstance(X, np.ndarray)
    n_rows, n_features = X.shape
    lo, hi = X.min(), X.max()
    out = X.copy() if dense else X.tolil(True)
    for i in range(n_rows):
        cols = np.random.randint(0, n_features, v)
        coins = np.random.random(len(cols))
        if len(cols) == 0:
            continue
        # last occurrence of every column id decides its value
        rev_cols = cols[::-1]
        uniq, first_in_rev = np.unique(rev_cols, return_index=True)
        vals = np.where(coins[::-1][first_in_rev] < 0.5, lo, hi)
        if dense:
            out[i, uniq] = vals
        else:
            for c, val in zip(uniq.tolist(), vals.tolist()):
                out[i, c] = val
    return out if dense else out.tocsr()


def decay_noise(X, v):
    """All elements decayed by the fraction v: X * (1 - v)   (utils.py:147-159)."""
    return X.copy() * (1. - v)


def get_sparse_ind_val_shape(sparse_m):
    """(indices [nnz,2], values [nnz], shape) of a scipy sparse matrix in row-major sorted order
    (utils.py:162-180) -- the layout ``tf.sparse.placeholder`` was fed with.  The device path does not
    use it (the CSR stays resident in HBM); kept for API compatibility."""
    m = sparse_m if isinstance(sparse_m, sparse.csr_matrix) else sparse.csr_matrix(sparse_m)
    m.sort_indices()
    coo = m.tocoo()
    return np.column_stack((coo.row, coo.col)), coo.data, coo.shape


def masking_keep_bits(n_draws, v):
    """Packed keep decisions of one epoch's masking noise, drawn from NumPy's legacy GLOBAL stream by the native generator
    (csrc/dae_host_rng.cpp): bit e of the returned uint32 words == ``(np.random.rand(n_draws) >= v)[e]`` and the global
    RandomState ends up exactly where that call would leave it (tests/test_host_rng.py), ~4x faster than rand + packbits.
    For the dense path pass ``v = dense_masking_threshold(v)``."""
    import ctypes as C
    from .. import _lib as L
    assert 0. <= v <= 1.
    lib = L.load()
    st = np.random.get_state()
    assert st[0] == 'MT19937'
    key = np.ascontiguousarray(st[1], dtype=np.uint32).copy()
    pos = C.c_int32(int(st[2]))
    bits = np.empty((int(n_draws) + 31) // 32, np.uint32)
    rc = lib.dae_host_mt19937_keep_bits(key.ctypes.data, C.addressof(pos), int(n_draws), float(v), bits.ctypes.data)
    if rc != 0:
        raise RuntimeError("dae_host_mt19937_keep_bits failed (rc=%d)" % rc)
    np.random.set_state((st[0], key, pos.value, st[3], st[4]))
    return bits


def dense_masking_threshold(v):
    """``np.random.choice([0, 1], size, p=[v, 1-v])`` (dense masking, utils.py:108) keeps an element iff its uniform draw is
    >= cdf[0], with cdf = cumsum(p) / cumsum(p)[-1] as the legacy ``choice`` computes it."""
    cdf = np.cumsum(np.array([v, 1 - v], dtype=np.float64))
    cdf /= cdf[-1]
    return float(cdf[0])


def pack_keep_bits(keep):
    """bool[nnz] -> little-endian uint32 bit words (bit e = keep decision of stored entry e)."""
    bits = np.packbits(np.asarray(keep, dtype=bool), bitorder="little")
    pad = (-len(bits)) % 4
    if pad:
        bits = np.concatenate([bits, np.zeros(pad, np.uint8)])
    return bits.view(np.uint32)


def class_sort_batches(order, labels, batch):
    """Reorder an epoch's row permutation so that every mini-batch (consecutive `batch` rows) is sorted by label (stable).
    A mini-batch is a SET for the reference's objective -- every per-batch quantity it computes (weighted reconstruction loss,
    batch_all / batch_hard mining, their gradients) is a sum or mean over the batch -- so the batches keep exactly the rows the
    reference's shuffle gave them (utils.py:57-60) and only the position inside a batch changes.  The batch_all miner then finds
    an anchor's positives and negatives as index ranges instead of compacting them.  Returns the new permutation."""
    order = np.asarray(order)
    lab = np.asarray(labels)[order].astype(np.int64)
    key = (np.arange(order.size, dtype=np.int64) // int(batch)) * (int(lab.max()) + 1 if lab.size else 1) + lab
    return order[np.argsort(key, kind='stable')]
