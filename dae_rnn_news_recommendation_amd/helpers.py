"""Mirror of the parts of the reference's ``helpers.py`` either side of the training path.

* ``save_file`` / ``read_file`` (helpers.py:138-264): the on-disk artefact conventions -- format picked from the file
  extension, writer / reader picked from the container type (ndarray: csv, tsv, npy; scipy.sparse: npz, and csv / tsv
  after densifying; DataFrame: csv, tsv, parquet, pkl; Series: csv, tsv, pkl).  Host code.
* ``pairwise_similarity`` (helpers.py:11-50) -- the evaluation step
``main_autoencoder.py:307-317`` runs six times right after training (embeddings, binary BoW, TF-IDF; train and
validation).  Same name, arguments, assert and return value; the normalisation, the N x N product (exact-fp32 MFMA) and
the diagonal fill run on the MI355X through ``dae_pairwise_similarity``.  No CPU implementation: without the built
library / a GPU this raises."""
from __future__ import annotations

import numpy as np

from . import _lib as L

def _container_kind(data):
    import scipy.sparse as sp
    try:
        import pandas as pd
    except ImportError:                                                # pragma: no cover
        pd = None
    if isinstance(data, np.ndarray):
        return "numpy"
    if sp.issparse(data):
        return "scipy"
    if pd is not None and isinstance(data, pd.DataFrame):
        return "pandas_df"
    if pd is not None and isinstance(data, pd.Series):
        return "pandas_series"
    return None


_WRITABLE = {"numpy": ("csv", "tsv", "npy"), "scipy": ("npz",), "pandas_df": ("csv", "tsv", "parquet", "pkl"),
             "pandas_series": ("csv", "tsv", "pkl")}
_READABLE = {"numpy": ("csv", "tsv", "npy"), "scipy": ("csv", "tsv", "npz"), "pandas_df": ("csv", "tsv", "parquet", "pkl"),
             "pandas_series": ("csv", "tsv", "pkl")}


def save_file(data, path, format=None, **savekwargs):
    """Write ``data`` to ``path``; the format is the lower-cased extension unless given (helpers.py:138-199)."""
    import scipy.sparse as sp
    path = str(path)
    if format is None:
        format = path.lower().split(".")[-1]
    if sp.issparse(data) and format in ("csv", "tsv"):                 # text formats of a sparse matrix: densify first (:146-147)
        data = data.toarray()
    kind = _container_kind(data)
    assert kind is not None and format in _WRITABLE[kind], \
        "Shoule be one of following format {}".format(list(_WRITABLE.get(kind, ())))      # the reference's message (:198)
    sep = "," if format == "csv" else "\t"
    if kind == "numpy":
        if format == "npy":
            np.save(path, data, **savekwargs)
        else:
            np.savetxt(path, data, delimiter=sep, **savekwargs)
    elif kind == "scipy":
        sp.save_npz(path, data, **savekwargs)
    elif format in ("csv", "tsv"):
        data.to_csv(path, sep=sep, **savekwargs)
    elif format == "parquet":
        data.to_parquet(path, **savekwargs)
    else:
        data.to_pickle(path, **savekwargs)


def read_file(path, data_type=None, format=None, **readkwargs):
    """Read what ``save_file`` wrote.  ``data_type`` defaults from the format: npy -> 'numpy', npz -> 'scipy', everything
    else -> 'pandas_df' (helpers.py:202-264); pass 'pandas_series' for pickled / csv label vectors as the reference's
    scripts do (main_autoencoder.py:167-170)."""
    import os
    import scipy.sparse as sp
    path = str(path)
    assert os.path.isfile(path), "[Error] {} is not a file".format(path)
    if format is None:
        format = path.lower().split(".")[-1]
    if data_type is None:
        data_type = {"npy": "numpy", "npz": "scipy"}.get(format, "pandas_df")
    assert data_type in _READABLE
    assert format in _READABLE[data_type]
    sep = "," if format == "csv" else "\t"
    if data_type == "numpy":
        return np.load(path, **readkwargs) if format == "npy" else np.loadtxt(path, delimiter=sep, **readkwargs)
    if data_type == "scipy":
        return sp.load_npz(path, **readkwargs) if format == "npz" else sp.csr_matrix(np.loadtxt(path, delimiter=sep, **readkwargs))
    import pandas as pd
    if format == "parquet":
        return pd.read_parquet(path, **readkwargs)
    if format == "pkl":
        return pd.read_pickle(path, **readkwargs)
    if data_type == "pandas_df":
        return pd.read_csv(path, sep=sep, index_col=0, parse_dates=True, **readkwargs)
    # a Series written by to_csv: no header row, first column is the index (the reference asks read_csv for squeeze=True,
    # which current pandas spells .squeeze("columns"))
    return pd.read_csv(path, sep=sep, index_col=0, parse_dates=True, header=None, **readkwargs).squeeze("columns")


_NORMS = {"": 0, "l1": 1, "l2": 2, "max": 3}
_METRICS = {"cosine": 0, "linear kernel": 1}


def _csr_to_dense(torch, m, dev):
    return torch.sparse_csr_tensor(torch.from_numpy(m.indptr.astype(np.int64)), torch.from_numpy(m.indices.astype(np.int64)),
                                   torch.from_numpy(m.data), size=m.shape).to(dev).to_dense()


def pairwise_similarity(in_df, norm="", metric="cosine", set_diagonal_zero=True, *, return_tensor=False, device=None):
    """Pairwise similarity of the rows of ``in_df`` (ndarray, scipy.sparse matrix, list, or a CUDA float32 tensor).

    norm: '' or sklearn.preprocessing.normalize's 'l1' / 'l2' / 'max', applied first; metric: 'cosine' or
    'linear kernel'; set_diagonal_zero as in the reference.  Returns a float32 ndarray [N x N]
    (``return_tensor=True``: the CUDA tensor view, no copy to the host).

    Sparse input is densified on the device (one fp32 image of the matrix); the reference's sklearn path would return
    a sparse matrix for ``linear kernel`` + sparse input with dense_output left at its default -- here the result is
    always dense, which is what every caller in the reference needs (they index and plot it)."""
    import torch
    assert metric in ["cosine", "linear kernel"]                      # helpers.py:34
    if norm not in _NORMS:
        raise ValueError(f"'{norm}' is not a supported norm")         # sklearn.preprocessing.normalize's message
    lib = L.load()
    dev = torch.device("cuda" if device is None else device)
    if isinstance(in_df, torch.Tensor):
        X = in_df.to(device=dev, dtype=torch.float32)
    else:
        try:
            import scipy.sparse as sp
            is_sparse = sp.issparse(in_df)
        except ImportError:                                            # pragma: no cover
            is_sparse = False
        if is_sparse:
            import warnings
            m = in_df.tocsr().astype(np.float32)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")                        # torch: "sparse CSR tensor support is in beta state"
                X = _csr_to_dense(torch, m, dev)
        else:
            X = torch.as_tensor(np.asarray(in_df, dtype=np.float32)).to(dev)
    if X.dim() != 2:
        raise ValueError("Expected 2D array")
    X = X.contiguous()
    N, D = int(X.shape[0]), int(X.shape[1])
    Np = L.pad(N)
    out = torch.empty((Np, Np), dtype=torch.float32, device=dev)
    ws_bytes = int(lib.dae_pairwise_similarity_workspace(N, D))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        L.call("dae_pairwise_similarity", L.ptr(X), X.stride(0), N, D, _NORMS[norm], _METRICS[metric], 1 if set_diagonal_zero else 0,
               L.ptr(out), Np, L.ptr(ws), ws_bytes, L.current_stream())
    res = out[:N, :N]
    if return_tensor:
        return res
    return res.cpu().numpy()


_STAT_KEYS = ("auroc", "n_related", "n_unrelated", "mean_related", "mean_unrelated")


def visualize_pairwise_similarity(labels, pairwise_similarity_metrics, plot='boxplot', title=None, figsize=(16, 9), save_path=None,
                                  **plot_kwargs):
    """The numbers behind the reference's ROC + box plot figure (helpers.py:79-135), computed on the device.

    Same arguments and asserts.  Pairs (i, j), j < i, with both labels >= 0 are 'related' when the labels are equal,
    'unrelated' otherwise; returns a dict with the AUROC of related-vs-unrelated scores (what ``roc_curve`` + ``auc`` give
    the reference), the population sizes, means and the box-plot five-number summaries.  Nothing is drawn (matplotlib is
    not part of this stack): with ``save_path`` the dict is written as JSON next to where the figure would have gone
    (``.png`` -> ``.json``).  ``pairwise_similarity_metrics`` may be an ndarray or a CUDA tensor (e.g. from
    ``pairwise_similarity(..., return_tensor=True)``)."""
    import ctypes
    import json
    import torch
    labels = np.asarray(labels)
    assert labels.shape[0] == pairwise_similarity_metrics.shape[0]
    assert pairwise_similarity_metrics.shape[0] == pairwise_similarity_metrics.shape[1]
    assert plot in ['scatter', 'boxplot']
    lib = L.load()
    S = pairwise_similarity_metrics
    if not isinstance(S, torch.Tensor):
        S = torch.as_tensor(np.asarray(S, dtype=np.float32))
    S = S.to(device="cuda" if not S.is_cuda else S.device, dtype=torch.float32)
    if S.stride(1) != 1:
        S = S.contiguous()
    N = int(S.shape[0])
    lab = np.asarray(labels).reshape(N, -1)[:, 0]
    if lab.dtype.kind == 'f':                      # NaN / inf 'missing' labels: the reference filters them with labels >= 0
        lab = np.where(np.isfinite(lab), lab, -1.0)
    lab = np.ascontiguousarray(lab.astype(np.int32))
    ws_bytes = int(lib.dae_pair_stats_workspace(N))
    ws = torch.empty(ws_bytes + 256, dtype=torch.uint8, device=S.device)
    off = (-ws.data_ptr()) % 256
    out = (ctypes.c_double * 16)()
    with torch.cuda.device(S.device):
        L.call("dae_pair_stats", L.ptr(S), S.stride(0), lab.ctypes.data_as(ctypes.c_void_p), N, ctypes.cast(out, ctypes.c_void_p),
               ctypes.c_void_p(ws.data_ptr() + off), ws_bytes, L.current_stream())
    v = [float(x) for x in out]
    res = dict(zip(_STAT_KEYS, v[:5]))
    res["n_related"], res["n_unrelated"] = int(v[1]), int(v[2])
    res["related"] = dict(zip(("min", "q1", "median", "q3", "max"), v[5:10]))
    res["unrelated"] = dict(zip(("min", "q1", "median", "q3", "max"), v[10:15]))
    res["title"] = title
    if save_path is not None:
        path = str(save_path)
        path = path[:-4] + ".json" if path.lower().endswith(".png") else path + ".json"
        with open(path, "w") as fh:
            json.dump(res, fh, indent=1)
    return res
