#!/usr/bin/env python3
"""Generate tests/golden/reference_vectors.npz by EXECUTING THE REFERENCE'S OWN PYTHON FILES.

Runs only in the build container (needs /root/reference); the GPU box and the test-suite use the
committed .npz.  TensorFlow 1.12 cannot be installed here, so the reference modules
``autoencoder/triplet_loss_utils.py`` and ``autoencoder/utils.py`` are imported with a tiny eager
NumPy stand-in registered as ``tensorflow`` (``_TFShim`` below: the ~30 ops those two files call,
fp32-preserving, with TF 1.12's softplus thresholds).  The reference code itself -- masks, the B^3
broadcast, batch_hard's shifted min / masked max / float equality, weighted_loss, the legacy-RNG
noise and shuffling -- is what executes; only the op kernels underneath are NumPy instead of Eigen.

usage: python tests/golden/make_golden.py [--reference /root/reference]
"""
import argparse
import importlib.util
import os
import sys
import types

import numpy as np
from scipy import sparse


# --------------------------------------------------------------------------- #
# eager NumPy stand-in for the TF ops used by the two reference files
# --------------------------------------------------------------------------- #
def _softplus(x):
    x = np.asarray(x)
    thr = np.log(np.finfo(x.dtype).eps) + 2.0
    ex = np.exp(np.minimum(x, -thr))
    return np.where(x > -thr, x, np.where(x < thr, ex, np.log1p(ex))).astype(x.dtype)


def _axis(a):
    return tuple(a) if isinstance(a, (list, tuple)) else a


def make_tf_shim():
    tf = types.ModuleType("tensorflow")
    tf.bool = np.bool_
    tf.float32 = np.float32
    tf.cast = lambda x, dtype: np.asarray(x).astype(dtype)
    tf.eye = lambda n: np.eye(int(n), dtype=np.float32)
    tf.shape = lambda x: np.asarray(np.shape(x))
    tf.logical_not = np.logical_not
    tf.logical_and = np.logical_and
    tf.equal = lambda a, b: np.equal(a, b)
    tf.greater = lambda a, b: np.greater(a, np.asarray(b, dtype=np.asarray(a).dtype))
    tf.expand_dims = lambda x, axis: np.expand_dims(x, axis)
    tf.transpose = lambda x: np.transpose(x)
    tf.matmul = lambda a, b: np.matmul(a, b)
    tf.multiply = lambda a, b: np.multiply(a, b)
    tf.to_float = lambda x: np.asarray(x).astype(np.float32)
    tf.reduce_sum = lambda x, axis=None, keepdims=False: np.sum(x, axis=_axis(axis), keepdims=keepdims)
    tf.reduce_max = lambda x, axis=None, keepdims=False: np.max(x, axis=_axis(axis), keepdims=keepdims)
    tf.reduce_min = lambda x, axis=None, keepdims=False: np.min(x, axis=_axis(axis), keepdims=keepdims)
    tf.reduce_mean = lambda x, axis=None, keepdims=False: np.mean(x, axis=_axis(axis), keepdims=keepdims)
    tf.maximum = lambda a, b: np.maximum(a, np.asarray(b, dtype=np.asarray(a).dtype))
    tf.squeeze = lambda x: np.squeeze(x)
    tf.log = lambda x: np.log(x)
    tf.log_sigmoid = lambda x: -_softplus(-np.asarray(x))
    tf.squared_difference = lambda a, b: (a - b) * (a - b)
    tf.ones = lambda shape: np.ones(tuple(np.atleast_1d(shape).astype(int)), np.float32)
    nn = types.SimpleNamespace()
    nn.l2_normalize = lambda x, axis: x / np.sqrt(np.maximum(np.sum(x * x, axis=axis, keepdims=True),
                                                              np.asarray(1e-12, x.dtype)))
    tf.nn = nn
    tf.summary = types.SimpleNamespace(scalar=lambda *a, **k: None, histogram=lambda *a, **k: None)
    tf.sparse = types.SimpleNamespace(
        reduce_sum=lambda x, axis=None: np.asarray(x.sum(axis=axis)),
        to_dense=lambda x: np.asarray(x.toarray(), dtype=np.float32))
    tf.random_uniform = lambda shape, minval, maxval: np.random.uniform(minval, maxval, shape).astype(np.float32)
    return tf


def load_reference(ref_root):
    sys.modules["tensorflow"] = make_tf_shim()
    mods = {}
    for name in ("triplet_loss_utils", "utils"):
        path = os.path.join(ref_root, "autoencoder", name + ".py")
        spec = importlib.util.spec_from_file_location("_ref_" + name, path)
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        mods[name] = m
    return mods["triplet_loss_utils"], mods["utils"]


def golden_similar_articles(ref_root, out_path):
    """datasets/articles.py::similar_articles executed as shipped (pandas / NumPy are installed; ``jieba`` is only imported
    at module top for the tokenizer, so an empty stand-in module is registered for the import)."""
    import pandas as pd
    sys.modules.setdefault("jieba", types.ModuleType("jieba"))
    spec = importlib.util.spec_from_file_location("ref_articles", os.path.join(ref_root, "datasets", "articles.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = {}
    cases = [dict(n=60, classes=5, seed=1, min_cate=2, max_cate=None), dict(n=200, classes=40, seed=2, min_cate=2, max_cate=None),
             dict(n=120, classes=7, seed=3, min_cate=3, max_cate=25)]
    for k, c in enumerate(cases):
        rng = np.random.RandomState(100 + k)
        cat = rng.randint(0, c["classes"], c["n"])
        df = pd.DataFrame({"article_id": np.arange(1, c["n"] + 1), "label": cat})
        np.random.seed(c["seed"])
        res = mod.similar_articles(df.copy(), id_colname="article_id", cate_colname="label", min_cate=c["min_cate"], max_cate=c["max_cate"])
        out[f"c{k}_label"] = cat
        out[f"c{k}_cfg"] = np.array([c["n"], c["classes"], c["seed"], c["min_cate"], -1 if c["max_cate"] is None else c["max_cate"]])
        out[f"c{k}_pos"] = res["article_id_pos"].to_numpy().astype(np.int64)
        out[f"c{k}_neg"] = res["article_id_neg"].to_numpy().astype(np.int64)
        out[f"c{k}_valid"] = res["valid_triplet_data"].to_numpy().astype(np.int64)
    out["n_cases"] = np.array(len(cases))
    np.savez_compressed(out_path, **out)
    print("wrote", out_path, {k: v.shape for k, v in out.items() if k.endswith("_pos")})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.abspath(__file__)),
                                                   "reference_vectors.npz"))
    args = ap.parse_args()
    T, U = load_reference(args.reference)
    out = {}

    # ---- A. masks + miners (triplet_loss_utils.py:6-131, 202-259) ---- #
    case = 0
    for classes in (1, 3, 5):
        for signed in (0, 1):
            rng = np.random.default_rng(1000 + 10 * classes + signed)
            B, d = 24, 6
            lab = rng.integers(0, classes, size=B).astype(np.float32)
            h = (rng.random((B, d)) - 0.5 * signed).astype(np.float32)
            if signed:
                h *= np.float32(3.0)
            k = f"miner{case}_"
            out[k + "labels"] = lab; out[k + "encode"] = h
            out[k + "mask3"] = T._get_triplet_mask(lab)
            out[k + "mask_ap"] = T._get_anchor_positive_triplet_mask(lab)
            out[k + "mask_an"] = T._get_anchor_negative_triplet_mask(lab)
            for pos_only in (False, True):
                l, dw, fr, num = T.batch_all_triplet_loss(False, lab, h, pos_only)
                s = "pos" if pos_only else "all"
                out[k + f"ba_{s}_loss"] = np.float32(l); out[k + f"ba_{s}_dw"] = np.asarray(dw, np.float32)
                out[k + f"ba_{s}_frac"] = np.float32(fr); out[k + f"ba_{s}_num"] = np.float32(num)
            l, dw, fr, num = T.batch_hard_triplet_loss(False, lab, h)
            out[k + "bh_loss"] = np.float32(l); out[k + "bh_dw"] = np.asarray(dw, np.float32)
            out[k + "bh_frac"] = np.float32(fr); out[k + "bh_num"] = np.float32(num)
            case += 1
    out["n_miner_cases"] = np.int64(case)

    # ---- B. weighted_loss (triplet_loss_utils.py:262-277), dense and sparse feeds ---- #
    rng = np.random.default_rng(77)
    n, f = 16, 40
    xb = (rng.random((n, f)) < 0.25).astype(np.float32)
    xt = (xb * rng.random((n, f))).astype(np.float32)
    y = rng.random((n, f)).astype(np.float32)
    w = rng.integers(0, 50, n).astype(np.float32)
    out["wl_xb"] = xb; out["wl_xt"] = xt; out["wl_y"] = y; out["wl_w"] = w
    for lf in ("cross_entropy", "mean_squared", "cosine_proximity"):
        x = xb if lf == "cross_entropy" else xt
        out[f"wl_{lf}_unw"] = np.float32(T.weighted_loss(False, x, y, loss_func=lf))
        out[f"wl_{lf}_w"] = np.float32(T.weighted_loss(False, x, y, loss_func=lf, weight=w))
        out[f"wl_{lf}_sparse_w"] = np.float32(T.weighted_loss(True, sparse.csr_matrix(x), y, loss_func=lf, weight=w))

    # ---- C. host utils (utils.py:29-180): legacy-RNG noise, shuffling, CSR feed ---- #
    rng = np.random.default_rng(5)
    Xd = ((rng.random((30, 50)) < 0.2) * rng.random((30, 50))).astype(np.float32)
    Xs = sparse.csr_matrix(Xd)
    out["u_X"] = Xd
    np.random.seed(123)
    m = U.masking_noise(Xs, 0.3)
    out["u_mask_sparse_seed123"] = m.toarray()
    out["u_mask_sparse_sorted"] = np.bool_(m.has_sorted_indices)
    np.random.seed(123)
    out["u_mask_dense_seed123"] = np.asarray(U.masking_noise(Xd, 0.3), np.float32)
    np.random.seed(7)
    out["u_sp_sparse_seed7_v5"] = U.salt_and_pepper_noise(Xs, 5).toarray()
    np.random.seed(7)
    out["u_sp_dense_seed7_v5"] = np.asarray(U.salt_and_pepper_noise(Xd, 5), np.float32)
    out["u_decay_sparse"] = U.decay_noise(Xs, 0.3).toarray()
    out["u_decay_dense"] = np.asarray(U.decay_noise(Xd, 0.3))
    ind, val, shp = U.get_sparse_ind_val_shape(sparse.coo_matrix(Xd))
    out["u_feed_indices"] = np.asarray(ind, np.int64); out["u_feed_values"] = np.asarray(val)
    out["u_feed_shape"] = np.asarray(shp, np.int64)
    # one full "epoch" of host work in the reference's RNG order (autoencoder.py:218-220)
    ident = np.arange(30, dtype=np.float32).reshape(-1, 1)
    for bs, tag in ((4, "bs4"), (0.3, "bs0p3")):
        np.random.seed(42)
        xc = U.masking_noise(Xs, 0.3)
        order = []; sizes = []
        for b in U.gen_batches(sparse.csr_matrix(ident), sparse.csr_matrix(ident), bs,
                               data_label=np.arange(30)):
            order.extend(b[2].tolist()); sizes.append(b[0].shape[0])
        out[f"u_epoch_{tag}_xc"] = xc.toarray(); out[f"u_epoch_{tag}_order"] = np.asarray(order, np.int64)
        out[f"u_epoch_{tag}_sizes"] = np.asarray(sizes, np.int64)
    # gen_batches_triplet shares one shuffle across the three matrices (utils.py:87-91)
    np.random.seed(9)
    d3 = {k: ident.copy() for k in ("org", "pos", "neg")}
    order = []
    for a, b in U.gen_batches_triplet(d3, d3, 4):
        assert (a[0] == a[1]).all() and (a[0] == a[2]).all()
        order.extend(a[0][:, 0].astype(int).tolist())
    out["u_triplet_bs4_seed9_order"] = np.asarray(order, np.int64)

    np.savez_compressed(args.out, **out)
    golden_similar_articles(args.reference, os.path.join(os.path.dirname(os.path.abspath(args.out)), "similar_articles.npz"))
    print("wrote", args.out, len(out), "arrays", os.path.getsize(args.out), "bytes")


if __name__ == "__main__":
    main()
