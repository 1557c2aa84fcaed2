#!/usr/bin/env python3
"""Generate tests/golden/reference_graph_vectors.npz by EXECUTING THE REFERENCE'S MODEL CLASSES AS SHIPPED.

``/root/reference/autoencoder/autoencoder.py`` (DenoisingAutoencoder) and ``autoencoder_triplet.py``
(DenoisingAutoencoderTriplet) are imported UNCHANGED with ``tests/golden/tf_graph_shim.py`` registered as
``tensorflow`` (a lazy graph on torch-CPU autograd; TF 1.12 has no wheel for this Python and there is no
network).  What runs is the reference's own code:

  S  single steps:  ``_build_model`` (:322-338 -> _create_encode_layer :371, _create_decode_layer :395,
     _create_cost_function_node :417) for strategy x loss x activations x {sparse, dense} input; fetched:
     encode, decode, cost, autoencoder / triplet loss, fraction, num and ``tf.gradients(cost, [W, bh, bv])``
     -- the tied-weight gradient of THE REFERENCE'S graph (incl. the -act(bh) term).
  O  optimizers:    three consecutive ``session.run(train_step)`` for each of the four ``tf.train`` optimizers
     (:444-477); the update rules themselves are the shim's restatement of TF 1.12 (see the shim header).
  F  whole fits:    ``DenoisingAutoencoder.fit()`` (:126-246) as shipped -- per-epoch corruption, shuffling,
     CSR->COO feeds, session.run per batch -- with a subclass that only RECORDS the per-batch lists the
     reference averages at :283-294; then ``get_model_parameters()`` and ``transform()``.
  T  explicit triplets: ``DenoisingAutoencoderTriplet.fit()`` (autoencoder_triplet.py:40-146).

Every case is evaluated in float64 (truth; suffix-less keys) and the costs also in float32 (``*_f32``: TF's
arithmetic width).  Runs only in the build container (needs /root/reference); tests use the committed .npz.

usage: python tests/golden/make_golden_graph.py [--reference /root/reference] [--out ...]
"""
import argparse
import contextlib
import io
import json
import os
import sys
import tempfile

import numpy as np
import torch
from scipy import sparse

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import tf_graph_shim as shim  # noqa: E402


def load_reference(ref_root):
    if not hasattr(np, "int"):
        np.int = int                     # the reference targets numpy 1.15 (autoencoder.py:187 uses np.int)
    tf = shim.make_module()
    sys.modules["tensorflow"] = tf
    sys.path.insert(0, ref_root)
    import importlib
    RA = importlib.import_module("autoencoder.autoencoder")
    RT = importlib.import_module("autoencoder.autoencoder_triplet")
    assert RA.__file__.startswith(ref_root), RA.__file__
    return tf, RA, RT


def binary_csr(rng, n, f, density):
    m = (rng.random((n, f)) < density).astype(np.float64)
    m[np.arange(n), rng.integers(0, f, n)] = 1.0            # no empty rows
    return sparse.csr_matrix(m)


def valued_csr(rng, n, f, density):
    """CSR with values in (0,1]: no two corrupted rows can coincide.  batch_hard's data_weight compares floats for EQUALITY
    (triplet_loss_utils.py:251-253); with duplicate rows the outcome depends on the BLAS' last bit (numpy and torch disagree
    on D[i,j] == D[i,j'] for identical columns j, j'), so the batch_hard fits use valued rows to stay BLAS-independent."""
    m = binary_csr(rng, n, f, density)
    m.data = np.round(0.05 + 0.95 * rng.random(m.nnz), 3).astype(np.float32).astype(np.float64)   # fp32-exact values
    return m


def tfidf_dense(rng, n, f, density):
    m = (rng.random((n, f)) < density) * (0.1 + 0.9 * rng.random((n, f)))
    m[np.arange(n), rng.integers(0, f, n)] = 0.7
    m = m / np.sqrt((m * m).sum(1, keepdims=True))
    return np.ascontiguousarray(m.astype(np.float32).astype(np.float64))


ACT_PAIRS = [("sigmoid", "sigmoid"), ("tanh", "sigmoid"), ("tanh", "none"), ("none", "tanh"), ("sigmoid", "tanh")]


def single_steps(tf, RA, out):
    """Section S."""
    cases = []
    for strategy in ("none", "batch_all", "batch_hard"):
        for loss in ("cross_entropy", "mean_squared", "cosine_proximity"):
            for enc, dec in ACT_PAIRS:
                if loss == "cross_entropy" and dec != "sigmoid":
                    continue                                  # log of a non-probability: NaN in the reference too
                for kind in ("sparse", "dense"):
                    cases.append((strategy, loss, enc, dec, kind))
    B, F, H = 16, 30, 5
    for ci, (strategy, loss, enc, dec, kind) in enumerate(cases):
        rng = np.random.default_rng(4000 + ci)
        if kind == "sparse":
            x = binary_csr(rng, B, F, 0.2)
            keep = rng.random(x.nnz) >= 0.3
            xc = x.copy(); xc.data = xc.data * keep; xc.eliminate_zeros()
        else:
            x = tfidf_dense(rng, B, F, 0.3)
            xc = x * (rng.random(x.shape) >= 0.3)
        lab = rng.integers(0, 3, B).astype(np.float64)
        W0 = rng.uniform(-0.6, 0.6, (F, H)); bh0 = rng.uniform(-0.3, 0.3, H); bv0 = rng.uniform(-0.3, 0.3, F)
        k = f"S{ci}_"
        out[k + "cfg"] = np.array(json.dumps(dict(strategy=strategy, loss=loss, enc=enc, dec=dec, kind=kind, alpha=0.7)))
        out[k + "x"] = (x.toarray() if kind == "sparse" else x).astype(np.float32)
        out[k + "xc"] = (xc.toarray() if kind == "sparse" else xc).astype(np.float32)
        out[k + "labels"] = lab.astype(np.int32)
        out[k + "W0"] = W0; out[k + "bh0"] = bh0; out[k + "bv0"] = bv0
        for dt, sfx in ((torch.float64, ""), (torch.float32, "_f32")):
            shim.set_dtype(dt); shim.reset()
            shim.INJECT.update({"enc-w": W0, "hidden-bias": bh0, "visible-bias": bv0})
            m = RA.DenoisingAutoencoder(enc_act_func=enc, dec_act_func=dec, loss_func=loss, alpha=0.7, triplet_strategy=strategy,
                                        compress_factor=F // H)
            m.sparse_input = kind == "sparse"
            m.n_components = H
            m._build_model(F)
            with tf.Session() as s:
                s.run(tf.global_variables_initializer())
                conv = RA.utils.get_sparse_ind_val_shape if kind == "sparse" else (lambda a: a)
                feed = {m.input_data: conv(x), m.input_data_corr: conv(xc), m.input_label: lab}
                fetch = [m.encode, m.decode, m.cost] + tf.gradients(m.cost, [m.W_, m.bh_, m.bv_])
                if strategy != "none":
                    fetch += [m.autoencoder_loss, m.triplet_loss, m.fraction_triplet, m.num_triplet]
                r = s.run(fetch, feed_dict=feed)
            if sfx == "":
                out[k + "h"], out[k + "y"], out[k + "cost"], out[k + "dW"], out[k + "dbh"], out[k + "dbv"] = r[:6]
                if strategy != "none":
                    out[k + "ae"], out[k + "triplet"], out[k + "fraction"], out[k + "num"] = r[6:]
            else:
                out[k + "cost_f32"] = r[2]
    out["S_n"] = np.int64(len(cases))


def optimizer_steps(tf, RA, out):
    """Section O."""
    B, F, H = 16, 30, 5
    rng = np.random.default_rng(77)
    x = binary_csr(rng, B, F, 0.2)
    keep = rng.random(x.nnz) >= 0.3
    xc = x.copy(); xc.data = xc.data * keep; xc.eliminate_zeros()
    lab = rng.integers(0, 3, B).astype(np.float64)
    W0 = rng.uniform(-0.6, 0.6, (F, H)); bh0 = rng.uniform(-0.3, 0.3, H); bv0 = rng.uniform(-0.3, 0.3, F)
    out["O_x"] = x.toarray().astype(np.float32); out["O_xc"] = xc.toarray().astype(np.float32); out["O_labels"] = lab.astype(np.int32)
    out["O_W0"] = W0; out["O_bh0"] = bh0; out["O_bv0"] = bv0
    for opt in ("gradient_descent", "ada_grad", "momentum", "adam"):
        shim.set_dtype(torch.float64); shim.reset()
        shim.INJECT.update({"enc-w": W0, "hidden-bias": bh0, "visible-bias": bv0})
        m = RA.DenoisingAutoencoder(enc_act_func="sigmoid", dec_act_func="sigmoid", loss_func="cross_entropy", opt=opt,
                                    learning_rate=0.05, momentum=0.6, triplet_strategy="batch_all", compress_factor=F // H)
        m.sparse_input = True; m.n_components = H
        m._build_model(F)
        with tf.Session() as s:
            s.run(tf.global_variables_initializer())
            feed = {m.input_data: RA.utils.get_sparse_ind_val_shape(x), m.input_data_corr: RA.utils.get_sparse_ind_val_shape(xc),
                    m.input_label: lab}
            for t in range(3):
                _, c = s.run([m.train_step, m.cost], feed_dict=feed)
                out[f"O_{opt}_cost{t}"] = c
                out[f"O_{opt}_W{t}"], out[f"O_{opt}_bh{t}"], out[f"O_{opt}_bv{t}"] = [v.numpy() for v in shim.VARIABLES]


def _recording(cls):
    class Rec(cls):
        def _run_validation_error_and_summaries(self, epoch, *a):
            self.rec = getattr(self, "rec", [])
            self.rec.append(dict(cost=[float(v) for v in self.train_cost_batch[0]], ae=[float(v) for v in self.train_cost_batch[1]],
                                 triplet=[float(v) for v in self.train_cost_batch[2]],
                                 fraction=[float(v) for v in getattr(self, "fraction_triplet_batch", [])],
                                 num=[float(v) for v in getattr(self, "num_triplet_batch", [])]))
            return super()._run_validation_error_and_summaries(epoch, *a)
    return Rec


FIT_CASES = [
    dict(tag="F0", kind="sparse", strategy="none", loss="cross_entropy", enc="sigmoid", dec="sigmoid", corr="masking", frac=0.3,
         opt="gradient_descent", lr=0.1, bs=12, seed=3),
    dict(tag="F1", kind="sparse", strategy="batch_all", loss="cross_entropy", enc="sigmoid", dec="sigmoid", corr="masking", frac=0.3,
         opt="gradient_descent", lr=0.1, bs=12, seed=4),
    dict(tag="F2", kind="sparse_valued", strategy="batch_hard", loss="cross_entropy", enc="sigmoid", dec="sigmoid", corr="masking", frac=0.3,
         opt="gradient_descent", lr=0.1, bs=12, seed=5),
    dict(tag="F3", kind="sparse", strategy="batch_all", loss="mean_squared", enc="tanh", dec="none", corr="decay", frac=0.2,
         opt="momentum", lr=0.02, bs=12, seed=6),
    dict(tag="F4", kind="sparse", strategy="none", loss="cross_entropy", enc="sigmoid", dec="sigmoid", corr="salt_and_pepper", frac=0.1,
         opt="ada_grad", lr=0.1, bs=12, seed=7),
    dict(tag="F5", kind="dense", strategy="batch_all", loss="cosine_proximity", enc="tanh", dec="sigmoid", corr="masking", frac=0.3,
         opt="gradient_descent", lr=0.1, bs=0.25, seed=8),
    dict(tag="F6", kind="sparse_valued", strategy="batch_hard", loss="cross_entropy", enc="sigmoid", dec="sigmoid", corr="masking", frac=0.3,
         opt="adam", lr=0.01, bs=12, seed=9),
]


def whole_fits(tf, RA, out):
    """Section F."""
    Rec = _recording(RA.DenoisingAutoencoder)
    N, F, cf, epochs = 60, 40, 8, 3
    H = F // cf
    for c in FIT_CASES:
        rng = np.random.default_rng(900 + c["seed"])
        X = {"sparse": lambda: binary_csr(rng, N, F, 0.15), "sparse_valued": lambda: valued_csr(rng, N, F, 0.15),
             "dense": lambda: tfidf_dense(rng, N, F, 0.3)}[c["kind"]]()
        lab = rng.integers(0, 4, N)
        W0 = rng.uniform(-0.4, 0.4, (F, H))
        k = c["tag"] + "_"
        out[k + "cfg"] = np.array(json.dumps(dict(c, N=N, F=F, compress_factor=cf, epochs=epochs, alpha=1.5)))
        out[k + "X"] = (X if c["kind"] == "dense" else X.toarray()).astype(np.float32)
        out[k + "labels"] = lab.astype(np.int32); out[k + "W0"] = W0
        for dt, sfx in ((torch.float64, ""), (torch.float32, "_f32")):
            shim.set_dtype(dt); shim.reset()
            shim.INJECT.update({"enc-w": W0})
            m = Rec(model_name=c["tag"], main_dir=c["tag"], compress_factor=cf, enc_act_func=c["enc"], dec_act_func=c["dec"],
                    loss_func=c["loss"], num_epochs=epochs, batch_size=c["bs"], opt=c["opt"], learning_rate=c["lr"], momentum=0.5,
                    corr_type=c["corr"], corr_frac=c["frac"], verbose=1, verbose_step=1, seed=c["seed"], alpha=1.5,
                    triplet_strategy=c["strategy"])
            with contextlib.redirect_stdout(io.StringIO()) as so:
                m.fit(X, train_set_label=lab)                  # labels are always passed: autoencoder.py:227-230 needs them
                params = m.get_model_parameters()
                enc = m.transform(X)
            for name in ("cost", "ae", "triplet", "fraction", "num"):
                a = np.array([r[name] for r in m.rec], np.float64)
                if a.size:
                    out[k + name + sfx] = a
            if sfx == "":
                out[k + "W"], out[k + "bh"], out[k + "bv"] = params["enc_w"], params["enc_b"], params["dec_b"]
                out[k + "transform"] = enc
                out[k + "stdout"] = np.array(so.getvalue())
    out["F_tags"] = np.array(json.dumps([c["tag"] for c in FIT_CASES]))


TRIPLET_CASES = [
    dict(tag="T0", loss="cross_entropy", enc="sigmoid", dec="sigmoid", corr="masking", frac=0.3, lr=0.1, seed=11),
    dict(tag="T1", loss="cosine_proximity", enc="tanh", dec="none", corr="decay", frac=0.2, lr=0.05, seed=12),
]


def triplet_fits(tf, RT, out):
    """Section T."""
    Rec = _recording(RT.DenoisingAutoencoderTriplet)
    N, F, cf, epochs = 40, 36, 6, 2
    H = F // cf
    for c in TRIPLET_CASES:
        rng = np.random.default_rng(1300 + c["seed"])
        data = {key: binary_csr(rng, N, F, 0.18) for key in ("org", "pos", "neg")}
        W0 = rng.uniform(-0.4, 0.4, (F, H))
        k = c["tag"] + "_"
        out[k + "cfg"] = np.array(json.dumps(dict(c, N=N, F=F, compress_factor=cf, epochs=epochs, alpha=2.0, bs=10)))
        for key in data:
            out[k + "X_" + key] = data[key].toarray().astype(np.float32)
        out[k + "W0"] = W0
        for dt, sfx in ((torch.float64, ""), (torch.float32, "_f32")):
            shim.set_dtype(dt); shim.reset()
            shim.INJECT.update({"enc-w": W0})
            m = Rec(model_name=c["tag"], main_dir=c["tag"], compress_factor=cf, enc_act_func=c["enc"], dec_act_func=c["dec"],
                    loss_func=c["loss"], num_epochs=epochs, batch_size=10, opt="gradient_descent", learning_rate=c["lr"],
                    corr_type=c["corr"], corr_frac=c["frac"], verbose=1, verbose_step=1, seed=c["seed"], alpha=2.0)
            m.train_summary = None          # the shipped class reads an attribute it never sets (autoencoder_triplet.py:146)
            with contextlib.redirect_stdout(io.StringIO()):
                m.fit(data)
                params = m.get_model_parameters()
            for name in ("cost", "ae", "triplet"):
                out[k + name + sfx] = np.array([r[name] for r in m.rec], np.float64)
            if sfx == "":
                out[k + "W"], out[k + "bh"], out[k + "bv"] = params["enc_w"], params["enc_b"], params["dec_b"]
    out["T_tags"] = np.array(json.dumps([c["tag"] for c in TRIPLET_CASES]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(HERE, "reference_graph_vectors.npz"))
    a = ap.parse_args()
    out_path = os.path.abspath(a.out)
    tf, RA, RT = load_reference(os.path.abspath(a.reference))
    out = {}
    with tempfile.TemporaryDirectory() as tmp:      # the reference writes results/<algo>/<main_dir>/... relative to the cwd
        cwd = os.getcwd()
        os.chdir(tmp)
        try:
            single_steps(tf, RA, out)
            optimizer_steps(tf, RA, out)
            whole_fits(tf, RA, out)
            triplet_fits(tf, RT, out)
        finally:
            os.chdir(cwd)
    np.savez_compressed(out_path, **out)
    print("wrote", out_path, len(out), "arrays", os.path.getsize(out_path), "bytes")


if __name__ == "__main__":
    main()
