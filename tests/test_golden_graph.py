"""Pin the oracle's encode / decode / cost / tied-weight gradients / optimizers / fit loop / transform / explicit-triplet
cost against vectors produced by EXECUTING THE REFERENCE'S OWN MODEL CLASSES (autoencoder/autoencoder.py,
autoencoder_triplet.py) over a torch-backed graph-mode ``tensorflow`` stand-in (tests/golden/make_golden_graph.py,
tests/golden/tf_graph_shim.py).  float64 on both sides: agreement is to rounding, not to a tolerance budget."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
from scipy import sparse

import oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
G = np.load(os.path.join(HERE, "golden", "reference_graph_vectors.npz"))
TOL = 1e-11


def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def cfg(tag):
    return json.loads(str(G[tag + "_cfg"]))


@pytest.mark.parametrize("ci", range(int(G["S_n"])))
def test_single_step_matches_reference_graph(ci):
    """h, y, cost (and its legs) and tf.gradients(cost, [W, bh, bv]) of the reference's graph (autoencoder.py:371-442)."""
    k = f"S{ci}_"; c = cfg(f"S{ci}")
    r = O.forward_backward(G[k + "W0"], G[k + "bh0"], G[k + "bv0"], G[k + "x"].astype(np.float64), G[k + "xc"].astype(np.float64),
                           G[k + "labels"], enc_act=c["enc"], dec_act=c["dec"], loss_func=c["loss"], triplet_strategy=c["strategy"],
                           alpha=c["alpha"], dt=np.float64)
    for mine, ref in (("h", "h"), ("y", "y"), ("cost", "cost"), ("dW", "dW"), ("dbh", "dbh"), ("dbv", "dbv")):
        assert rel(r[mine], G[k + ref]) < TOL, (c, mine)
    if c["strategy"] != "none":
        assert rel(r["ae_loss"], G[k + "ae"]) < TOL and rel(r["triplet_loss"], G[k + "triplet"]) < TOL
        assert float(r["num"]) == float(G[k + "num"]) and rel(r["fraction"], G[k + "fraction"]) < TOL
    # the reference evaluated in float32 (TF's width) stays within 1e-4 of the float64 truth on these cases
    assert abs(float(G[k + "cost_f32"]) - float(G[k + "cost"])) <= 1e-4 * abs(float(G[k + "cost"]))


@pytest.mark.parametrize("opt", ["gradient_descent", "ada_grad", "momentum", "adam"])
def test_optimizer_steps_match_reference_graph(opt):
    """three session.run(train_step) of tf.train.*Optimizer.minimize(cost) (autoencoder.py:444-477)."""
    W, bh, bv = [G["O_" + n].astype(np.float64).copy() for n in ("W0", "bh0", "bv0")]
    st = O.OptState(opt, [W.shape, bh.shape, bv.shape], np.float64)
    for t in range(3):
        r = O.forward_backward(W, bh, bv, G["O_x"].astype(np.float64), G["O_xc"].astype(np.float64), G["O_labels"],
                               triplet_strategy="batch_all", alpha=1.0, dt=np.float64)
        O.opt_apply(st, [W, bh, bv], [r["dW"], r["dbh"], r["dbv"]], 0.05, 0.6, np.float64)
        assert rel(r["cost"], G[f"O_{opt}_cost{t}"]) < TOL
        assert rel(W, G[f"O_{opt}_W{t}"]) < TOL and rel(bh, G[f"O_{opt}_bh{t}"]) < TOL and rel(bv, G[f"O_{opt}_bv{t}"]) < TOL


def fit_input(tag):
    c = cfg(tag)
    X = G[tag + "_X"].astype(np.float64)
    return c, (X if c["kind"] == "dense" else sparse.csr_matrix(X))


@pytest.mark.parametrize("tag", json.loads(str(G["F_tags"])))
def test_fit_matches_reference_fit(tag):
    """DenoisingAutoencoder.fit() as shipped (:126-246): per-batch cost / ae / triplet / fraction / num of every epoch, the
    final parameters (get_model_parameters) and transform()."""
    c, X = fit_input(tag)
    r = O.fit_reference(X, G[tag + "_labels"], G[tag + "_W0"], enc_act=c["enc"], dec_act=c["dec"], loss_func=c["loss"],
                        num_epochs=c["epochs"], batch_size=c["bs"], opt=c["opt"], learning_rate=c["lr"], momentum=0.5,
                        corr_type=c["corr"], corr_frac=c["frac"], seed=c["seed"], alpha=c["alpha"], triplet_strategy=c["strategy"],
                        dt=np.float64)
    assert rel([h["cost"] for h in r["history"]], G[tag + "_cost"]) < TOL
    if c["strategy"] != "none":
        for mine, ref in (("ae", "ae"), ("triplet", "triplet"), ("fraction", "fraction")):
            assert rel([h[mine] for h in r["history"]], G[tag + "_" + ref]) < TOL
        assert (np.array([h["num"] for h in r["history"]]) == G[tag + "_num"]).all()
    assert rel(r["W"], G[tag + "_W"]) < TOL and rel(r["bh"], G[tag + "_bh"]) < 1e-9 and rel(r["bv"], G[tag + "_bv"]) < TOL
    h, _ = O.encode(X, r["W"], r["bh"], c["enc"], np.float64)
    assert rel(h, G[tag + "_transform"]) < 1e-9
    assert rel(G[tag + "_cost_f32"], G[tag + "_cost"]) < 1e-5          # fp32 evaluation of the reference vs its fp64 truth


@pytest.mark.parametrize("tag", json.loads(str(G["T_tags"])))
def test_explicit_triplet_fit_matches_reference(tag):
    """DenoisingAutoencoderTriplet.fit() as shipped (autoencoder_triplet.py:40-146, cost :296-314)."""
    c = cfg(tag)
    data = [sparse.csr_matrix(G[tag + "_X_" + k].astype(np.float64)) for k in ("org", "pos", "neg")]
    np.random.seed(c["seed"])
    W = G[tag + "_W0"].copy(); bh = np.zeros(W.shape[1]); bv = np.zeros(W.shape[0])
    st = O.OptState("gradient_descent", [W.shape, bh.shape, bv.shape], np.float64)
    cost, ae, tri = [], [], []
    for _ in range(c["epochs"]):
        xc = [O.masking_noise(m, c["frac"]) if c["corr"] == "masking" else O.decay_noise(m, c["frac"]) for m in data]
        rows = ([], [], [])
        for idx in O.gen_batches_index(c["N"], c["bs"]):
            r = O.explicit_triplet_forward_backward(W, bh, bv, [m[idx] for m in data], [m[idx] for m in xc], enc_act=c["enc"],
                                                    dec_act=c["dec"], loss_func=c["loss"], alpha=c["alpha"], dt=np.float64)
            O.opt_apply(st, [W, bh, bv], [r["dW"], r["dbh"], r["dbv"]], c["lr"], 0.5, np.float64)
            rows[0].append(float(r["cost"])); rows[1].append(float(r["ae_loss"])); rows[2].append(float(r["triplet_loss"]))
        cost.append(rows[0]); ae.append(rows[1]); tri.append(rows[2])
    assert rel(cost, G[tag + "_cost"]) < TOL and rel(ae, G[tag + "_ae"]) < TOL and rel(tri, G[tag + "_triplet"]) < TOL
    assert rel(W, G[tag + "_W"]) < TOL and rel(bh, G[tag + "_bh"]) < 1e-9 and rel(bv, G[tag + "_bv"]) < TOL


def test_stdout_line_of_the_reference_fit():
    """the per-epoch line the reference prints (:283-294) -- our estimator prints the same format (tests/test_hip_cli.py)."""
    s = str(G["F1_stdout"])
    assert s.startswith("At step 1 (") and "[Train Stat (average over past steps)] - Triplet: Fraction=" in s
    assert "Cost: Overall=" in s and "Autoencoder=" in s and "Triplet=" in s


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="the reference checkout only exists in the build container")
def test_fixture_regenerates_bit_for_bit(tmp_path):
    out = tmp_path / "regen.npz"
    subprocess.check_call([sys.executable, os.path.join(HERE, "golden", "make_golden_graph.py"), "--out", str(out)],
                          stdout=subprocess.DEVNULL)
    R = np.load(out)
    assert sorted(R.files) == sorted(G.files)
    for k in G.files:
        if G[k].dtype.kind == "U":
            if not k.endswith("_stdout"):              # the stdout capture holds wall-clock seconds
                assert str(G[k]) == str(R[k]), k
        else:
            assert G[k].shape == R[k].shape and (G[k] == R[k]).all(), k
