"""Full-shape 20-step loss curves of BASELINE.json configs 1, 3, 4 and 5 (configs[1] = c2: tests/test_hip_full_curve.py) against the FLOAT32
oracle's per-batch losses frozen by tests/golden/make_curves.py -- the north star's gate is a CURVE ("loss curve matching reference within 1e-4"),
so the product's default precision ('auto') is held to 1e-4 on every step of every named config, through the drop-in estimators' own fit():
same regenerated inputs, reference-exact legacy RNG stream, injected W0.

  c1  strategy none                     (reference autoencoder.py:283-294: the per-batch values the epoch line averages)
  c3  batch_hard + 4 category labels    (triplet_loss_utils.py:202-259).  The reference's own float32 arithmetic does NOT determine this curve to 1e-4:
                                         the fp32 oracle re-run with ONE of its 5 million initial weights moved by one ulp (6e-8) leaves its own frozen
                                         curve by 5.3e-5 (cost) / 1.45e-4 (triplet) once the decoder saturates (step 5 on) -- tools/curve_sensitivity.py,
                                         profiles/r05_curve_sensitivity.txt; c1 under the same perturbation: 0.  batch_hard picks ONE hardest positive and
                                         negative per anchor (min / max + float equality), so a last-bit difference in the Gram matrix re-routes whole
                                         gradient rows.  No implementation whose fp32 sums are ordered differently from TensorFlow's can hold 1e-4 there
                                         (this repo's exact-fp32 MFMA mode measures 8.9e-5 / 1.8e-4): steps 1-4 are held to the north star's 1e-4, the
                                         rest to an ORACLE-DERIVED envelope (c3_gate below: 25 oracle runs, one-ulp and summation-order families; round 5
                                         had hand-sized tail gates here).  precision='auto' resolves to f16x2h for batch_hard too: the cheapest mode well inside
                                         that envelope (f16x2 0.83-1.28 x the gate depending on the Gram's summation order, the same mask without delta1 1.06 x: outside; f16x2h 0.29-0.40 x, bf16x3 0.25 x, f16x3 0.32 x, fp32 0.17 x)
  c4  dense tf-idf ndarray, F = 50000   (N = 1600 rows: 10 epochs of 2 steps)
  c5  explicit triplets, cosine loss    (autoencoder_triplet.py:296-314)"""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))


def _fit(name, precision, tmp):
    import make_curves as M
    from dae_rnn_news_recommendation_amd.autoencoder import DenoisingAutoencoder, DenoisingAutoencoderTriplet
    c, k = M.CFGS[name], M.COMMON
    G = np.load(M.path(name))
    data, lab, W0 = M.inputs(name)
    assert M.checksum(data, lab).tolist() == G["inputs_checksum"].tolist()          # the same regenerated inputs
    kw = dict(model_name=name, main_dir=name, compress_factor=c["cf"], enc_act_func="sigmoid", dec_act_func="sigmoid", loss_func=c["loss"],
              num_epochs=c["epochs"], batch_size=c["batch"], opt="gradient_descent", learning_rate=k["learning_rate"], corr_type="masking",
              corr_frac=k["corr_frac"], verbose=0, verbose_step=1, seed=k["seed"], alpha=k["alpha"], precision=precision, rng="numpy",
              init_weights=W0, results_root=str(tmp) + "/")
    if c["strategy"] == "explicit":
        model = DenoisingAutoencoderTriplet(**kw)
        model.fit({"org": data[0], "pos": data[1], "neg": data[2]})
    else:
        model = DenoisingAutoencoder(triplet_strategy=c["strategy"], **kw)
        model.fit(data, train_set_label=lab if c["strategy"] != "none" else None)
    pb = np.concatenate([model.epoch_stats(e + 1)["per_batch"] for e in range(c["epochs"])])
    gold = {key: G[key].reshape(-1) for key in ("cost", "ae", "triplet", "num")}
    W = model.engine.get_params()[0].astype(np.float64)
    wsum = np.array([np.abs(W).sum(), (W ** 2).sum(), W[17, 3], W[-1, -1]])
    wdev = np.abs(wsum - G["W_checksum"]).max() / np.abs(G["W_checksum"]).max()
    return model, pb, gold, wdev


def _dev(pb, gold, col, key):
    g = gold[key]
    return np.abs(pb[:, col] - g) / np.maximum(np.abs(g), 1e-30)


CASES = [("c1", "auto"), ("c3", "auto"), ("c3", "fp32"), ("c3", "bf16x3"), ("c3", "f16x3"), ("c4", "auto"), ("c5", "auto"), ("c1", "bf16x3")]


def c3_gate(leg):
    """Per-step gate of the batch_hard curve, derived from the ORACLE alone (tests/golden/envelope_c3.npz, make_long_curves.py envelope c3 16 8): 25 float32
    oracle runs of the same 20 steps -- run 0 unperturbed, 16 with ONE of the 5 million initial weights moved by one ulp, 8 with the hidden units permuted (the
    same arithmetic in another summation order) -- give per step the largest pairwise relative deviation and its running maximum `envmono`.  An implementation is
    held to max(1e-4, 3 x envmono) per step, and to the north star's 1e-4 on steps 1-4 (before the decoder saturates nothing is chaotic)."""
    import make_long_curves as ML
    E = np.load(ML.envelope_path("c3"))
    g = np.maximum(1e-4, 3.0 * E["envmono_" + leg])
    g[:4] = 1e-4
    return g, E


@pytest.mark.parametrize("name,precision", CASES, ids=[f"{n}-{p}" for n, p in CASES])
def test_full_shape_curve_of_config(tmp_path, name, precision):
    import make_curves as M
    import make_long_curves as ML
    from dae_rnn_news_recommendation_amd import _lib as L
    if not os.path.exists(M.path(name)):
        pytest.skip(f"{M.path(name)} not generated (python tests/golden/make_curves.py {name})")
    model, pb, gold, wdev = _fit(name, precision, tmp_path)
    c = M.CFGS[name]
    assert pb.shape[0] == 20
    if precision == "auto":
        assert model.precision_used == L.auto_precision(c["strategy"]), (model.precision_used, c["strategy"])
    dc, da = _dev(pb, gold, 0, "cost"), _dev(pb, gold, 1, "ae")
    msg = f"[curve] {name} {precision} (= {model.precision_used}): cost max {dc.max():.2e} (step {int(dc.argmax()) + 1})  ae {da.max():.2e}"
    if c["strategy"] != "none":
        dt = _dev(pb, gold, 2, "triplet")
        msg += f"  triplet {dt.max():.2e} (step {int(dt.argmax()) + 1})"
    print(msg + f"  W checksum {wdev:.2e}")
    gate = np.full(20, 2e-5 if precision == "fp32" else 1e-4)
    gate_a, gate_t = gate.copy(), gate.copy()
    if c["strategy"] == "batch_hard":
        if not os.path.exists(ML.envelope_path("c3")):
            pytest.skip("tests/golden/envelope_c3.npz not generated (python tests/golden/make_long_curves.py envelope c3 16 8)")
        (gate, E), gate_a, gate_t = c3_gate("cost"), c3_gate("ae")[0], c3_gate("triplet")[0]
        assert np.array_equal(E["runs_cost"][0], gold["cost"])                 # the envelope's run 0 IS the frozen curve
        print(f"[curve] c3 {precision}: worst deviation / gate: cost {(dc / gate).max():.2f} (step {int((dc / gate).argmax()) + 1}), triplet "
              f"{(dt / gate_t).max():.2f} (step {int((dt / gate_t).argmax()) + 1}); gate at step 20: cost {gate[-1]:.2e}, triplet {gate_t[-1]:.2e}")
    assert (dc <= gate).all() and (da <= gate_a).all(), (name, precision, dc, da)
    if c["strategy"] != "none":
        assert (dt <= gate_t).all(), (name, precision, dt)
    if c["strategy"] == "batch_hard":
        print(f"[curve] {name} {precision}: hard-triplet count differs from the oracle's by at most {np.abs(pb[:, 4] - gold['num']).max():.0f} of ~800")
        assert np.abs(pb[:, 4] - gold["num"]).max() <= 1, (pb[:, 4], gold["num"])        # (every oracle run of the envelope counts 800 on all 20 steps)
    assert wdev <= (1e-5 if precision == "fp32" and c["strategy"] != "batch_hard" else 3e-4), wdev
