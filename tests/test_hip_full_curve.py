"""Full-shape loss curve of BASELINE.json configs[1] (north star: "loss curve matching reference within 1e-4" on
8000 x 10000 batch_all): DenoisingAutoencoder.fit() for 2 epochs = 20 steps against the FLOAT32 oracle's per-batch costs
frozen by tests/golden/make_full_curve.py (same regenerated inputs, reference-exact legacy RNG, injected W0).

With the CLI's lr 0.1 the decoder saturates from the 5th step on: logits beyond ~17 round y to exactly 1.0f, the reference's
literal cross entropy then charges log(1e-16) = -36.8 for such a unit instead of ~-z, and its autodiff gives it a zero gradient.
The fp32 step reproduces that to 1e-5 over all 20 steps.  With bf16 MFMA operands a logit carries ~4e-3 relative error, so units
within ~0.07 of the rounding threshold land on the other side of a 20-unit jump of the loss: steps in the saturated regime are
held to 6e-4 (measured: <= 2.8e-4 at steps 4-7, <= 6e-6 from step 10 on; profiles/r03_bf16_curve.txt), the steps before it to the
1e-4 gate.  The triplet leg (0.7 of a cost of 3500-7600) is checked on its own: <= 1e-4 for steps 0-2 (measured 2.6e-5), 5e-4 at
step 3, 1e-2 afterwards (measured: growing to 6.8e-3 at step 19) -- the encode reads the fp32 master weights, so h is exact GIVEN W;
what drifts is W itself (bf16 delta2 / h / W operands in the three gradient GEMMs, accumulated over the steps), and the triplet
loss ~ 0.69 + mean(T)/2 turns a 1e-2 shift of the mean distance gap into 7e-3.  precision='fp32' holds 2e-5 throughout."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
PATH = os.path.join(HERE, "golden", "full_curve_c2.npz")


# bf16x3-dropped-terms: plan options x3_dec_wlo = 0 / x3_dh_hlo = 0 drop the two lo product terms a CPU replay of this curve called droppable
# (decode h_hi.W_lo, dh Gs.h^T_lo: tools/precision_study.py --per-term predicted cost 2.5e-5 / triplet 4.4e-5).  Measured here: cost 7.0e-5,
# triplet 1.56e-4 -- OUTSIDE the 1e-4 gate, which is why the product keeps all terms (profiles/r04_precision_terms.txt); the case pins that measurement
DROPPED = {"x3_dec_wlo": 0, "x3_dh_hlo": 0}
# f16x2: the fp16 build's parity mode -- fp16 operand images, only W kept as hi + lo (lo terms: decode (h, W_lo), dh (delta2, W^T_lo)); the CPU replay
# (tools/precision_study.py --golden --scheme W=f16split) predicted cost 1.4e-5 / triplet 6.5e-5.  f16x2-h-split adds the three h terms (decode (h_lo, W),
# dh (Gs, h^T_lo), dW (delta2^T, h^T_lo)): predicted 1.0e-5 / 2.75e-5 -- the fallback segment list should the two-term form leave the gate
F16_WH = {"x3_terms": 1 | 4 | 2 | 16 | 64}


@pytest.mark.skipif(not os.path.exists(PATH), reason="tests/golden/full_curve_c2.npz not generated")
@pytest.mark.parametrize("precision,tol,tol_saturated,plan_options",
                         [("fp32", 2e-5, 2e-5, None), ("bf16", 1e-4, 6e-4, None), ("bf16x3", 1e-4, 1e-4, None), ("bf16x3", 1e-4, 5e-4, DROPPED),
                          ("f16x2", 1e-4, 1e-4, None), ("f16x2", 1e-4, 1e-4, F16_WH), ("auto", 1e-4, 1e-4, None)],
                         ids=["fp32", "bf16", "bf16x3", "bf16x3-dropped-terms", "f16x2", "f16x2-h-split", "auto"])
def test_full_shape_loss_curve(tmp_path, precision, tol, tol_saturated, plan_options):
    sys.path.insert(0, os.path.join(HERE, "golden"))
    import make_full_curve as M
    from dae_rnn_news_recommendation_amd.autoencoder import DenoisingAutoencoder
    G = np.load(PATH)
    c = M.CFG
    m, lab, W0 = M.inputs()
    assert [m.nnz, int(m.indices[::997].astype(np.int64).sum()), int(lab.sum())] == G["indices_checksum"].tolist()   # same inputs
    model = DenoisingAutoencoder(model_name="full", main_dir="full", compress_factor=c["compress_factor"], enc_act_func="sigmoid",
                                 dec_act_func="sigmoid", loss_func="cross_entropy", num_epochs=c["epochs"], batch_size=c["batch"],
                                 opt="gradient_descent", learning_rate=c["learning_rate"], corr_type="masking", corr_frac=c["corr_frac"],
                                 verbose=0, verbose_step=1, seed=c["seed"], alpha=c["alpha"], triplet_strategy="batch_all",
                                 precision=precision, rng="numpy", init_weights=W0, results_root=str(tmp_path) + "/", plan_options=plan_options)
    model.fit(m, train_set_label=lab)
    for e in range(c["epochs"]):
        pb = model.epoch_stats(e + 1)["per_batch"]
        for col, key in ((0, "cost"), (1, "ae"), (2, "triplet")):
            rel = np.abs(pb[:, col] - G[key][e]) / np.abs(G[key][e])
            gate = np.where(np.arange(pb.shape[0]) + e * pb.shape[0] < 4, tol, tol_saturated)     # steps 0-3: no saturated logit yet
            if key == "triplet" and precision == "bf16":
                step = np.arange(pb.shape[0]) + e * pb.shape[0]
                gate = np.where(step < 3, 1e-4, np.where(step == 3, 5e-4, 1e-2))
            print(f"[curve] {precision}{' ' + str(plan_options) if plan_options else ''} epoch {e} {key}: max rel {rel.max():.2e} at batch {int(rel.argmax())}")
            assert (rel <= gate).all(), (precision, e, key, rel)
        if precision == "fp32":
            assert np.abs(pb[:, 4] - G["num"][e]).max() <= 200        # of ~5*10^7 positive triplets: near-ties of the fp32 Gram matrix
    W = model.engine.get_params()[0].astype(np.float64)
    got = np.array([np.abs(W).sum(), (W ** 2).sum(), W[17, 3], W[9999, 499]])
    wdev = np.abs(got - G["W_checksum"]).max() / np.abs(G["W_checksum"]).max()
    print(f"[curve] {precision} W checksum deviation {wdev:.2e}; resolved precision {model.precision_used}")
    gate_w = {"fp32": 1e-5, "bf16x3": 1e-4, "f16x2": 3e-4, "auto": 3e-4}.get(precision, 5e-3)
    assert wdev <= (gate_w if (not plan_options or precision == "f16x2") else 5e-3)
