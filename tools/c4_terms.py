#!/usr/bin/env python3
"""20-step full-shape curve of config c4 (dense tf-idf ndarray, F = 50000; tests/golden/full_curve_c4.npz) for a list of lo-term masks of the split mode:
does the dense-input ENCODE need its (x~, W^T_lo) term (x3_terms bit 8)?   usage: python tools/c4_terms.py 261,5"""
import os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_curves as M
from dae_rnn_news_recommendation_amd.autoencoder import DenoisingAutoencoder
name = "c4"; c, k = M.CFGS[name], M.COMMON
G = np.load(M.path(name)); data, lab, W0 = M.inputs(name)
for t in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "261,5").split(",")]:
    with tempfile.TemporaryDirectory() as tmp:
        m = DenoisingAutoencoder(model_name=name, main_dir=name, compress_factor=c["cf"], enc_act_func="sigmoid", dec_act_func="sigmoid", loss_func=c["loss"],
                                 num_epochs=c["epochs"], batch_size=c["batch"], opt="gradient_descent", learning_rate=k["learning_rate"], corr_type="masking",
                                 corr_frac=k["corr_frac"], verbose=0, verbose_step=1, seed=k["seed"], alpha=k["alpha"], precision="f16x2", rng="numpy",
                                 init_weights=W0, results_root=tmp + "/", triplet_strategy=c["strategy"], plan_options={"x3_terms": t})
        m.fit(data, train_set_label=lab)
        pb = np.concatenate([m.epoch_stats(e + 1)["per_batch"] for e in range(c["epochs"])])
    dev = {key: np.abs(pb[:, col] - G[key].reshape(-1)) / np.abs(G[key].reshape(-1)) for col, key in ((0, "cost"), (2, "triplet"))}
    print(f"c4 f16x2 x3_terms={t:4d} ({t:011b}): cost {dev['cost'].max():.2e} (step {int(dev['cost'].argmax()) + 1})  triplet {dev['triplet'].max():.2e} (step {int(dev['triplet'].argmax()) + 1})", flush=True)
