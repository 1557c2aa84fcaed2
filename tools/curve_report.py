#!/usr/bin/env python3
"""Per-step relative errors of the full-shape (8000 x 10000, B 800, batch_all) loss curve against the float32 oracle's golden
curve (tests/golden/full_curve_c2.npz): fp32 mode, bf16 mode with the encode reading the fp32 master weights (default), and
bf16 mode with the encode reading W_lo.  usage: python tools/curve_report.py > profiles/rNN_curve_report.txt"""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np
import make_full_curve as M
from dae_rnn_news_recommendation_amd.autoencoder import DenoisingAutoencoder

G = np.load(os.path.join(ROOT, "tests", "golden", "full_curve_c2.npz"))
c = M.CFG
m, lab, W0 = M.inputs()
np.set_printoptions(linewidth=250, precision=2)
for name, precision, opts in (("fp32", "fp32", {}), ("bf16 encode_w32=1 (default)", "bf16", {"encode_w32": 1}),
                              ("bf16 encode_w32=0", "bf16", {"encode_w32": 0}),
                              ("bf16 encode_w32=1 gram_fp32", "bf16", {"encode_w32": 1, "gram_fp32": 1})):
    with tempfile.TemporaryDirectory() as tmp:
        model = DenoisingAutoencoder(model_name="full", main_dir="full", compress_factor=c["compress_factor"], enc_act_func="sigmoid",
                                     dec_act_func="sigmoid", loss_func="cross_entropy", num_epochs=c["epochs"], batch_size=c["batch"],
                                     opt="gradient_descent", learning_rate=c["learning_rate"], corr_type="masking", corr_frac=c["corr_frac"],
                                     verbose=0, verbose_step=1, seed=c["seed"], alpha=c["alpha"], triplet_strategy="batch_all",
                                     precision=precision, rng="numpy", init_weights=W0, results_root=tmp + "/", plan_options=opts)
        try:
            model.fit(m, train_set_label=lab)
        except Exception as e:      # noqa: BLE001
            print("==", name, "FAILED:", e); continue
        print("==", name)
        for col, key in ((0, "cost"), (1, "ae"), (2, "triplet")):
            rel = np.concatenate([np.abs(model.epoch_stats(e + 1)["per_batch"][:, col] - G[key][e]) / np.abs(G[key][e]) for e in range(c["epochs"])])
            print("  %-8s max %.2e  per step: %s" % (key, rel.max(), " ".join("%.1e" % v for v in rel)))
