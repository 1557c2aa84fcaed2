#!/usr/bin/env python3
"""20-step full-shape loss curve (tests/golden/full_curve_c2.npz, what tests/test_hip_full_curve.py holds the GPU to) for a list of lo-term masks of
the split 16-bit mode (plan option x3_terms, bits in include/dae_hip.h) and the step time of each: which product terms does the 1e-4 gate need?
usage: python tools/curve_terms.py --precision f16x2 --terms 261,277,325,341 [--time]"""
import argparse
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--precision", default="f16x2")
    ap.add_argument("--terms", default="261")
    ap.add_argument("--scale-log2", default="")
    ap.add_argument("--time", action="store_true")
    a = ap.parse_args()
    import torch
    import make_full_curve as M
    from dae_rnn_news_recommendation_amd.autoencoder import DenoisingAutoencoder
    G = np.load(os.path.join(ROOT, "tests", "golden", "full_curve_c2.npz"))
    c = M.CFG
    m, lab, W0 = M.inputs()
    for t in [int(x) for x in a.terms.split(",")]:
        for sc in ([int(x) for x in a.scale_log2.split(",")] if a.scale_log2 else [None]):
            opts = {"x3_terms": t}
            if sc is not None:
                opts["op_scale_log2"] = sc
            with tempfile.TemporaryDirectory() as tmp:
                model = DenoisingAutoencoder(model_name="full", main_dir="full", compress_factor=c["compress_factor"], enc_act_func="sigmoid",
                                             dec_act_func="sigmoid", loss_func="cross_entropy", num_epochs=c["epochs"], batch_size=c["batch"],
                                             opt="gradient_descent", learning_rate=c["learning_rate"], corr_type="masking", corr_frac=c["corr_frac"],
                                             verbose=0, verbose_step=1, seed=c["seed"], alpha=c["alpha"], triplet_strategy="batch_all",
                                             precision=a.precision, rng="numpy", init_weights=W0, results_root=tmp + "/", plan_options=opts)
                model.fit(m, train_set_label=lab)
                dev = {}
                for col, key in ((0, "cost"), (1, "ae"), (2, "triplet")):
                    pb = np.concatenate([model.epoch_stats(e + 1)["per_batch"][:, col] for e in range(c["epochs"])])
                    g = np.concatenate([G[key][e] for e in range(c["epochs"])])
                    rel = np.abs(pb - g) / np.abs(g)
                    dev[key] = (rel.max(), int(rel.argmax()))
                us = ""
                if a.time:
                    eng = model.engine
                    stats = torch.zeros(8, device="cuda")
                    idx = torch.arange(c["batch"], dtype=torch.int32, device="cuda")
                    labs = torch.from_numpy(lab[:c["batch"]].astype(np.int32)).cuda()
                    from dae_rnn_news_recommendation_amd import _lib as L
                    kw = dict(corr_mode=L.CORR_PHILOX_MASK, seed=1, rng_stream=0, corr_frac=0.3, phase=3)
                    for _ in range(30):
                        eng.train_step(idx, labs, stats, **kw)
                    torch.cuda.synchronize(); t0 = time.perf_counter()
                    for _ in range(300):
                        eng.train_step(idx, labs, stats, **kw)
                    torch.cuda.synchronize()
                    us = "  step %.1f us" % ((time.perf_counter() - t0) / 300 * 1e6)
                print(f"{a.precision} x3_terms={t:4d} ({t:011b}) scale_log2={sc}: cost {dev['cost'][0]:.2e} (step {dev['cost'][1] + 1})  "
                      f"triplet {dev['triplet'][0]:.2e} (step {dev['triplet'][1] + 1}){us}", flush=True)


if __name__ == "__main__":
    main()
