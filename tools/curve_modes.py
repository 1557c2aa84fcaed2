#!/usr/bin/env python3
"""Which precision mode holds which curve?  DenoisingAutoencoder.fit() on the GPU, per mode, against the round-6 fixtures of
tests/golden/make_long_curves.py:
  --config c1|c2  the 100-step float32-oracle curve (long_curve_<cfg>.npz): max relative deviation per leg, the first step that leaves 1e-4
  --config c3     the 20-step batch_hard curve against the oracle's own envelope (envelope_c3.npz: K+1 oracle runs, one weight +-1 ulp each):
                  per leg the largest deviation from run 0 and the largest ratio deviation / gate, gate = max(1e-4, 3 x running envelope)
A mode is `name[:x3_terms][:option=value ...]` (plan options; x3_terms = the lo-term mask of the split 16-bit modes).  Also prints the step time of each mode (--time).
usage: python tools/curve_modes.py --config c3 --modes f16x2,f16x2:343,f16x3,bf16x3,fp32 [--time]"""
import argparse
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def fit_curve(name, mode, data, lab, W0, kw, epochs, opts):
    from dae_rnn_news_recommendation_amd.autoencoder import DenoisingAutoencoder
    import make_curves as M
    import make_full_curve as MF
    cf = MF.CFG["compress_factor"] if name == "c2" else M.CFGS[name]["cf"]
    with tempfile.TemporaryDirectory() as tmp:
        model = DenoisingAutoencoder(model_name=name, main_dir=name, compress_factor=cf, enc_act_func="sigmoid", dec_act_func="sigmoid",
                                     loss_func=kw["loss_func"], num_epochs=epochs, batch_size=kw["batch_size"], opt="gradient_descent",
                                     learning_rate=kw["learning_rate"], corr_type="masking", corr_frac=kw["corr_frac"], verbose=0, verbose_step=1,
                                     seed=kw["seed"], alpha=kw["alpha"], triplet_strategy=kw["triplet_strategy"], precision=mode, rng="numpy",
                                     init_weights=W0, results_root=tmp + "/", plan_options=opts)
        model.fit(data, train_set_label=lab)
        pb = np.concatenate([model.epoch_stats(e + 1)["per_batch"] for e in range(epochs)])
        return model, pb


def step_time(model, lab, B, strategy):
    import torch
    from dae_rnn_news_recommendation_amd import _lib as L
    eng = model.engine
    stats = torch.zeros(8, device="cuda")
    idx = torch.arange(B, dtype=torch.int32, device="cuda")
    labs = torch.from_numpy(np.sort(np.asarray(lab[:B])).astype(np.int32)).cuda() if strategy != "none" else None
    kw = dict(corr_mode=L.CORR_PHILOX_MASK, seed=1, rng_stream=0, corr_frac=0.3, phase=3)
    for _ in range(30):
        eng.train_step(idx, labs, stats, **kw)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(300):
        eng.train_step(idx, labs, stats, **kw)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 300 * 1e6


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c2")
    ap.add_argument("--modes", default="f16x2")
    ap.add_argument("--time", action="store_true")
    ap.add_argument("--epochs", type=int, default=0, help="c1 / c5: compare against long_curve_<cfg>_e<epochs>.npz (50 = the CLI's default run, 500 steps)")
    a = ap.parse_args()
    import make_long_curves as ML
    name = a.config
    data, lab, W0, kw, epochs = ML.config(name)
    if name == "c3":
        E = np.load(ML.envelope_path("c3"))
        gold = {q: E["runs_" + q][0] for q in ("cost", "ae", "triplet")}
        gate = {q: np.maximum(1e-4, 3.0 * E["envmono_" + q]) for q in ("cost", "ae", "triplet")}
        gate = {q: np.where(np.arange(len(g)) < 4, 1e-4, g) for q, g in gate.items()}
    else:
        epochs = 50 if name == "c4" else (a.epochs or ML.LONG_EPOCHS)
        G = np.load(ML.long_path(name, epochs))
        gold = {q: G[q] for q in ("cost", "ae", "triplet")}
        gate = {q: np.full(len(G["cost"]), 1e-4) for q in gold}
    for spec in a.modes.split(","):
        parts = spec.split(":")                 # name[:x3_terms][:option=value ...]
        mode = parts[0]
        opts = {}
        for q in parts[1:]:
            if "=" in q:
                k, v = q.split("="); opts[k] = int(v)
            elif q:
                opts["x3_terms"] = int(q)
        opts = opts or None
        model, pb = fit_curve(name, mode, data, lab, W0, kw, epochs, opts)
        out = []
        for col, q in ((0, "cost"), (1, "ae"), (2, "triplet")):
            if np.abs(gold[q]).max() == 0:
                continue
            d = np.abs(pb[:, col] - gold[q]) / np.abs(gold[q])
            over = np.nonzero(d > gate[q])[0]
            ratio = d / gate[q]
            out.append(f"{q} max {d.max():.2e} (step {int(d.argmax()) + 1}), at step 20 {d[:20].max():.2e}, worst dev/gate {ratio.max():.2f} "
                       f"(step {int(ratio.argmax()) + 1}), first step outside {'none' if len(over) == 0 else int(over[0]) + 1}")
        us = "  step %.1f us" % step_time(model, lab, kw["batch_size"], kw["triplet_strategy"]) if a.time else ""
        print(f"[{name}] {spec} (= {model.precision_used}): " + "; ".join(out) + us, flush=True)


if __name__ == "__main__":
    main()
