#!/usr/bin/env python3
"""Freeze the per-batch loss curve of BASELINE.json configs[1] at FULL shape (8000 x 10000 binary CSR, H = 500, B = 800,
batch_all, masking 0.3 with the reference-exact legacy-RNG stream, injected W0, SGD lr 0.1): 2 epochs = 20 steps of the
oracle (oracle.fit_reference, pinned to the reference's own fit loop by tests/test_golden_graph.py) IN FLOAT32, the
reference's dtype.  That matters at this shape: with lr 0.1 the decoder saturates from the 4th step on (y rounds to exactly
1.0f for z > ~17), where TF's literal cross entropy evaluates log(1 - y + 1e-16) = log(1e-16) and its autodiff yields
d loss/d z = 1e16 * y * (1 - y) = 0 for those units -- float64 arithmetic does neither (a float64 run of the same loop is
5 % away at step 4 and 120 % away at step 10), so the float32 restatement is the oracle here.
The GPU test (tests/test_hip_full_curve.py) runs DenoisingAutoencoder.fit() on the same regenerated inputs and compares
batch by batch.  Takes a few minutes of CPU; the matrix is NOT stored (it is regenerated from the seeded generator).

usage: python tests/golden/make_full_curve.py [--out tests/golden/full_curve_c2.npz]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

CFG = dict(rows=8000, features=10000, compress_factor=20, batch=800, epochs=2, seed=0, data_seed=1234, w_seed=42, corr_frac=0.3,
           learning_rate=0.1, alpha=1.0)


def inputs():
    from dae_rnn_news_recommendation_amd.synthetic import synthetic_csr, synthetic_labels, xavier_uniform
    c = CFG
    m = synthetic_csr(c["rows"], c["features"], nnz_per_row=200, seed=c["data_seed"])
    lab = synthetic_labels(c["rows"], kind="category", seed=c["data_seed"])
    W0 = xavier_uniform(c["features"], c["features"] // c["compress_factor"], seed=c["w_seed"])
    return m, lab, W0


def main():
    import oracle as O
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "full_curve_c2.npz"))
    a = ap.parse_args()
    c = CFG
    m, lab, W0 = inputs()
    t0 = time.time()
    r = O.fit_reference(m, lab, W0, enc_act="sigmoid", dec_act="sigmoid", loss_func="cross_entropy", num_epochs=c["epochs"],
                        batch_size=c["batch"], opt="gradient_descent", learning_rate=c["learning_rate"], corr_type="masking",
                        corr_frac=c["corr_frac"], seed=c["seed"], alpha=c["alpha"], triplet_strategy="batch_all", dt=np.float32)
    out = {k: np.array([h[k] for h in r["history"]], np.float64) for k in ("cost", "ae", "triplet", "fraction", "num")}
    out["W_checksum"] = np.array([np.abs(r["W"]).sum(), (r["W"] ** 2).sum(), r["W"][17, 3], r["W"][9999, 499]])
    out["indices_checksum"] = np.array([m.nnz, int(m.indices[::997].astype(np.int64).sum()), int(lab.sum())], np.int64)
    np.savez_compressed(a.out, **out)
    print("wrote", a.out, "in %.0f s" % (time.time() - t0), out["cost"])


if __name__ == "__main__":
    main()
