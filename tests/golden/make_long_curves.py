#!/usr/bin/env python3
"""Round-6 fixtures behind the north star's CURVE gate ("loss curve matching reference within 1e-4"), both from the FLOAT32 oracle with the
reference-exact legacy-RNG stream, exactly like make_full_curve.py / make_curves.py (whose inputs and hyper-parameters they reuse):

  long c1|c2|c4|c5 100 per-batch steps (10 epochs of 10 batches; reference loop autoencoder.py:175-246, the values its epoch line averages at :283-294;
                  the CLI's default run is 50 epochs, main_autoencoder.py:71-72) -> long_curve_<cfg>.npz.  The 20-step files stop while the default
                  mode's deviation is still growing; these say where it goes.
  envelope c3 [K [K_order]] the oracle's OWN determinacy of the batch_hard curve (triplet_loss_utils.py:202-259: min / max + float equality route every anchor's
                  gradient through one hardest positive and one hardest negative): K runs of the frozen 20-step c3 curve, each with ONE initial
                  weight (seeded position) moved by +-1 float32 ulp, plus K_order runs of the SAME arithmetic in another summation order (the hidden
                  units permuted: see perturbed()); run 0 is the unperturbed curve.  Stored: every run's cost / ae / triplet / num per step and, per
                  step, the largest pairwise relative deviation among the runs (`env_*`; `env_ulp_*` over the one-ulp family alone) and its running
                  maximum (`envmono_*`).  tests/test_hip_curves.py gates c3 at max(1e-4, 3 x envmono) per step -- an oracle-derived envelope instead
                  of a hand-sized tail.
  envelope c2 [K] the same for c2 over the 100-step horizon (context for the long curve: does the reference's arithmetic pin ITS OWN step 100 to 1e-4?)

CPU only; c2 costs ~50 s per STEP (the literal B^3 batch_all in float32: 86 minutes for its 100 steps), c4 57 s per step, c1 / c5 1-4 s.  usage: python tests/golden/make_long_curves.py long c2 [epochs] | envelope c3 16 8   (long c1 50 / long c5 50: the CLI's default 50 epochs = 500 steps)"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

LONG_EPOCHS = 10
KEYS = ("cost", "ae", "triplet", "fraction", "num")


def long_path(name, epochs=None):
    """long_curve_<cfg>.npz = the 100-step curve; long_curve_<cfg>_e<epochs>.npz = a longer one (c1 / c5 at the CLI's default 50 epochs = 500 steps)."""
    return os.path.join(HERE, f"long_curve_{name}.npz" if epochs in (None, LONG_EPOCHS) or name == "c4" else f"long_curve_{name}_e{epochs}.npz")


def envelope_path(name):
    return os.path.join(HERE, f"envelope_{name}.npz")


def config(name):
    """(data, labels, W0, fit kwargs, strategy) of c1 / c2 / c3 as the 20-step generators define them."""
    import make_curves as M
    import make_full_curve as MF
    if name == "c2":
        c = MF.CFG
        data, lab, W0 = MF.inputs()
        kw = dict(loss_func="cross_entropy", batch_size=c["batch"], learning_rate=c["learning_rate"], corr_frac=c["corr_frac"], seed=c["seed"],
                  alpha=c["alpha"], triplet_strategy="batch_all")
        return data, lab, W0, kw, c["epochs"]
    c, k = M.CFGS[name], M.COMMON
    data, lab, W0 = M.inputs(name)
    kw = dict(loss_func=c["loss"], batch_size=c["batch"], learning_rate=k["learning_rate"], corr_frac=k["corr_frac"], seed=k["seed"], alpha=k["alpha"],
              triplet_strategy=c["strategy"])
    return data, (lab if c["strategy"] != "none" else None), W0, kw, c["epochs"]


def run(data, lab, W0, kw, epochs):
    import oracle as O
    r = O.fit_reference(data, lab, W0, enc_act="sigmoid", dec_act="sigmoid", num_epochs=epochs, opt="gradient_descent", corr_type="masking",
                        dt=np.float32, **kw)
    out = {k: np.array([h[k] for h in r["history"]], np.float64).reshape(-1) for k in KEYS}
    W = r["W"].astype(np.float64)
    out["W_checksum"] = np.array([np.abs(W).sum(), (W ** 2).sum(), W[17, 3], W[-1, -1]])
    return out


def make_long(name, epochs=None):
    import make_curves as M
    t0 = time.time()
    epochs = epochs or LONG_EPOCHS
    if name == "c5":           # explicit (anchor, pos, neg) triplets: the estimator's own epoch loop restated in make_curves.fit_explicit
        data, lab, W0 = M.inputs("c5")
        r = M.fit_explicit(data, W0, dict(M.CFGS["c5"], epochs=epochs))
        out = {k: np.array([h[k] for h in r["history"]], np.float64).reshape(-1) for k in KEYS}
        W = r["W"].astype(np.float64)
        out["W_checksum"] = np.array([np.abs(W).sum(), (W ** 2).sum(), W[17, 3], W[-1, -1]])
        out["inputs_checksum"] = M.checksum(data, lab)
        np.savez_compressed(long_path(name, epochs), **out)
        print("wrote", long_path(name, epochs), "in %.0f s" % (time.time() - t0), "cost", out["cost"][[0, 19, -1]], flush=True)
        return
    data, lab, W0, kw, _ = config(name)
    out = run(data, lab, W0, kw, 50 if name == "c4" else epochs)       # (c4: 1600 rows = 2 steps per epoch -> 50 epochs for 100 steps)
    out["inputs_checksum"] = M.checksum(data, lab)
    np.savez_compressed(long_path(name, epochs), **out)
    print("wrote", long_path(name, epochs), "in %.0f s" % (time.time() - t0), "cost", out["cost"][[0, 19, -1]], flush=True)


def perturbed(W0, k, n_ulp):
    """Member k of the envelope family (k = 0: the unperturbed run).
    k = 1 .. n_ulp: ONE weight moved by one float32 ulp (position and direction from a generator seeded with k) -- the smallest perturbation there is.
    k > n_ulp:      the SAME arithmetic in another summation order -- the hidden units (columns of W0) permuted.  Mathematically nothing changes (h's columns
                    are permuted with them; D = h h^T, the decode h W^T and every loss are invariant; the data, the corruption stream and the batch order
                    are untouched), but every float32 sum over the hidden axis (Gram matrix, decode logits, their backward products) meets its terms in
                    another order -- what separates ANY other float32 implementation (TensorFlow's Eigen kernels included) from this NumPy restatement."""
    if k == 0:
        return W0, (-1, -1, 0)
    g = np.random.default_rng(9000 + k)
    if k > n_ulp:
        return np.ascontiguousarray(W0[:, g.permutation(W0.shape[1])]), (-2, -2, k)
    i, j = int(g.integers(W0.shape[0])), int(g.integers(W0.shape[1]))
    up = bool(g.integers(2))
    W1 = W0.copy()
    W1[i, j] = np.nextafter(W1[i, j], np.float32(np.inf if up else -np.inf))
    return W1, (i, j, 1 if up else -1)


def pairwise_envelope(curves, base):
    """curves [K+1, steps]: per step the largest |C_i - C_j| over all pairs, relative to |base| (the unperturbed run)."""
    hi, lo = curves.max(axis=0), curves.min(axis=0)
    return (hi - lo) / np.maximum(np.abs(base), 1e-30)


def make_envelope(name, K, K_order=0):
    import make_curves as M
    data, lab, W0, kw, epochs = config(name)
    if name != "c3":
        epochs = LONG_EPOCHS
    runs, where = [], []
    for k in range(K + K_order + 1):
        t0 = time.time()
        Wk, w = perturbed(W0, k, K)
        runs.append(run(data, lab, Wk, kw, epochs)); where.append(w)
        print(f"envelope {name}: run {k} (weight {w}) in {time.time() - t0:.0f} s", flush=True)
    out = {"where": np.array(where, np.int64), "inputs_checksum": M.checksum(data, lab), "n_ulp": np.array(K), "n_order": np.array(K_order)}
    for q in ("cost", "ae", "triplet", "num"):
        C = np.stack([r[q] for r in runs])
        out["runs_" + q] = C
        if q != "num" and np.abs(C[0]).max() > 0:
            for tag, rows in (("", slice(None)), ("ulp_", slice(0, K + 1))):        # env_* over every member; env_ulp_* over the one-ulp family alone
                env = pairwise_envelope(C[rows], C[0])
                out["env_" + tag + q] = env
                out["envmono_" + tag + q] = np.maximum.accumulate(env)
                print(f"envelope {name} {tag}{q}: max pairwise relative deviation {env.max():.2e} at step {int(env.argmax()) + 1}", flush=True)
    if name == "c3":            # run 0 must BE the frozen 20-step curve
        G = np.load(M.path("c3"))
        assert np.array_equal(out["runs_cost"][0], G["cost"].reshape(-1)), "unperturbed run differs from full_curve_c3.npz"
    np.savez_compressed(envelope_path(name), **out)
    print("wrote", envelope_path(name), flush=True)


if __name__ == "__main__":
    mode, name = sys.argv[1], sys.argv[2]
    if mode == "long":
        make_long(name, int(sys.argv[3]) if len(sys.argv) > 3 else None)
    elif mode == "envelope":
        make_envelope(name, int(sys.argv[3]) if len(sys.argv) > 3 else 16, int(sys.argv[4]) if len(sys.argv) > 4 else 8)
    else:
        raise SystemExit(__doc__)
