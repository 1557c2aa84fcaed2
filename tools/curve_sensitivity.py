#!/usr/bin/env python3
"""How well does the reference's OWN float32 arithmetic determine its 20-step loss curves?  CPU experiment on the oracle (no GPU): re-run the frozen
full-shape curve of a config with ONE of the ~5 million initial weights moved by one float32 ulp (a 6e-8 relative change of one number) and report the
largest relative deviation from the frozen curve.  A curve that moves by more than 1e-4 under that perturbation cannot be matched to 1e-4 by ANY
implementation whose float32 sums are not ordered exactly like TensorFlow's (neither this repo's exact-fp32 MFMA mode nor NumPy's BLAS are): the gate
of tests/test_hip_curves.py for that config is then set from this measurement instead of the north star's 1e-4.
usage: python tools/curve_sensitivity.py c1 c2 c3 c5   (c2: tests/golden/full_curve_c2.npz)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def curve(name, W0, data, lab):
    import make_curves as M
    import make_full_curve as MF
    import oracle as O
    if name == "c2":
        c = MF.CFG
        r = O.fit_reference(data, lab, W0, enc_act="sigmoid", dec_act="sigmoid", loss_func="cross_entropy", num_epochs=c["epochs"], batch_size=c["batch"],
                            opt="gradient_descent", learning_rate=c["learning_rate"], corr_type="masking", corr_frac=c["corr_frac"], seed=c["seed"],
                            alpha=c["alpha"], triplet_strategy="batch_all", dt=np.float32)
    elif M.CFGS[name]["strategy"] == "explicit":
        r = M.fit_explicit(data, W0, M.CFGS[name])
    else:
        c, k = M.CFGS[name], M.COMMON
        r = O.fit_reference(data, lab if c["strategy"] != "none" else None, W0, enc_act="sigmoid", dec_act="sigmoid", loss_func=c["loss"],
                            num_epochs=c["epochs"], batch_size=c["batch"], opt="gradient_descent", learning_rate=k["learning_rate"], corr_type="masking",
                            corr_frac=k["corr_frac"], seed=k["seed"], alpha=k["alpha"], triplet_strategy=c["strategy"], dt=np.float32)
    return {q: np.array([h[q] for h in r["history"]], np.float64).reshape(-1) for q in ("cost", "ae", "triplet")}


def main():
    import make_curves as M
    import make_full_curve as MF
    for name in sys.argv[1:] or ["c1", "c3", "c5", "c2"]:
        if name == "c2":
            data, lab, W0 = MF.inputs(); G = np.load(os.path.join(ROOT, "tests", "golden", "full_curve_c2.npz"))
        else:
            data, lab, W0 = M.inputs(name); G = np.load(M.path(name))
        g = {q: G[q].reshape(-1) for q in ("cost", "ae", "triplet")}
        same = curve(name, W0, data, lab)
        rep = max(np.abs(same[q] - g[q]).max() for q in g)
        W1 = W0.copy(); W1[123, 45] = np.nextafter(W1[123, 45], np.float32(1))
        b = curve(name, W1, data, lab)
        out = []
        for q in ("cost", "ae", "triplet"):
            if np.abs(g[q]).max() == 0:
                continue
            d = np.abs(b[q] - g[q]) / np.abs(g[q])
            out.append(f"{q} {d.max():.2e} (step {int(d.argmax()) + 1})")
        print(f"{name}: re-run reproduces the frozen curve to {rep:.1e} (absolute); one weight moved by 1 ulp -> max relative deviation  " + "  ".join(out), flush=True)


if __name__ == "__main__":
    main()
