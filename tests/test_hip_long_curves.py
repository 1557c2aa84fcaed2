"""100-step (10-epoch; c4: 50 epochs of its 1600-row set) full-shape loss curves of BASELINE.json configs c1, c2, c4 and c5 against the FLOAT32 oracle's per-batch losses frozen by
tests/golden/make_long_curves.py -- the north star's gate is a CURVE ("loss curve matching reference within 1e-4"), the reference CLI trains 50 epochs
(main_autoencoder.py:71-72; the per-batch values its epoch line averages: autoencoder.py:283-294), and a 20-step curve stops while a low-precision mode's
deviation is still growing: round 5's default 'f16x2' (fp16 images, W alone hi + lo) holds 20 steps and then leaves 1e-4 at step 37 of c2 (triplet leg,
2.8e-4 at step 42) and at step 76 of c1 (cost 2.7e-4 and growing) -- measured here and pinned below.  What precision='auto' resolves to per strategy
(_lib.AUTO_BY_STRATEGY) is the cheapest mode that holds all 100 steps: 'f16x2d' for strategy none (delta2 as hi + lo in BOTH gradient GEMMs: with the
lo image in only one of them the curve leaves the gate by 8.7e-3), 'f16x2h' for batch_all (W + every h term + delta1).
Measurements per mode and lo-term mask: profiles/r06_curve_modes.txt (tools/curve_modes.py)."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))

# (config, precision, gate on every step, what the case pins)
CASES = [("c2", "auto", 1e-4), ("c1", "auto", 1e-4), ("c5", "auto", 1e-4), ("c4", "auto", 1e-4), ("c2", "bf16x3", 2e-5), ("c1", "bf16x3", 2e-5),
         ("c2", "f16x2", 6e-4), ("c1", "f16x2", 6e-4), ("c1@50", "auto", 1e-4), ("c5@50", "auto", 1e-4)]      # @50: the CLI's default 50 epochs = 500 steps


def _epochs(name):
    import make_long_curves as ML
    if "@" in name:
        return int(name.split("@")[1])
    return 50 if name == "c4" else ML.LONG_EPOCHS          # c4's set is 1600 rows = 2 steps per epoch


def _fit(case, precision, tmp):
    import make_curves as M
    import make_long_curves as ML
    from dae_rnn_news_recommendation_amd.autoencoder import DenoisingAutoencoder, DenoisingAutoencoderTriplet
    name = case.split("@")[0]
    G = np.load(ML.long_path(name, _epochs(case)))
    if name == "c5":
        c, k = M.CFGS["c5"], M.COMMON
        data, lab, W0 = M.inputs("c5")
        kw = dict(loss_func=c["loss"], batch_size=c["batch"], learning_rate=k["learning_rate"], corr_frac=k["corr_frac"], seed=k["seed"], alpha=k["alpha"])
        cf = c["cf"]
    else:
        data, lab, W0, kw, _ = ML.config(name)
        cf = M.CFGS[name]["cf"] if name in M.CFGS else 20
    assert M.checksum(data, lab).tolist() == G["inputs_checksum"].tolist()          # the same regenerated inputs
    common = dict(model_name=name, main_dir=name, compress_factor=cf, enc_act_func="sigmoid", dec_act_func="sigmoid", loss_func=kw["loss_func"],
                  num_epochs=_epochs(case), batch_size=kw["batch_size"], opt="gradient_descent", learning_rate=kw["learning_rate"], corr_type="masking",
                  corr_frac=kw["corr_frac"], verbose=0, verbose_step=1, seed=kw["seed"], alpha=kw["alpha"], precision=precision, rng="numpy",
                  init_weights=W0, results_root=str(tmp) + "/")
    if name == "c5":
        model = DenoisingAutoencoderTriplet(**common)
        model.fit({"org": data[0], "pos": data[1], "neg": data[2]})
    else:
        model = DenoisingAutoencoder(triplet_strategy=kw["triplet_strategy"], **common)
        model.fit(data, train_set_label=lab)
    pb = np.concatenate([model.epoch_stats(e + 1)["per_batch"] for e in range(_epochs(case))])
    return model, pb, G


@pytest.mark.parametrize("name,precision,gate", CASES, ids=[f"{n}-{p}" for n, p, _ in CASES])
def test_hundred_step_curve(tmp_path, name, precision, gate):
    import make_long_curves as ML
    from dae_rnn_news_recommendation_amd import _lib as L
    path = ML.long_path(name.split("@")[0], _epochs(name))
    if not os.path.exists(path):
        pytest.skip(f"{path} not generated (python tests/golden/make_long_curves.py long {name.split('@')[0]} [epochs])")
    model, pb, G = _fit(name, precision, tmp_path)
    assert pb.shape[0] == (100 if "@" not in name else 10 * _epochs(name))
    name = name.split("@")[0]
    if precision == "auto":
        assert model.precision_used == L.auto_precision({"c1": "none", "c2": "batch_all", "c4": "batch_all", "c5": "explicit"}[name])
    worst = {}
    for col, q in ((0, "cost"), (1, "ae"), (2, "triplet")):
        g = G[q]
        if np.abs(g).max() == 0:
            continue
        d = np.abs(pb[:, col] - g) / np.abs(g)
        over = np.nonzero(d > 1e-4)[0]
        worst[q] = d.max()
        print(f"[long curve] {name} {precision} (= {model.precision_used}) {q}: max {d.max():.2e} at step {int(d.argmax()) + 1}; over the first 20 steps "
              f"{d[:20].max():.2e}; first step outside 1e-4: {'none' if len(over) == 0 else int(over[0]) + 1}")
        assert (d <= gate).all(), (name, precision, q, float(d.max()), int(d.argmax()) + 1)
    if precision == "f16x2":        # the measurement that moved the default: inside 1e-4 for the 20 steps round 5 froze, outside it before step 100
        assert max(worst.values()) > 1e-4, worst
