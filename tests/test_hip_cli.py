"""GPU smoke/parity of the remaining drop-in surface: the CLI (flag-compatible main_autoencoder.py) and the
explicit-triplet estimator."""
import os

import numpy as np
import pytest
from scipy import sparse

import oracle as O

pytestmark = pytest.mark.gpu


def test_cli_end_to_end(tmp_path, monkeypatch, capsys):
    import main_autoencoder as cli
    monkeypatch.chdir(tmp_path)
    model = cli.main(["--model_name", "demo", "--num_epochs", "3", "--train_row", "600", "--validate_row", "200",
                      "--validation", "--max_features", "1000", "--verbose", "--verbose_step", "1", "--seed", "3",
                      "--triplet_strategy", "batch_hard", "--opt", "ada_grad"])
    out = capsys.readouterr().out
    assert out.count("At step") == 3 and "Triplet: Fraction=" in out and "[Validation Stat" in out
    d = model.data_dir
    assert os.path.exists(d + "article_encoded_train.npy") and os.path.exists(d + "article_encoded_validate.npy")
    emb = np.load(d + "article_encoded_train.npy")
    assert emb.shape == (600, 50) and np.isfinite(emb).all()
    assert os.path.exists(model.model_path + ".npz") and "triplet_strategy=batch_hard" in open(model.parameter_file).read()
    costs = [h["cost"] for h in model.history]
    assert costs[-1] < costs[0]                               # it trains
    # the evaluation step the reference runs next (:307-317): four similarity matrices, summarised per label
    assert "calculate similarity done" in out and out.count("AUROC") == 4
    assert "embedding (validate)" in out and "200 x 200" in out
    import json
    st = json.load(open(model.plot_dir + "similarity_boxplot_encoded.json"))     # the figure's numbers, under the figure's name
    assert 0.0 <= st["auroc"] <= 1.0 and st["n_related"] + st["n_unrelated"] == 600 * 599 // 2


def test_cli_artefacts_and_restore(tmp_path, monkeypatch, capsys):
    """Reference artefact names / formats in the model's data directory (main_autoencoder.py:227-240) and
    --restore_previous_data / --restore_previous_model (:161-174): a second run picks the data and the weights up."""
    import main_autoencoder as cli
    from dae_rnn_news_recommendation_amd import helpers
    monkeypatch.chdir(tmp_path)
    common = ["--model_name", "keep", "--train_row", "300", "--validate_row", "100", "--validation", "--max_features", "600",
              "--seed", "2", "--triplet_strategy", "batch_all", "--similarity", "false"]
    m1 = cli.main(common + ["--num_epochs", "2", "--save_tsv"])
    d = m1.data_dir
    X = helpers.read_file(d + "article_binary_count_vectorized.npz")
    assert sparse.issparse(X) and X.shape == (300, 600)
    assert helpers.read_file(d + "article_binary_count_vectorized_validate.npz").shape == (100, 600)
    lab = helpers.read_file(d + "article_label_category_publish_name.pkl", data_type="pandas_series")
    assert len(lab) == 300
    assert os.path.exists(m1.tsv_dir + "article_encoded.tsv") and os.path.exists(m1.tsv_dir + "article_label.tsv")
    e1 = np.load(d + "article_encoded_train.npy")
    w1 = m1.get_model_parameters()
    m2 = cli.main(common + ["--num_epochs", "0", "--restore_previous_data", "--restore_previous_model"])
    w2 = m2.get_model_parameters()
    assert all(np.array_equal(w1[k], w2[k]) for k in w1)                # --num_epochs 0: the restored model is not trained further
    assert np.allclose(np.load(m2.data_dir + "article_encoded_train.npy"), e1, rtol=1e-5, atol=1e-6)   # same data, same weights


def test_triplet_cli_end_to_end(tmp_path, monkeypatch, capsys):
    """main_autoencoder_triplet.py: similar_articles -> {'org','pos','neg'} -> DenoisingAutoencoderTriplet.fit -> transform ->
    device similarity (reference main_autoencoder_triplet.py:44-59, 236-290)."""
    import main_autoencoder_triplet as cli
    monkeypatch.chdir(tmp_path)
    model = cli.main(["--model_name", "demo3", "--num_epochs", "3", "--train_row", "400", "--validate_row", "100", "--validation",
                      "--max_features", "800", "--verbose", "--verbose_step", "1", "--seed", "5", "--opt", "momentum",
                      "--encode_full"])
    out = capsys.readouterr().out
    assert "similar_articles:" in out and "fit done" in out and out.count("At step") == 3
    assert "calculate similarity done" in out and out.count("AUROC") == 4
    emb = np.load(model.data_dir + "article_encoded.npy")
    assert emb.shape == (400, 40) and np.isfinite(emb).all()
    costs = [h["cost"] for h in model.history]
    assert costs[-1] < costs[0]


def test_explicit_triplet_estimator_matches_oracle(tmp_path):
    from dae_rnn_news_recommendation_amd.autoencoder.autoencoder_triplet import DenoisingAutoencoderTriplet
    rng = np.random.default_rng(0)
    N, F = 120, 400
    H = F // 10

    def mk(seed):
        m = sparse.random(N, F, density=0.05, random_state=np.random.RandomState(seed), format="csr", dtype=np.float32)
        m.data = (m.data * 0.9 + 0.1).astype(np.float32); m.sort_indices()
        return m
    data = {"org": mk(1), "pos": mk(2), "neg": mk(3)}
    W0 = rng.uniform(-0.2, 0.2, (F, H)).astype(np.float32)
    model = DenoisingAutoencoderTriplet(model_name="t3", main_dir="t3", compress_factor=10, enc_act_func="sigmoid",
                                        dec_act_func="sigmoid", loss_func="cosine_proximity", num_epochs=2, batch_size=40,
                                        learning_rate=0.05, corr_type="none", verbose=False, verbose_step=1, seed=5, alpha=2,
                                        precision="fp32", init_weights=W0, results_root=str(tmp_path) + "/")
    model.fit(data)
    # oracle: same shuffles (one legacy-RNG shuffle per epoch, shared by the three blocks)
    np.random.seed(5)
    W = W0.astype(np.float64); bh = np.zeros(H); bv = np.zeros(F)
    st = O.OptState("gradient_descent", [W.shape, bh.shape, bv.shape], np.float64)
    hist = []
    for e in range(2):
        order = O.gen_batches_index(N, 40)
        costs = []
        for idx in order:
            xs = [data[k][idx].toarray() for k in ("org", "pos", "neg")]
            r = O.explicit_triplet_forward_backward(W, bh, bv, xs, xs, loss_func="cosine_proximity", alpha=2.0, dt=np.float64)
            O.opt_apply(st, [W, bh, bv], [r["dW"], r["dbh"], r["dbv"]], 0.05, 0.5, np.float64)
            costs.append(float(r["cost"]))
        hist.append(np.mean(costs))
    for e in range(2):
        got = model.epoch_stats(e + 1)["cost"]
        assert abs(got - hist[e]) <= 1e-4 * abs(hist[e]), (e, got, hist[e])
    assert np.abs(model.engine.get_params()[0] - W).max() <= 2e-5 * np.abs(W).max()


def test_explicit_triplet_masking_noise_feeder_matches_oracle(tmp_path):
    """Explicit-triplet fit with the reference-exact masking stream: per epoch ONE rand() draw over the stacked (org, pos, neg) set,
    then ONE shuffle shared by the three blocks -- drawn one epoch ahead by the feeder thread."""
    from dae_rnn_news_recommendation_amd.autoencoder.autoencoder_triplet import DenoisingAutoencoderTriplet
    rng = np.random.default_rng(1)
    N, F = 90, 300
    H = F // 10

    def mk(seed):
        m = sparse.random(N, F, density=0.06, random_state=np.random.RandomState(seed), format="csr", dtype=np.float32)
        m.data = (m.data * 0.9 + 0.1).astype(np.float32); m.sort_indices()
        return m
    data = {"org": mk(4), "pos": mk(5), "neg": mk(6)}
    W0 = rng.uniform(-0.2, 0.2, (F, H)).astype(np.float32)
    model = DenoisingAutoencoderTriplet(model_name="t4", main_dir="t4", compress_factor=10, enc_act_func="sigmoid",
                                        dec_act_func="sigmoid", loss_func="mean_squared", num_epochs=3, batch_size=30,
                                        learning_rate=0.05, corr_type="masking", corr_frac=0.3, verbose=False, verbose_step=1, seed=8,
                                        alpha=1, precision="fp32", rng="numpy", init_weights=W0, results_root=str(tmp_path) + "/")
    model.fit(data)
    stacked = sparse.vstack([data[k] for k in ("org", "pos", "neg")]).tocsr()
    np.random.seed(8)
    W = W0.astype(np.float64); bh = np.zeros(H); bv = np.zeros(F)
    st = O.OptState("gradient_descent", [W.shape, bh.shape, bv.shape], np.float64)
    for e in range(3):
        keep = np.random.rand(stacked.nnz) >= 0.3                       # utils.masking_noise over the stacked set (storage order)
        sc = stacked.copy(); sc.data = sc.data * keep
        order = O.gen_batches_index(N, 30)
        costs = []
        for idx in order:
            xs = [stacked[k * N + np.asarray(idx)].toarray() for k in range(3)]
            xcs = [sc[k * N + np.asarray(idx)].toarray() for k in range(3)]
            r = O.explicit_triplet_forward_backward(W, bh, bv, xs, xcs, loss_func="mean_squared", alpha=1.0, dt=np.float64)
            O.opt_apply(st, [W, bh, bv], [r["dW"], r["dbh"], r["dbv"]], 0.05, 0.5, np.float64)
            costs.append(float(r["cost"]))
        got = model.epoch_stats(e + 1)["cost"]
        assert abs(got - np.mean(costs)) <= 1e-4 * abs(np.mean(costs)), (e, got, np.mean(costs))
    assert model.samples_per_sec > 0


def test_explicit_triplet_device_salt_and_pepper_runs(tmp_path):
    """corr_type='salt_and_pepper' with rng='philox' on the explicit-triplet estimator: flips drawn per batch on the device."""
    from dae_rnn_news_recommendation_amd.autoencoder.autoencoder_triplet import DenoisingAutoencoderTriplet
    N, F = 60, 300

    def mk(seed):
        m = sparse.random(N, F, density=0.06, random_state=np.random.RandomState(seed), format="csr", dtype=np.float32)
        m.data[:] = 1.0; m.sort_indices()
        return m
    data = {"org": mk(7), "pos": mk(8), "neg": mk(9)}
    costs = []
    for _ in range(2):
        model = DenoisingAutoencoderTriplet(model_name="t5", main_dir="t5", compress_factor=10, enc_act_func="sigmoid",
                                            dec_act_func="sigmoid", loss_func="cross_entropy", num_epochs=2, batch_size=20,
                                            learning_rate=0.05, corr_type="salt_and_pepper", corr_frac=0.05, verbose=False, verbose_step=1,
                                            seed=3, alpha=1, precision="fp32", rng="philox", results_root=str(tmp_path) + "/")
        model.fit(data)
        costs.append([model.epoch_stats(e + 1)["cost"] for e in range(2)])
    assert np.isfinite(costs).all() and costs[0] == costs[1]           # counter RNG: the same seed gives the same run


def test_bench_two_ranks_on_one_gpu(tmp_path):
    """The N > 1 path of bench.py (phase-1 step -> reduce-scatter of the W gradient -> sharded optimizer -> all-gather of the
    low-precision shadow, dp.ShardedExchange) with two processes
    sharing this box's single GPU (gloo collectives; RCCL needs one GPU per rank): `python bench.py --gpus 2 ...` with NO launcher around it
    (bench.py starts its own ranks), one JSON line from rank 0, n_gpus = 2, a finite loss that went down."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "12", "--warmup", "3",
           "--backend", "gloo", "--single-device", "--no-cpu-baseline", "--no-roofline", "--rows", "1600"]
    out = subprocess.run(cmd, cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 12 and d["scaling"] == "weak" and d["config"]["parallelism"] == "dp2"
    assert d["value"] > 0 and np.isfinite(d["final_losses"]["cost"])


@pytest.mark.parametrize("precision,tol", [("fp32", 3e-5), ("auto", 1e-4)])
@pytest.mark.parametrize("strategy,opt", [("batch_all", "gradient_descent"), ("none", "momentum")])
def test_cli_training_values_match_the_oracle(tmp_path, monkeypatch, capsys, precision, tol, strategy, opt):
    """The CLI's printed per-epoch losses are VALUES of the reference's arithmetic, not just "a cost that goes down": the oracle's fit loop
    (oracle.fit_reference, pinned to the reference's own fit by tests/test_golden_graph.py) on exactly what the run trained on -- the matrix and labels the CLI
    saved under the reference's artefact names (main_autoencoder.py:227-240), the Xavier draw of the same seed, the reference-exact legacy RNG stream of
    masking and shuffles -- reproduces every epoch line (mean cost / AE / triplet over the epoch's batches, autoencoder.py:283-294)."""
    import re
    import main_autoencoder as cli
    from dae_rnn_news_recommendation_amd import helpers
    from dae_rnn_news_recommendation_amd.autoencoder import utils
    monkeypatch.chdir(tmp_path)
    seed, epochs = 4, 3
    model = cli.main(["--model_name", "vals", "--num_epochs", str(epochs), "--train_row", "300", "--max_features", "600", "--verbose", "--verbose_step", "1",
                      "--seed", str(seed), "--triplet_strategy", strategy, "--opt", opt, "--similarity", "false", "--precision", precision])
    out = capsys.readouterr().out
    X = helpers.read_file(model.data_dir + "article_binary_count_vectorized.npz")
    lab = np.asarray(helpers.read_file(model.data_dir + "article_label_category_publish_name.pkl", data_type="pandas_series"))
    F = X.shape[1]; H = F // 20
    W0 = utils.xavier_init(F, H, 1, rng=np.random.RandomState(seed))
    r = O.fit_reference(X.tocsr(), lab if strategy != "none" else None, W0, enc_act="sigmoid", dec_act="sigmoid", loss_func="cross_entropy", num_epochs=epochs,
                        batch_size=0.1, opt=opt, learning_rate=0.1, momentum=0.5, corr_type="masking", corr_frac=0.3, seed=seed, alpha=1.0,
                        triplet_strategy=strategy, dt=np.float32)
    lines = re.findall(r"Overall=([0-9.]+)\s+Autoencoder=([0-9.]+)\s+Triplet=([0-9.]+)", out) if strategy != "none" else re.findall(r"Overall=([0-9.]+)()()", out)
    assert len(model.history) == epochs
    for e in range(epochs):
        want = {k: float(np.mean(r["history"][e][k])) for k in ("cost", "ae", "triplet")}
        got = model.history[e]
        assert abs(got["cost"] - want["cost"]) <= tol * abs(want["cost"]), (e, got["cost"], want["cost"])
        assert abs(got["ae"] - want["ae"]) <= tol * abs(want["ae"]), (e, got["ae"], want["ae"])
        if strategy != "none":
            assert abs(got["triplet"] - want["triplet"]) <= tol * abs(want["triplet"]), (e, got["triplet"], want["triplet"])
        if lines:                                         # ... and what the epoch line printed is that value to its four decimals
            assert abs(float(lines[e][0]) - want["cost"]) <= tol * abs(want["cost"]) + 1e-4
    p = model.get_model_parameters()
    assert np.abs(p["enc_w"] - r["W"]).max() <= (2e-4 if precision == "fp32" else 2e-3) * np.abs(r["W"]).max()


@pytest.mark.parametrize("precision,tol", [("fp32", 3e-5), ("auto", 1e-4)])
def test_triplet_cli_training_values_match_the_oracle(tmp_path, monkeypatch, capsys, precision, tol):
    """main_autoencoder_triplet.py on values: the (org, pos, neg) matrices rebuilt exactly as the CLI builds them (same seed, similar_articles on the same labels),
    then the explicit-triplet fit restated on the oracle's step -- corrupt org / pos / neg in dict order, ONE shared shuffle, fractional batch size
    (autoencoder_triplet.py:106-146, utils.py:73-91) -- reproduces every epoch's mean cost / AE / triplet loss and the final weights."""
    import argparse
    import main_autoencoder as base
    import main_autoencoder_triplet as cli
    from dae_rnn_news_recommendation_amd.autoencoder import utils
    monkeypatch.chdir(tmp_path)
    seed, epochs, rows, F = 6, 3, 300, 600
    argv = ["--model_name", "vals3", "--num_epochs", str(epochs), "--train_row", str(rows), "--max_features", str(F), "--verbose", "--verbose_step", "1",
            "--seed", str(seed), "--similarity", "false", "--precision", precision]
    model = cli.main(argv)
    a = base.validate(cli.build_parser().parse_args(argv))
    np.random.seed(seed)
    wide = argparse.Namespace(**vars(a)); wide.train_row, wide.validate_row = int(a.train_row * 1.25) + 8, int(a.validate_row * 1.25) + 8
    X, y = base.load_data(wide)
    train, _, _, _, _ = cli.build_triplets(X, y, a.train_row, a.validate_row, a.validation)
    ms = [train[k].tocsr() for k in ("org", "pos", "neg")]
    N = ms[0].shape[0]; H = F // 20
    dt = np.float32
    W = utils.xavier_init(F, H, 1, rng=np.random.RandomState(seed)).astype(dt); bh = np.zeros(H, dt); bv = np.zeros(F, dt)
    st = O.OptState("gradient_descent", [W.shape, bh.shape, bv.shape], dt)
    np.random.seed(seed)                                                    # the estimator's constructor re-seeds the legacy stream (reference :72-73)
    bs = max(round(N * 0.1), 1)
    assert len(model.history) == epochs
    for e in range(epochs):
        xcs = [O.masking_noise(m, 0.3) for m in ms]
        index = list(range(N)); np.random.shuffle(index)
        rec = dict(cost=[], ae=[], triplet=[])
        for i in range(0, N, bs):
            idx = index[i:i + bs]
            r = O.explicit_triplet_forward_backward(W, bh, bv, [m[idx].toarray() for m in ms], [x[idx].toarray() for x in xcs], loss_func="cross_entropy",
                                                    alpha=1.0, dt=dt)
            O.opt_apply(st, [W, bh, bv], [r["dW"], r["dbh"], r["dbv"]], 0.1, 0.5, dt)
            rec["cost"].append(float(r["cost"])); rec["ae"].append(float(r["ae_loss"])); rec["triplet"].append(float(r["triplet_loss"]))
        got = model.history[e]
        for k in ("cost", "ae", "triplet"):
            want = float(np.mean(rec[k]))
            assert abs(got[k] - want) <= tol * abs(want), (e, k, got[k], want)
    p = model.get_model_parameters()
    assert np.abs(p["enc_w"] - W).max() <= (2e-4 if precision == "fp32" else 2e-3) * np.abs(W).max()
