#!/usr/bin/env python3
"""Command-line trainer with the flag surface of the reference's ``main_autoencoder.py`` (:27-74) on the
MI355X-native estimator.

Kept: flag names, defaults, choices and the validation asserts (:94-109), ``main_dir`` defaulting to
``model_name`` (:111), the fit -> transform(decay-compensated) sequence (:277-290), parameter.txt, the saved
artefact names under results/<algo>/<main_dir>/data/.  Not kept (out of the hot path): the parquet /
jieba / CountVectorizer preprocessing and the matplotlib ROC plots -- the UCI dataset blob is absent from the
reference checkout (.MISSING_LARGE_BLOBS:2), so data comes from ``--data <matrix.npz> [--labels <labels.npy>]``
(scipy CSR / dense .npy) or from the seeded synthetic generator (default).

examples:
  python main_autoencoder.py --model_name demo --num_epochs 5 --verbose --verbose_step 1
  python main_autoencoder.py --model_name uci --data X.npz --labels y.npy --triplet_strategy batch_hard
"""
import argparse
import os
import sys

import numpy as np
from scipy import sparse

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def str2bool(v):
    return str(v).lower() in ("1", "true", "t", "yes", "y")


def build_parser():
    p = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    b = dict(type=str2bool, nargs="?", const=True)
    # Global configuration (reference :27-35)
    p.add_argument("--verbose", default=False, **b)
    p.add_argument("--verbose_step", type=int, default=5)
    p.add_argument("--encode_full", default=False, **b)
    p.add_argument("--validation", default=False, **b)
    p.add_argument("--input_format", default="binary", choices=["binary", "tfidf"])
    p.add_argument("--label", default="category_publish_name", choices=["category_publish_name", "story"])
    p.add_argument("--save_tsv", default=False, **b)
    p.add_argument("--train_row", type=int, default=8000)
    p.add_argument("--validate_row", type=int, default=2000)
    # Count-vectorizer parameters (:47-50); only max_features matters for synthetic data
    p.add_argument("--restore_previous_data", default=False, **b)
    p.add_argument("--min_df", type=float, default=0)
    p.add_argument("--max_df", type=float, default=0.99)
    p.add_argument("--max_features", type=int, default=10000)
    # Autoencoder parameters (:56-74)
    p.add_argument("--model_name", default="")
    p.add_argument("--restore_previous_model", default=False, **b)
    p.add_argument("--seed", type=int, default=-1)
    p.add_argument("--compress_factor", type=int, default=20)
    p.add_argument("--corr_type", default="masking", choices=["masking", "salt_and_pepper", "decay", "none"])
    p.add_argument("--corr_frac", type=float, default=0.3)
    p.add_argument("--xavier_init", type=int, default=1)
    p.add_argument("--enc_act_func", default="sigmoid", choices=["sigmoid", "tanh"])
    p.add_argument("--dec_act_func", default="sigmoid", choices=["sigmoid", "tanh", "none"])
    p.add_argument("--main_dir", default="")
    p.add_argument("--loss_func", default="cross_entropy", choices=["mean_squared", "cross_entropy", "cosine_proximity"])
    p.add_argument("--opt", default="gradient_descent", choices=["gradient_descent", "ada_grad", "momentum"])
    p.add_argument("--learning_rate", type=float, default=0.1)
    p.add_argument("--momentum", type=float, default=0.5)
    p.add_argument("--num_epochs", type=int, default=50)
    p.add_argument("--batch_size", type=float, default=0.1)
    p.add_argument("--alpha", type=float, default=1)
    p.add_argument("--triplet_strategy", default="batch_all", choices=["batch_all", "batch_hard", "none"])
    # MI355X-side additions
    p.add_argument("--precision", default="auto", choices=["auto", "f16x2h", "f16x2d", "bf16x3", "fp32", "f16x3", "f16x2", "bf16", "f16"],
                   help="auto (default): per triplet strategy the cheapest mode measured to hold the reference's loss curve within 1e-4 over 100 steps -- batch_all "
                        "/ batch_hard f16x2h, none f16x2d (fp16 MFMA operand images; W and the operands the strategy is sensitive to as hi + lo); bf16x3: split-bf16, "
                        "hi + lo images of every operand; fp32: exact-fp32 MFMA; f16x3: every operand hi + lo fp16; f16x2 (W alone hi + lo) holds 20 steps, not 100; "
                        "bf16 / f16 (single images) are faster still and outside the gate")
    p.add_argument("--rng", default="numpy", choices=["numpy", "philox"])
    p.add_argument("--data", default="", help="scipy-sparse .npz or dense .npy feature matrix (rows = articles)")
    p.add_argument("--labels", default="", help=".npy label vector aligned with --data")
    p.add_argument("--data_parallel", default=False, **b)
    p.add_argument("--similarity", default=True, **b,
                   help="after training: the N x N cosine similarities the reference computes (:307-317), on the device")
    return p


def validate(a):
    """The reference's asserts (:94-109)."""
    assert 0. <= a.min_df <= 1.
    assert 0. <= a.max_df <= 1.
    assert a.max_features >= 1
    assert 0. <= a.corr_frac <= 1.
    assert a.verbose_step > 0
    if a.input_format == 'tfidf':
        assert a.loss_func in ['mean_squared', 'cosine_proximity']
    if a.main_dir == '':
        a.main_dir = a.model_name
    assert a.model_name != '', "--model_name is required"
    return a


def load_data(a):
    from dae_rnn_news_recommendation_amd.synthetic import synthetic_csr, synthetic_labels
    n = a.train_row + (a.validate_row if a.validation else 0)
    if a.data:
        X = sparse.load_npz(a.data).tocsr() if a.data.endswith(".npz") else np.load(a.data)
        y = np.load(a.labels, allow_pickle=True) if a.labels else None
        X = X[:n]; y = None if y is None else y[:n]
    else:
        seed = a.seed if a.seed >= 0 else 1234
        X = synthetic_csr(n, a.max_features, nnz_per_row=200, seed=seed, tfidf=(a.input_format == "tfidf"))
        y = synthetic_labels(n, kind="category" if a.label == "category_publish_name" else "story", seed=seed)
    if a.input_format == "binary" and sparse.issparse(X):
        X.data[:] = 1                                                   # reference :235-236
    return X, y


def evaluate_similarity(a, trX, vlX, trY, vlY, emb, emb_v, plot_dir=None):
    """Reference :307-317 + :322-345 -- pairwise cosine similarity of the input vectors and of the embeddings (train,
    validation) and, per label, the related-vs-unrelated comparison the reference draws as ROC curve + box plot
    (helpers.visualize_pairwise_similarity).  Both run on the MI355X (dae_pairwise_similarity, dae_pair_stats); the figure's
    numbers are printed and, with ``plot_dir``, written as JSON under the reference's figure names."""
    from dae_rnn_news_recommendation_amd import helpers
    print('calculate similarity')
    metric_in = 'linear kernel' if a.input_format == 'tfidf' else 'cosine'   # TF-IDF rows are l2-normalised already (:311)
    stem_in = 'tfidf' if a.input_format == 'tfidf' else 'binary_count'
    jobs = [('input vectors (train)', trX, metric_in, trY, 'similarity_boxplot_' + stem_in),
            ('embedding (train)', emb, 'cosine', trY, 'similarity_boxplot_encoded')]
    if vlX is not None:
        jobs += [('input vectors (validate)', vlX, metric_in, vlY, 'similarity_boxplot_' + stem_in + '_validate'),
                 ('embedding (validate)', emb_v, 'cosine', vlY, 'similarity_boxplot_encoded_validate')]
    rows = []
    for name, M, metric, y, fig in jobs:
        S = helpers.pairwise_similarity(M, metric=metric, return_tensor=True)
        line = '  %-26s %5d x %-5d' % (name, S.shape[0], S.shape[1])
        if y is not None:
            ids = np.unique(np.asarray(y), return_inverse=True)[1]
            st = helpers.visualize_pairwise_similarity(ids, S, plot='boxplot', title=name,
                                                       save_path=None if plot_dir is None else plot_dir + fig + '.png')
            line += '  AUROC %.4f  median sim related %.4f  unrelated %.4f  (%d / %d pairs)' % (
                st['auroc'], st['related']['median'] if st['n_related'] else float('nan'),
                st['unrelated']['median'] if st['n_unrelated'] else float('nan'), st['n_related'], st['n_unrelated'])
            rows.append((name, st))
        print(line)
        del S
    print('calculate similarity done')
    return rows


# artefact names of the reference's data directory (main_autoencoder.py:227-244, restored at :162-174)
def _artefact(kind, a, validate=False):
    suffix = "_validate" if validate else ""
    if kind == "features":
        stem = "article_tfidf_vectorized" if a.input_format == "tfidf" else "article_binary_count_vectorized"
        return stem + suffix + ".npz"
    return "article_label_" + a.label + suffix + ".pkl"


def load_or_restore(a, data_dir, helpers):
    """Train / validation matrices and label vectors: restored from the model's data directory
    (``--restore_previous_data``, reference :161-174) or built by ``load_data`` and saved there under the reference's
    artefact names and formats (scipy .npz for the vectorised text, pickled pandas Series for the labels, :227-240)."""
    import pandas as pd
    if a.restore_previous_data:
        trX = helpers.read_file(data_dir + _artefact("features", a))
        vlX = helpers.read_file(data_dir + _artefact("features", a, True)) if a.validation else None
        trY = helpers.read_file(data_dir + _artefact("label", a), data_type='pandas_series').to_numpy()
        vlY = helpers.read_file(data_dir + _artefact("label", a, True), data_type='pandas_series').to_numpy() if a.validation else None
        return trX, vlX, trY, vlY
    X, y = load_data(a)
    trX, vlX = X[:a.train_row], (X[a.train_row:a.train_row + a.validate_row] if a.validation else None)
    trY = None if y is None else np.asarray(y[:a.train_row])
    vlY = None if (y is None or not a.validation) else np.asarray(y[a.train_row:a.train_row + a.validate_row])
    from dae_rnn_news_recommendation_amd import dp
    if dp.rank() != 0:                     # data parallel: rank 0 is the only writer of the shared artefact directory
        return trX, vlX, trY, vlY
    for M, val in ((trX, False), (vlX, True)):
        if M is not None:
            helpers.save_file(M if sparse.issparse(M) else np.asarray(M), data_dir + (_artefact("features", a, val) if sparse.issparse(M)
                              else _artefact("features", a, val).replace(".npz", ".npy")))
    for v, val in ((trY, False), (vlY, True)):
        if v is not None:
            helpers.save_file(pd.Series(v, name="label_" + a.label), data_dir + _artefact("label", a, val))
    return trX, vlX, trY, vlY


def main(argv=None):
    a = validate(build_parser().parse_args(argv))
    print(__file__ + ': Start')
    from dae_rnn_news_recommendation_amd import dp
    from dae_rnn_news_recommendation_amd.autoencoder import DenoisingAutoencoder, utils
    if a.data_parallel:
        dp.init_from_env()
    model = DenoisingAutoencoder(
        model_name=a.model_name, main_dir=a.main_dir, compress_factor=a.compress_factor, enc_act_func=a.enc_act_func,
        dec_act_func=a.dec_act_func, loss_func=a.loss_func, num_epochs=a.num_epochs, batch_size=a.batch_size,
        xavier_init=a.xavier_init, opt=a.opt, learning_rate=a.learning_rate, momentum=a.momentum, corr_type=a.corr_type,
        corr_frac=a.corr_frac, verbose=a.verbose, verbose_step=a.verbose_step, seed=a.seed, alpha=a.alpha,
        triplet_strategy=a.triplet_strategy, precision=a.precision, rng=a.rng, data_parallel=a.data_parallel)
    from dae_rnn_news_recommendation_amd import helpers
    trX, vlX, trY, vlY = load_or_restore(a, model.data_dir, helpers)
    need_labels = a.triplet_strategy != 'none'
    model.fit(trX, vlX, trY if need_labels else None, vlY if need_labels else None,
              restore_previous_model=a.restore_previous_model)
    if dp.rank() == 0:
        with open(model.parameter_file, 'a+') as fh:                    # reference :279-285
            print('train_row={}'.format(a.train_row), file=fh)
            print('validate_row={}'.format(a.validate_row), file=fh)
            print('input_format={}'.format(a.input_format), file=fh)
            print('label={}'.format(a.label), file=fh)
    # encode with the decay compensation the reference applies at inference (:289-290)
    emb = model.transform(utils.decay_noise(trX, a.corr_frac), name='article_encoded_train', save=True)
    emb_v = None
    if vlX is not None:
        emb_v = model.transform(utils.decay_noise(vlX, a.corr_frac), name='article_encoded_validate', save=True)
    if a.save_tsv and dp.rank() == 0:                                  # TensorBoard-projector exports (reference :293-303)
        import pandas as pd
        helpers.save_file(np.asarray(emb), model.tsv_dir + 'article_encoded.tsv')
        if emb_v is not None:
            helpers.save_file(np.asarray(emb_v), model.tsv_dir + 'article_encoded_validate.tsv')
        if trY is not None:
            helpers.save_file(pd.DataFrame({'label_' + a.label: trY}), model.tsv_dir + 'article_label.tsv')
        if vlY is not None:
            helpers.save_file(pd.DataFrame({'label_' + a.label: vlY}), model.tsv_dir + 'article_label_validate.tsv')
    if a.similarity and dp.rank() == 0:
        evaluate_similarity(a, trX, vlX, trY, vlY, emb, emb_v, model.plot_dir)
    if model.samples_per_sec:
        print('training throughput: %.0f samples/s over %d epochs; embeddings %s -> %s' %
              (model.samples_per_sec, a.num_epochs, emb.shape, model.data_dir))
    print(__file__ + ': End')
    return model


if __name__ == '__main__':
    main()
