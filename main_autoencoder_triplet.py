#!/usr/bin/env python3
"""Explicit-triplet trainer CLI -- flag-compatible counterpart of the reference's ``main_autoencoder_triplet.py``
(flags :17-53, triplet construction :44-59, fit -> transform -> similarity :236-290) on the MI355X kernels.

The reference builds (article, positive, negative) texts from a private parquet file, tokenises them with jieba and
vectorises with CountVectorizer; that host text pipeline is out of scope.  Here the vectorised matrix comes from
``--data`` (scipy .npz / .npy) + ``--labels`` (.npy) or from the synthetic generator, and the triplets are mapped by
``datasets.articles.similar_articles`` exactly as :44 does (label column = the category).  Everything after that point --
``DenoisingAutoencoderTriplet.fit({'org','pos','neg'})``, the decay-compensated ``transform``, the pairwise cosine
similarities -- runs on the GPU."""
from __future__ import annotations

import numpy as np
from scipy import sparse

import main_autoencoder as base


def build_parser():
    p = base.build_parser()
    p.description = __doc__
    p.set_defaults(triplet_strategy="none")             # the triplets are explicit: no in-batch mining (reference has no such flag)
    return p


def build_triplets(X, y, train_row, validate_row, validation):
    """article ids 1..n (0 means "missing" in similar_articles), category = label; keep the valid rows, then slice
    train / validation as the reference does (:46-59)."""
    import pandas as pd
    from dae_rnn_news_recommendation_amd.datasets import similar_articles
    n = X.shape[0]
    df = similar_articles(pd.DataFrame({"article_id": np.arange(1, n + 1), "label": np.asarray(y)[:n]}),
                          id_colname="article_id", cate_colname="label", min_cate=2)
    v = df[df.valid_triplet_data == 1]
    rows = v.article_id.to_numpy() - 1
    pos = v.article_id_pos.to_numpy() - 1
    neg = v.article_id_neg.to_numpy() - 1

    def block(lo, hi):
        s = slice(lo, hi)
        return {"org": X[rows[s]], "pos": X[pos[s]], "neg": X[neg[s]]}, np.asarray(y)[rows[s]]
    train, ytr = block(0, train_row)
    val, yvl = (block(train_row, train_row + validate_row) if validation else (None, None))
    return train, ytr, val, yvl, len(rows)


def main(argv=None):
    a = base.validate(build_parser().parse_args(argv))
    print(__file__ + ': Start')
    from dae_rnn_news_recommendation_amd.autoencoder import utils
    from dae_rnn_news_recommendation_amd.autoencoder.autoencoder_triplet import DenoisingAutoencoderTriplet
    if a.seed >= 0:
        np.random.seed(a.seed)                            # similar_articles samples negatives from the global NumPy RNG
    import argparse
    wide = argparse.Namespace(**vars(a))                  # rows without a positive / negative are dropped: load with head-room
    wide.train_row, wide.validate_row = int(a.train_row * 1.25) + 8, int(a.validate_row * 1.25) + 8
    X, y = base.load_data(wide)
    assert y is not None, "explicit triplets need --labels (or the synthetic generator)"
    train, ytr, val, yvl, n_valid = build_triplets(X, y, a.train_row, a.validate_row, a.validation)
    print('similar_articles: %d of %d rows have a positive and a negative; train %d, validation %d' %
          (n_valid, X.shape[0], train["org"].shape[0], 0 if val is None else val["org"].shape[0]))
    assert train["org"].shape[0] > 0, "no valid triplets"
    model = DenoisingAutoencoderTriplet(
        model_name=a.model_name, main_dir=a.main_dir, compress_factor=a.compress_factor, enc_act_func=a.enc_act_func,
        dec_act_func=a.dec_act_func, loss_func=a.loss_func, num_epochs=a.num_epochs, batch_size=a.batch_size,
        xavier_init=a.xavier_init, opt=a.opt, learning_rate=a.learning_rate, momentum=a.momentum, corr_type=a.corr_type,
        corr_frac=a.corr_frac, verbose=a.verbose, verbose_step=a.verbose_step, seed=a.seed, alpha=a.alpha,
        precision=a.precision, rng=a.rng)
    print('fit')
    model.fit(train_set=train, validation_set=val, restore_previous_model=a.restore_previous_model)
    with open(model.parameter_file, 'a+') as fh:                        # reference :241-245
        print('train_row={}'.format(a.train_row), file=fh)
        print('validate_row={}'.format(a.validate_row), file=fh)
        print('input_format={}'.format(a.input_format), file=fh)
        print('label={}'.format(a.label), file=fh)
    print('fit done')
    emb = model.transform(utils.decay_noise(train["org"], a.corr_frac), name='article_encoded', save=a.encode_full)
    emb_v = None
    if val is not None:
        emb_v = model.transform(utils.decay_noise(val["org"], a.corr_frac), name='article_encoded_validate', save=a.encode_full)
    if a.similarity:
        base.evaluate_similarity(a, train["org"], None if val is None else val["org"], ytr, yvl, emb, emb_v, model.plot_dir)
    if model.samples_per_sec:
        print('training throughput: %.0f rows/s (org+pos+neg) over %d epochs' % (model.samples_per_sec, a.num_epochs))
    print(__file__ + ': End')
    return model


if __name__ == '__main__':
    main()
