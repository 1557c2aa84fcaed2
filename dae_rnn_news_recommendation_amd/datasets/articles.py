"""``similar_articles`` -- reference ``datasets/articles.py:83-128``: give every article a positive (another article of
the same category) and a negative (an article of another category) for the explicit-triplet trainer
(``main_autoencoder_triplet.py:44``).  Host code (pandas + NumPy), same arguments, column names, RNG consumption and
quirks; the parquet / jieba / CountVectorizer parts of that module are out of scope.

Semantics kept from the reference:
  * categories are visited in ``value_counts()`` order, only those with ``min_cate <= count <= max_cate``;
  * the positive of an article is the NEXT article of its category in row order (the last one has none);
  * its negative is drawn with ``DataFrame.sample`` (global NumPy RNG, without replacement) from the ids outside the
    category, one draw per category, in that visiting order;
  * id 0 means "missing": ``valid_triplet_data`` is 1 only where both ids are non-zero (so an article whose positive
    or negative happens to BE id 0 is dropped -- the reference's note asks for numeric, in practice positive, ids).
"""
from __future__ import annotations

import numpy as np


def similar_articles(out_df, id_colname='article_id', cate_colname='main_category_id', min_cate=2, max_cate=None):
    id_pos_colname = id_colname + '_pos'
    id_neg_colname = id_colname + '_neg'
    counts = out_df[cate_colname].value_counts()
    upper = np.inf if max_cate is None else max_cate
    counts = counts[(counts <= upper) & (counts >= min_cate)]

    ids = out_df[id_colname].to_numpy()
    cats = out_df[cate_colname].to_numpy()
    pos = np.zeros(len(out_df), dtype=np.int64)
    neg = np.zeros(len(out_df), dtype=np.int64)
    for cate in counts.index:
        members = np.flatnonzero(cats == cate)            # row positions of the category, in row order
        with_next = members[:-1]                          # everyone but the last has a successor
        if with_next.size == 0:
            continue
        pos[with_next] = ids[members[1:]].astype(np.int64)
        outside = out_df.loc[cats != cate, id_colname]
        neg[with_next] = outside.sample(with_next.size).to_numpy().astype(np.int64)

    out_df = out_df.copy()
    out_df[id_pos_colname] = pos
    out_df[id_neg_colname] = neg
    out_df['valid_triplet_data'] = ((pos != 0) & (neg != 0)).astype(np.int64)
    return out_df
