"""NumPy restatement of the reference DAE training step (TEST INFRASTRUCTURE ONLY).

Every function cites the reference file:line it follows (paths relative to the
reference repo root).  ``dt`` selects the arithmetic type: ``np.float32`` mimics the
TF-CPU fp32 graph, ``np.float64`` is the "truth" used for gradient checks.

Parity status: see ``oracle/__init__.py`` -- miners/losses/noise/batching are pinned
by golden vectors generated from the reference's own Python files; the encode/decode
/gradient/optimizer restatement is parity-UNPINNED against TensorFlow 1.12 itself
(it cannot run here) and is cross-checked against torch-CPU autograd + fp64 finite
differences instead.
"""
from __future__ import annotations

import numpy as np
from scipy import sparse

__all__ = [
    "act", "act_grad_from_output", "softplus_tf", "sigmoid",
    "get_anchor_positive_triplet_mask", "get_anchor_negative_triplet_mask", "get_triplet_mask",
    "batch_all_triplet_loss", "batch_all_triplet_loss_loops", "batch_all_closed_form",
    "batch_hard_triplet_loss", "batch_hard_triplet_loss_loops",
    "weighted_loss", "weighted_loss_rows",
    "encode", "decode", "forward_backward", "explicit_triplet_forward_backward",
    "OptState", "opt_apply",
    "masking_noise", "salt_and_pepper_noise", "decay_noise", "gen_batches_index",
    "get_sparse_ind_val_shape", "xavier_bound", "epoch_plan", "fit_reference",
    "philox4x32", "philox_uniform", "philox_uniform_dense", "salt_and_pepper_philox", "pairwise_similarity", "pair_stats",
]

EPS = 1e-16


# --------------------------------------------------------------------------- #
# activations (autoencoder/autoencoder.py:380-389, 398-411)
# --------------------------------------------------------------------------- #
def sigmoid(z):
    """tf.nn.sigmoid; numerically stable form (TF/Eigen result agrees to a few ulp)."""
    z = np.asarray(z)
    out = np.empty_like(z)
    pos = z >= 0
    out[pos] = 1.0 / (1.0 + np.exp(-z[pos]))
    ez = np.exp(z[~pos])
    out[~pos] = ez / (1.0 + ez)
    return out


def act(name, z):
    """Activation chosen at autoencoder.py:380-387 / :398-408 ('none'/anything else = identity)."""
    if name == "sigmoid":
        return sigmoid(z)
    if name == "tanh":
        return np.tanh(z)
    return np.asarray(z).copy()


def act_grad_from_output(name, a):
    """d act / dz written in terms of the activation output a (TF SigmoidGrad / TanhGrad)."""
    if name == "sigmoid":
        return a * (1.0 - a)
    if name == "tanh":
        return 1.0 - a * a
    return np.ones_like(a)


def softplus_tf(x):
    """tf.nn.softplus as TF 1.12's Eigen functor computes it (threshold = log(eps)+2):
    x < thr -> exp(x); x > -thr -> x; else log1p(exp(x)).
    ``-tf.log_sigmoid(-t)`` (triplet_loss_utils.py:126,256) == softplus(t)."""
    x = np.asarray(x)
    thr = np.log(np.finfo(x.dtype).eps) + 2.0
    ex = np.exp(np.minimum(x, -thr))  # clamp: avoid overflow in the branch not taken
    return np.where(x > -thr, x, np.where(x < thr, ex, np.log1p(ex))).astype(x.dtype)


# --------------------------------------------------------------------------- #
# masks (autoencoder/triplet_loss_utils.py:6-76)
# --------------------------------------------------------------------------- #
def get_anchor_positive_triplet_mask(labels):
    """triplet_loss_utils.py:6-26: mask[a,p] = a != p and labels[a] == labels[p]."""
    labels = np.asarray(labels)
    eq = labels[None, :] == labels[:, None]
    return eq & ~np.eye(labels.shape[0], dtype=bool)


def get_anchor_negative_triplet_mask(labels):
    """triplet_loss_utils.py:29-44: mask[a,n] = labels[a] != labels[n]."""
    labels = np.asarray(labels)
    return ~(labels[None, :] == labels[:, None])


def get_triplet_mask(labels):
    """triplet_loss_utils.py:47-76: mask[i,j,k] = i,j,k distinct and lab_i==lab_j and lab_i!=lab_k."""
    labels = np.asarray(labels)
    n = labels.shape[0]
    ne = ~np.eye(n, dtype=bool)
    distinct = ne[:, :, None] & ne[:, None, :] & ne[None, :, :]
    eq = labels[None, :] == labels[:, None]
    valid = eq[:, :, None] & ~eq[:, None, :]
    return distinct & valid


# --------------------------------------------------------------------------- #
# batch_all miner (autoencoder/triplet_loss_utils.py:79-131)
# --------------------------------------------------------------------------- #
def batch_all_closed_form(labels):
    """Integer-exact N_valid and data_weight (pos_triplets_only=False) from the label
    histogram.  Equivalent to summing _get_triplet_mask."""
    labels = np.asarray(labels)
    B = labels.shape[0]
    _, inv, cnt = np.unique(labels, return_inverse=True, return_counts=True)
    n_i = cnt[inv].astype(np.int64)
    S = int(np.sum(cnt.astype(np.int64) * (cnt.astype(np.int64) - 1)))
    n_valid = int(np.sum((n_i - 1) * (B - n_i)))
    dw = 2 * (n_i - 1) * (B - n_i) + (S - n_i * (n_i - 1))
    return n_valid, dw.astype(np.int64)


def batch_all_triplet_loss(labels, h, pos_triplets_only=False, dt=np.float32, chunk=32,
                           return_grad=False, D=None):
    """triplet_loss_utils.py:79-131, evaluated anchor-chunk by anchor-chunk (arithmetically the
    same sums as the B^3 broadcast; chunking only bounds memory).

    Returns (loss, data_weight[B], fraction, num_pos) and, if ``return_grad``, also
    G = d loss / d D (gradient wrt the Gram matrix; masks carry no gradient)."""
    labels = np.asarray(labels)
    h = np.asarray(h, dtype=dt)
    B = h.shape[0]
    if D is None:
        D = h @ h.T                                        # :93
    D = np.asarray(D, dtype=dt)
    ap = get_anchor_positive_triplet_mask(labels)          # valid[i,j,k] = ap[i,j] & an[i,k]
    an = get_anchor_negative_triplet_mask(labels)
    num_valid = 0
    num_pos = 0
    loss_sum = 0.0                                         # python float == fp64 accumulator
    dw_valid = np.zeros(B, np.int64)
    dw_pos = np.zeros(B, np.int64)
    G = np.zeros((B, B), np.float64) if return_grad else None
    # pass 1: counts (needed before the gradient can be normalised)
    blocks = []
    for i0 in range(0, B, chunk):
        i1 = min(B, i0 + chunk)
        T = (-D[i0:i1, :, None] + D[i0:i1, None, :]).astype(dt)           # :96-106
        valid = ap[i0:i1, :, None] & an[i0:i1, None, :]                   # :110
        pos = (np.where(valid, T, dt(0)) > dt(EPS))                       # :114 (valid*T > 1e-16)
        num_valid += int(valid.sum())
        num_pos += int(pos.sum())
        mask = pos if pos_triplets_only else valid                        # :118-123
        sp = softplus_tf(T)
        loss_sum += float(np.sum(np.where(mask, sp, dt(0)), dtype=np.float64))   # :126
        m = mask
        dwa = m.sum(axis=(1, 2))
        dw = dw_pos if pos_triplets_only else dw_valid
        dw[i0:i1] += dwa                                                  # anchor role  :129 sum[1,2]
        dw += m.sum(axis=(0, 1))                                          # negative role sum[0,1]
        dw += m.sum(axis=(0, 2))                                          # positive role sum[0,2]
        if return_grad:
            blocks.append((i0, i1))
    num_triplet = num_pos if pos_triplets_only else num_valid
    loss = dt(loss_sum) / (dt(num_triplet) + dt(EPS))                     # :127
    data_weight = (dw_pos if pos_triplets_only else dw_valid).astype(dt)
    frac = dt(num_pos) / (dt(num_valid) + dt(EPS))
    if not return_grad:
        return loss, data_weight, frac, dt(num_pos)
    scale = 1.0 / (float(num_triplet) + EPS)
    for (i0, i1) in blocks:
        T = (-D[i0:i1, :, None] + D[i0:i1, None, :]).astype(dt)
        valid = ap[i0:i1, :, None] & an[i0:i1, None, :]
        if pos_triplets_only:
            mask = (np.where(valid, T, dt(0)) > dt(EPS))
        else:
            mask = valid
        sg = np.where(mask, sigmoid(T.astype(np.float64)), 0.0)           # SoftplusGrad = sigmoid
        G[i0:i1, :] += sg.sum(axis=1) * scale                             # d/dD[i,k] (negative role)
        G[i0:i1, :] -= sg.sum(axis=2) * scale                             # d/dD[i,j] (positive role)
    return loss, data_weight, frac, dt(num_pos), G.astype(dt)


def batch_all_triplet_loss_loops(labels, h, dt=np.float64):
    """The triple loop of the reference's own test (autoencoder/tests/test_triplet_loss_utils.py
    :95-119) -- the executable spec.  Returns dict with valid/pos variants."""
    labels = np.asarray(labels)
    h = np.asarray(h, dtype=dt)
    n = h.shape[0]
    D = h @ h.T
    dw_v = np.zeros(n); dw_p = np.zeros(n)
    lv = 0.0; lp = 0.0; nv = 0; npos = 0
    for i in range(n):
        for j in range(n):
            for k in range(n):
                if i == j or j == k or i == k:
                    continue
                if labels[i] == labels[j] and labels[i] != labels[k]:
                    dist = D[i, k] - D[i, j]
                    loss = np.log1p(np.exp(dist))
                    dw_v[[i, j, k]] += 1; lv += loss; nv += 1
                    if dist > 1e-16:
                        dw_p[[i, j, k]] += 1; lp += loss; npos += 1
    return dict(loss_valid=lv / (nv + 1e-16), loss_pos=lp / (npos + 1e-16), dw_valid=dw_v, dw_pos=dw_p,
                num_valid=nv, num_pos=npos)


# --------------------------------------------------------------------------- #
# batch_hard miner (autoencoder/triplet_loss_utils.py:202-259)
# --------------------------------------------------------------------------- #
def batch_hard_triplet_loss(labels, h, dt=np.float32, return_grad=False, D=None):
    """triplet_loss_utils.py:202-259, literal (including the quirks listed in SURVEY 8 a15:
    invalid negatives become 0 not -inf; invalid positives are shifted by the row max; data_weight
    uses float equality against D).  If ``return_grad`` also returns G = d loss / d D following
    TF's autodiff rules (reduce_min/max split the gradient equally among ties; masks and counts
    carry no gradient; the row-max shift does)."""
    labels = np.asarray(labels)
    h = np.asarray(h, dtype=dt)
    B = h.shape[0]
    if D is None:
        D = h @ h.T                                                        # :219
    D = np.asarray(D, dtype=dt)
    ap = get_anchor_positive_triplet_mask(labels).astype(dt)               # :223-224
    rowmax = D.max(axis=1, keepdims=True)                                  # :227
    apd = D + rowmax * (dt(1.0) - ap)                                      # :228
    hp = apd.min(axis=1, keepdims=True)                                    # :231
    an = get_anchor_negative_triplet_mask(labels).astype(dt)               # :236-237
    and_ = an * D                                                          # :240
    hn = and_.max(axis=1, keepdims=True)                                   # :243
    dist = np.maximum(hn - hp, dt(0.0))                                    # :247
    cnt = (dist > dt(0.0)).astype(dt)                                      # :249
    data_weight = (cnt[:, 0]
                   + np.sum(cnt * (D == hp).astype(dt), axis=0)
                   + np.sum(cnt * (D == hn).astype(dt), axis=0))           # :251-253
    tl = softplus_tf(dist) * cnt                                           # :256
    ncnt = cnt.sum(dtype=np.float64)
    loss = dt(tl.sum(dtype=np.float64)) / (dt(ncnt) + dt(EPS))             # :257
    frac = dt(ncnt) / dt(B)
    if not return_grad:
        return loss, data_weight.astype(dt), frac, dt(ncnt)
    gd = sigmoid(dist.astype(np.float64)) * cnt / (float(ncnt) + EPS)      # [B,1]
    # max(hn-hp, 0): TF routes the gradient to x where x >= 0; cnt already zeroes the rest.
    ind_n = (and_ == hn).astype(np.float64)
    d_and = gd * ind_n / ind_n.sum(axis=1, keepdims=True)
    G = an.astype(np.float64) * d_and
    ind_p = (apd == hp).astype(np.float64)
    d_apd = -gd * ind_p / ind_p.sum(axis=1, keepdims=True)
    G += d_apd
    d_rowmax = np.sum(d_apd * (1.0 - ap.astype(np.float64)), axis=1, keepdims=True)
    ind_m = (D == rowmax).astype(np.float64)
    G += d_rowmax * ind_m / ind_m.sum(axis=1, keepdims=True)
    return loss, data_weight.astype(dt), frac, dt(ncnt), G.astype(dt)


def batch_hard_triplet_loss_loops(labels, h, dt=np.float64):
    """Loop body of the reference's own test (test_triplet_loss_utils.py:160-196); only valid for
    non-negative dot products (the regime that test exercises)."""
    labels = np.asarray(labels)
    h = np.asarray(h, dtype=dt)
    n = h.shape[0]
    D = h @ h.T
    hp = np.full(n, np.nan); hpi = np.full(n, np.nan); hn = np.full(n, np.nan); hni = np.full(n, np.nan)
    for i in range(n):
        for j in range(n):
            if i == j:
                continue
            if labels[i] == labels[j]:
                if np.isnan(hp[i]) or D[i, j] < hp[i]:
                    hp[i] = D[i, j]; hpi[i] = j
            else:
                if np.isnan(hn[i]) or D[i, j] > hn[i]:
                    hn[i] = D[i, j]; hni[i] = j
    dw = np.zeros(n); loss = 0.0; num = 0
    dist = hn - hp
    for idx, val in enumerate(dist):
        if val > 0:
            dw[idx] += 1; dw[int(hpi[idx])] += 1; dw[int(hni[idx])] += 1
            loss += np.log1p(np.exp(val)); num += 1
    return dict(loss=loss / (num + 1e-16), data_weight=dw, fraction=num / n, num=num)


# --------------------------------------------------------------------------- #
# weighted reconstruction loss (autoencoder/triplet_loss_utils.py:262-277)
# --------------------------------------------------------------------------- #
def _to_dense(x, dt):
    if sparse.issparse(x):
        return np.asarray(x.toarray(), dtype=dt)                           # :264 tf.sparse.to_dense
    return np.asarray(x, dtype=dt)


def weighted_loss_rows(x, y, loss_func, dt=np.float32):
    """Per-row loss before weighting (triplet_loss_utils.py:268-273)."""
    x = _to_dense(x, dt); y = np.asarray(y, dtype=dt)
    one = dt(1.0); eps = dt(EPS)
    if loss_func == "cross_entropy":
        return -np.sum(x * np.log(y + eps) + (one - x) * np.log(one - y + eps), axis=1)   # :269
    if loss_func == "mean_squared":
        return np.sum((x - y) ** 2, axis=1)                                               # :271
    if loss_func == "cosine_proximity":
        xn = x / np.sqrt(np.maximum(np.sum(x * x, axis=1, keepdims=True), dt(1e-12)))     # tf.nn.l2_normalize
        yn = y / np.sqrt(np.maximum(np.sum(y * y, axis=1, keepdims=True), dt(1e-12)))
        return -np.sum(xn * yn, axis=1)                                                   # :273
    raise ValueError(loss_func)


def weighted_loss(x, y, loss_func="cross_entropy", weight=None, dt=np.float32):
    """triplet_loss_utils.py:262-277: sum(row*w)/(sum(w)+1e-16); w defaults to ones[B] (:266)."""
    rows = weighted_loss_rows(x, y, loss_func, dt)
    w = np.ones(rows.shape[0], dt) if weight is None else np.asarray(weight, dtype=dt)
    return dt(np.sum(rows * w, dtype=dt) / (np.sum(w, dtype=dt) + dt(EPS)))


def _loss_dy(x, y, loss_func, dt):
    """d rowloss_i / d y_if (before the c_i = w_i/(sum w + eps) factor)."""
    one = dt(1.0); eps = dt(EPS)
    if loss_func == "cross_entropy":
        # TF differentiates log(y+eps) and log(1-y+eps) separately (no sigmoid/CE fusion).
        return -(x / (y + eps) - (one - x) / ((one - y) + eps))
    if loss_func == "mean_squared":
        return dt(-2.0) * (x - y)
    if loss_func == "cosine_proximity":
        sx = np.maximum(np.sum(x * x, axis=1, keepdims=True), dt(1e-12))
        xn = x / np.sqrt(sx)
        sy = np.sum(y * y, axis=1, keepdims=True)
        big = sy >= dt(1e-12)                      # tf.maximum routes grad to sumsq when sumsq >= eps
        s = one / np.sqrt(np.maximum(sy, dt(1e-12)))
        dot = np.sum(xn * y, axis=1, keepdims=True)
        return -(xn * s - np.where(big, dot * s * s * s, dt(0.0)) * y)
    raise ValueError(loss_func)


# --------------------------------------------------------------------------- #
# encode / decode (autoencoder/autoencoder.py:371-415)
# --------------------------------------------------------------------------- #
def encode(x_corr, W, bh, enc_act="sigmoid", dt=np.float32):
    """autoencoder.py:389: act(x~ W + bh) - act(bh)."""
    W = np.asarray(W, dt); bh = np.asarray(bh, dt)
    if sparse.issparse(x_corr):
        z1 = np.asarray(x_corr.astype(dt) @ W, dtype=dt) + bh
    else:
        z1 = np.asarray(x_corr, dt) @ W + bh
    a1 = act(enc_act, z1)
    return (a1 - act(enc_act, bh)).astype(dt), a1.astype(dt)


def decode(h, W, bv, dec_act="sigmoid", dt=np.float32):
    """autoencoder.py:411: act(h W^T + bv)."""
    return act(dec_act, np.asarray(h, dt) @ np.asarray(W, dt).T + np.asarray(bv, dt)).astype(dt)


def forward_backward(W, bh, bv, x, x_corr, labels=None, *, enc_act="sigmoid", dec_act="sigmoid",
                     loss_func="cross_entropy", triplet_strategy="none", alpha=1.0, dt=np.float32,
                     want_grads=True):
    """One training step's forward + hand-derived backward, i.e. what
    ``tf_session.run([train_step, ...])`` (autoencoder.py:233) evaluates, minus the optimizer.

    Returns dict(cost, ae_loss, triplet_loss, fraction, num, data_weight, h, y, D, G,
                 dW, dbh, dbv, delta1, delta2, dh)."""
    W = np.asarray(W, dt); bh = np.asarray(bh, dt); bv = np.asarray(bv, dt)
    xd = _to_dense(x, dt); xc = _to_dense(x_corr, dt)
    B = xd.shape[0]
    h, a1 = encode(xc, W, bh, enc_act, dt)
    z2 = h @ W.T + bv
    y = act(dec_act, z2).astype(dt)
    out = dict(h=h, y=y)
    G = None
    if triplet_strategy != "none":                                         # autoencoder.py:424-438
        D = (h @ h.T).astype(dt)
        if triplet_strategy == "batch_all":
            tl, dw, frac, num, G = batch_all_triplet_loss(labels, h, False, dt, return_grad=True, D=D)
        elif triplet_strategy == "batch_hard":
            tl, dw, frac, num, G = batch_hard_triplet_loss(labels, h, dt, return_grad=True, D=D)
        else:
            raise ValueError(triplet_strategy)
        ae = weighted_loss(xd, y, loss_func, dw, dt)                       # :433
        cost = dt(ae + dt(alpha) * tl)                                     # :438
        w = dw
        out.update(D=D, G=G)
    else:                                                                  # :441
        ae = weighted_loss(xd, y, loss_func, None, dt)
        cost = ae; tl = dt(0); frac = dt(0); num = dt(0)
        w = np.ones(B, dt)
    out.update(cost=cost, ae_loss=ae, triplet_loss=tl, fraction=frac, num=num, data_weight=np.asarray(w, dt))
    if not want_grads:
        return out
    c = (np.asarray(w, dt) / (np.sum(np.asarray(w, dt), dtype=dt) + dt(EPS)))[:, None]
    dy = c * _loss_dy(xd, y, loss_func, dt)
    delta2 = (dy * act_grad_from_output(dec_act, y)).astype(dt)            # dL/dz2
    dbv = delta2.sum(axis=0)
    dh = delta2 @ W
    if G is not None:
        dh = dh + dt(alpha) * ((G + G.T) @ h)
    delta1 = (dh * act_grad_from_output(enc_act, a1)).astype(dt)           # dL/dz1
    dW = xc.T @ delta1 + delta2.T @ h                                      # tied weights
    ab = act(enc_act, bh)
    dbh = delta1.sum(axis=0) - act_grad_from_output(enc_act, ab) * dh.sum(axis=0)   # the -act(bh) term
    out.update(dW=dW.astype(dt), dbh=dbh.astype(dt), dbv=dbv.astype(dt), delta1=delta1, delta2=delta2,
               dh=dh.astype(dt))
    return out


def explicit_triplet_forward_backward(W, bh, bv, xs, xcs, *, enc_act="sigmoid", dec_act="sigmoid",
                                      loss_func="cross_entropy", alpha=1.0, dt=np.float32):
    """DenoisingAutoencoderTriplet cost (autoencoder/autoencoder_triplet.py:256-258,286-288,303-314):
    three row-blocks (org,pos,neg) through the same W; AE loss = sum of three unweighted row means;
    triplet = mean softplus(h.h_neg - h.h_pos).  xs / xcs: lists [org,pos,neg] (clean / corrupted)."""
    W = np.asarray(W, dt); bh = np.asarray(bh, dt); bv = np.asarray(bv, dt)
    hs, a1s, ys, xds, xcd = [], [], [], [], []
    for x, xc in zip(xs, xcs):
        xd = _to_dense(x, dt); xc_ = _to_dense(xc, dt)
        h, a1 = encode(xc_, W, bh, enc_act, dt)
        y = decode(h, W, bv, dec_act, dt)
        hs.append(h); a1s.append(a1); ys.append(y); xds.append(xd); xcd.append(xc_)
    B = hs[0].shape[0]
    ae = dt(sum(weighted_loss(xd, y, loss_func, None, dt) for xd, y in zip(xds, ys)))   # :303-305
    t = np.sum(hs[0] * hs[2] - hs[0] * hs[1], axis=1)                                    # :308-311
    tl = dt(np.mean(softplus_tf(t)))
    cost = dt(ae + dt(alpha) * tl)                                                       # :314
    # backward
    g_t = (dt(alpha) * sigmoid(t.astype(np.float64)) / B).astype(dt)[:, None]
    dhs_tri = [g_t * (hs[2] - hs[1]), -g_t * hs[0], g_t * hs[0]]
    dW = np.zeros_like(W); dbh = np.zeros_like(bh); dbv = np.zeros_like(bv)
    ab = act(enc_act, bh)
    for b in range(3):
        c = dt(1.0) / (dt(B) + dt(EPS))
        delta2 = (c * _loss_dy(xds[b], ys[b], loss_func, dt) * act_grad_from_output(dec_act, ys[b])).astype(dt)
        dh = delta2 @ W + dhs_tri[b]
        delta1 = dh * act_grad_from_output(enc_act, a1s[b])
        dW += xcd[b].T @ delta1 + delta2.T @ hs[b]
        dbv += delta2.sum(axis=0)
        dbh += delta1.sum(axis=0) - act_grad_from_output(enc_act, ab) * dh.sum(axis=0)
    return dict(cost=cost, ae_loss=ae, triplet_loss=tl, dW=dW, dbh=dbh, dbv=dbv, hs=hs, ys=ys)


# --------------------------------------------------------------------------- #
# optimizers (autoencoder/autoencoder.py:444-477; TF 1.12 tf.train.* semantics)
# --------------------------------------------------------------------------- #
class OptState:
    """Slot variables of tf.train.{GradientDescent,Adagrad,Momentum,Adam}Optimizer."""

    def __init__(self, opt, shapes, dt=np.float32):
        self.opt = opt
        self.t = 0
        if opt == "ada_grad":
            self.acc = [np.full(s, 0.1, dt) for s in shapes]    # initial_accumulator_value=0.1
        elif opt == "momentum":
            self.acc = [np.zeros(s, dt) for s in shapes]
        elif opt == "adam":
            self.m = [np.zeros(s, dt) for s in shapes]
            self.v = [np.zeros(s, dt) for s in shapes]


def opt_apply(state, params, grads, lr, momentum=0.5, dt=np.float32):
    """In-place parameter update.  gradient_descent :452, ada_grad :466, momentum :469, adam :472."""
    lr = dt(lr)
    if state.opt == "gradient_descent":
        for p, g in zip(params, grads):
            p -= lr * g
    elif state.opt == "ada_grad":
        for p, g, a in zip(params, grads, state.acc):
            a += g * g
            p -= lr * g / np.sqrt(a)
    elif state.opt == "momentum":
        for p, g, a in zip(params, grads, state.acc):
            a *= dt(momentum); a += g
            p -= lr * a
    elif state.opt == "adam":
        state.t += 1
        b1, b2, eps = 0.9, 0.999, 1e-8
        lr_t = dt(float(lr) * np.sqrt(1.0 - b2 ** state.t) / (1.0 - b1 ** state.t))
        for p, g, m, v in zip(params, grads, state.m, state.v):
            m *= dt(b1); m += dt(1 - b1) * g
            v *= dt(b2); v += dt(1 - b2) * g * g
            p -= lr_t * m / (np.sqrt(v) + dt(eps))
    else:
        raise ValueError(state.opt)


# --------------------------------------------------------------------------- #
# host-side noise / batching (autoencoder/utils.py) -- NumPy legacy global RandomState
# --------------------------------------------------------------------------- #
def masking_noise(X, v):
    """utils.py:94-115.  sparse: keep stored entry e iff rand(nnz)[e] >= v, draw order = COO order of
    X.tocoo() (= CSR storage order for a CSR input); dense: np.random.choice([0,1], p=[v,1-v])."""
    assert 0.0 <= v <= 1.0
    if isinstance(X, np.ndarray):
        mask = np.random.choice(a=[0, 1], size=X.shape, p=[v, 1 - v])
        return mask * X
    Xc = X.tocoo(True)
    keep = np.random.rand(Xc.nnz) >= v
    return sparse.coo_matrix((Xc.data[keep], (Xc.row[keep], Xc.col[keep])), shape=Xc.shape).tocsr()


def salt_and_pepper_noise(X, v):
    """utils.py:118-144: per row, v column ids with replacement (randint), each set to global min or
    max by a fair coin (one np.random.random() per id, later writes win)."""
    dense = isinstance(X, np.ndarray)
    Xn = X.copy() if dense else X.tolil(True)
    n_features = X.shape[1]
    mn = X.min(); mx = X.max()
    for i in range(X.shape[0]):
        cols = np.random.randint(0, n_features, v)
        coins = np.random.random(len(cols)) if len(cols) else np.zeros(0)   # == len(cols) scalar draws
        for m, c in zip(cols, coins):
            Xn[i, m] = mn if c < 0.5 else mx
    return Xn if dense else Xn.tocsr()


def decay_noise(X, v):
    """utils.py:147-159: X * (1 - v)."""
    return X.copy() * (1.0 - v)


def gen_batches_index(n_rows, batch_size, random=True):
    """Index lists that utils.gen_batches (:29-70) yields slices for: fractional batch_size ->
    max(round(N*bs),1) (:47), int() (:48), np.random.shuffle(list(range(N))) (:50-51)."""
    assert batch_size > 0.0
    if batch_size < 1.0:
        batch_size = max(round(n_rows * batch_size), 1)
    batch_size = int(batch_size)
    index = list(range(0, n_rows))
    if random:
        np.random.shuffle(index)
    return [index[i:i + batch_size] for i in range(0, n_rows, batch_size)]


def get_sparse_ind_val_shape(m):
    """utils.py:162-180."""
    if not isinstance(m, sparse.csr_matrix):
        m = sparse.csr_matrix(m)
    m.sort_indices()
    coo = sparse.coo_matrix(m)
    return np.column_stack((coo.row, coo.col)), coo.data, coo.shape


def xavier_bound(fan_in, fan_out, const=1):
    """utils.py:24-25 (the uniform draw itself is TF-seeded and not reproducible without TF)."""
    return const * np.sqrt(6.0 / (fan_in + fan_out))


def epoch_plan(train_set, corr_type, corr_frac, batch_size):
    """RNG-order-exact restatement of one epoch's host work (autoencoder.py:218-220):
    (1) corrupt the WHOLE set, (2) shuffle + slice.  Returns (x_corrupted, [index lists])."""
    if corr_type == "masking":
        xc = masking_noise(train_set, corr_frac)
    elif corr_type == "salt_and_pepper":
        xc = salt_and_pepper_noise(train_set, int(np.round(corr_frac * train_set.shape[1])))   # :187
    elif corr_type == "decay":
        xc = decay_noise(train_set, corr_frac)
    elif corr_type == "none":
        xc = train_set
    else:
        raise ValueError(corr_type)
    return xc, gen_batches_index(train_set.shape[0], batch_size)


def fit_reference(train_set, labels, W0, *, compress_factor=None, enc_act="sigmoid", dec_act="sigmoid",
                  loss_func="cross_entropy", num_epochs=1, batch_size=0.1, opt="gradient_descent",
                  learning_rate=0.1, momentum=0.5, corr_type="masking", corr_frac=0.3, seed=0, alpha=1.0,
                  triplet_strategy="none", dt=np.float32, bh0=None, bv0=None, plans=None):
    """Restated DenoisingAutoencoder.fit (autoencoder.py:126-246) with injected W0 (xavier via
    tf.random_uniform is not reproducible).  Returns dict(W,bh,bv,history) where history[e] holds the
    per-batch lists the reference averages at :283-294.  ``plans`` optionally injects per-epoch
    (x_corrupted, index lists) instead of drawing them from the legacy RNG."""
    if seed >= 0:
        np.random.seed(seed)                                               # autoencoder.py:72-73
    F = train_set.shape[1]
    W = np.array(W0, dtype=dt, copy=True)
    H = W.shape[1]
    bh = np.zeros(H, dt) if bh0 is None else np.array(bh0, dt)
    bv = np.zeros(F, dt) if bv0 is None else np.array(bv0, dt)
    st = OptState(opt, [W.shape, bh.shape, bv.shape], dt)
    labels = None if labels is None else np.asarray(labels)
    history = []
    for e in range(num_epochs):
        xc, batches = plans[e] if plans is not None else epoch_plan(train_set, corr_type, corr_frac, batch_size)
        rec = dict(cost=[], ae=[], triplet=[], fraction=[], num=[])
        for idx in batches:
            xb = train_set[idx]; xcb = xc[idx]
            lb = None if labels is None else labels[idx]
            r = forward_backward(W, bh, bv, xb, xcb, lb, enc_act=enc_act, dec_act=dec_act, loss_func=loss_func,
                                 triplet_strategy=triplet_strategy, alpha=alpha, dt=dt)
            opt_apply(st, [W, bh, bv], [r["dW"], r["dbh"], r["dbv"]], learning_rate, momentum, dt)
            rec["cost"].append(float(r["cost"])); rec["ae"].append(float(r["ae_loss"]))
            rec["triplet"].append(float(r["triplet_loss"])); rec["fraction"].append(float(r["fraction"]))
            rec["num"].append(float(r["num"]))
        history.append(rec)
    return dict(W=W, bh=bh, bv=bv, history=history)


# --------------------------------------------------------------------------- #
# Philox4x32-10 (the device-side counter RNG of the perf path), restated for tests
# --------------------------------------------------------------------------- #
_PH_M0 = np.uint64(0xD2511F53); _PH_M1 = np.uint64(0xCD9E8D57)
_PH_W0 = np.uint32(0x9E3779B9); _PH_W1 = np.uint32(0xBB67AE85)


def philox4x32(c0, c1, c2, c3, k0, k1, rounds=10):
    """Vectorised Philox4x32-10 (Salmon et al. 2011).  Inputs uint32 arrays; returns 4 uint32 arrays."""
    c0 = np.asarray(c0, np.uint32).copy(); c1 = np.asarray(c1, np.uint32).copy()
    c2 = np.asarray(c2, np.uint32).copy(); c3 = np.asarray(c3, np.uint32).copy()
    k0 = np.uint32(k0); k1 = np.uint32(k1)
    with np.errstate(over="ignore"):
        for _ in range(rounds):
            p0 = c0.astype(np.uint64) * _PH_M0
            p1 = c2.astype(np.uint64) * _PH_M1
            hi0 = (p0 >> np.uint64(32)).astype(np.uint32); lo0 = p0.astype(np.uint32)
            hi1 = (p1 >> np.uint64(32)).astype(np.uint32); lo1 = p1.astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            k0 = np.uint32(k0 + _PH_W0); k1 = np.uint32(k1 + _PH_W1)
    return c0, c1, c2, c3


def philox_uniform(idx, seed, stream):
    """Uniform [0,1) fp32 for element counter ``idx`` (uint64 array): counter = (idx_lo, idx_hi,
    stream, 0), key = (seed_lo, seed_hi); u = (x0 >> 8) * 2^-24.  Mirrors csrc/dae_rng.h."""
    idx = np.asarray(idx, np.uint64)
    lo = (idx & np.uint64(0xFFFFFFFF)).astype(np.uint32); hi = (idx >> np.uint64(32)).astype(np.uint32)
    x0, _, _, _ = philox4x32(lo, hi, np.full(lo.shape, np.uint32(stream)), np.zeros(lo.shape, np.uint32),
                             np.uint32(seed & 0xFFFFFFFF), np.uint32((seed >> 32) & 0xFFFFFFFF))
    return (x0 >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)


def philox_uniform_dense(rows, F, seed, stream):
    """Uniforms of the dense-ndarray masking (csrc/dae_rng.h::philox_dense4): element (row, f) takes word f & 3 of the draw at
    counter (f >> 2, row, stream, 2), key = (seed_lo, seed_hi).  Returns float32 [len(rows) x F]."""
    rows = np.asarray(rows, np.uint32)
    nq = (F + 3) // 4
    c0 = np.broadcast_to(np.arange(nq, dtype=np.uint32)[None, :], (len(rows), nq)).ravel()
    c1 = np.broadcast_to(rows[:, None], (len(rows), nq)).ravel()
    out = philox4x32(c0, c1, np.full(c0.shape, np.uint32(stream)), np.full(c0.shape, np.uint32(2)),
                     np.uint32(seed & 0xFFFFFFFF), np.uint32((seed >> 32) & 0xFFFFFFFF))
    w = np.stack(out, axis=1).reshape(len(rows), nq * 4)[:, :F]
    return (w >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)


def salt_and_pepper_philox(X, rows, v, seed, stream, lo=None, hi=None):
    """utils.salt_and_pepper_noise (utils.py:118-144) with the device's counter RNG instead of the host stream: for train-set
    row r the t-th draw is Philox4x32-10 at counter (r, t, stream, 1): column = floor(x0 / 2^32 * F), coin = top bit of x1
    (0 -> global min, 1 -> global max); later draws win.  Mirrors csrc/dae_gather.hip::salt_pepper_kernel.  Returns the dense
    corrupted rows [len(rows) x F]."""
    Xd = X.toarray() if sparse.issparse(X) else np.asarray(X)
    F = Xd.shape[1]
    lo = Xd.min() if lo is None else lo
    hi = Xd.max() if hi is None else hi
    out = Xd[np.asarray(rows)].astype(np.float64).copy()
    t = np.arange(v, dtype=np.uint32)
    for i, r in enumerate(np.asarray(rows)):
        x0, x1, _, _ = philox4x32(np.full(v, np.uint32(r)), t, np.full(v, np.uint32(stream)), np.ones(v, np.uint32),
                                  np.uint32(seed & 0xFFFFFFFF), np.uint32((seed >> 32) & 0xFFFFFFFF))
        cols = ((x0.astype(np.uint64) * np.uint64(F)) >> np.uint64(32)).astype(np.int64)
        coin = (x1 >> np.uint32(31)).astype(np.int64)
        for c, k in zip(cols, coin):                       # sequential: later draws win
            out[i, c] = hi if k else lo
    return out


# ------------------------------------------------------------------------------------------------------------------
# Evaluation step after the path: helpers.pairwise_similarity (helpers.py:11-50)
# ------------------------------------------------------------------------------------------------------------------
def pairwise_similarity(in_df, norm="", metric="cosine", set_diagonal_zero=True, dt=np.float64):
    """Restates helpers.pairwise_similarity: optional sklearn.preprocessing.normalize(in_df, norm) (rows with a zero norm
    are left as they are), then sklearn's cosine_similarity (l2-normalise the rows, X X^T) or linear_kernel (X X^T),
    then np.fill_diagonal(out, 0) (helpers.py:43-48).  Pinned against scikit-learn itself in tests/test_oracle.py."""
    assert metric in ["cosine", "linear kernel"]                       # helpers.py:34
    X = np.asarray(in_df.toarray() if hasattr(in_df, "toarray") else in_df, dtype=dt)

    def _normalize(A, kind):
        if kind == "l1":
            n = np.abs(A).sum(axis=1)
        elif kind == "l2":
            n = np.sqrt((A * A).sum(axis=1))
        elif kind == "max":
            n = np.abs(A).max(axis=1)
        else:
            raise ValueError(f"'{kind}' is not a supported norm")
        n = np.where(n == 0, 1.0, n)
        return A / n[:, None]

    if norm != "":
        X = _normalize(X, norm)
    if metric == "cosine":
        X = _normalize(X, "l2")
    out = X @ X.T
    if set_diagonal_zero:
        np.fill_diagonal(out, 0)
    return out


def pair_stats(labels, S):
    """Numeric content of helpers.visualize_pairwise_similarity (helpers.py:79-135): related / unrelated scores of the strict
    lower triangle (labels < 0 missing), AUROC with ties counted half (= sklearn roc_curve + auc, pinned in
    tests/test_oracle.py), and the box-plot numbers."""
    labels = np.asarray(labels).reshape(len(labels), -1)[:, 0]
    S = np.asarray(S, np.float64)
    ok = (labels[None, :] >= 0) & (labels[:, None] >= 0)
    same = (labels[None, :] == labels[:, None]) & ok
    low = np.tril(np.ones_like(same, dtype=bool), -1)
    rel = S[same & low]
    un = S[(~same) & ok & low]
    out = dict(n_related=len(rel), n_unrelated=len(un), auroc=float("nan"))
    if len(rel) and len(un):
        allv = np.concatenate([rel, un])
        order = np.argsort(allv, kind="mergesort")
        ranks = np.empty(len(allv))
        sv = allv[order]
        i = 0
        while i < len(sv):                                   # average ranks over ties
            j = i
            while j + 1 < len(sv) and sv[j + 1] == sv[i]:
                j += 1
            ranks[order[i:j + 1]] = 0.5 * (i + j) + 1.0
            i = j + 1
        u = ranks[:len(rel)].sum() - len(rel) * (len(rel) + 1) / 2.0
        out["auroc"] = float(u / (len(rel) * len(un)))
    for name, v in (("related", rel), ("unrelated", un)):
        if len(v):
            out["mean_" + name] = float(v.mean())
            out[name] = dict(min=float(v.min()), q1=float(np.percentile(v, 25)), median=float(np.percentile(v, 50)),
                             q3=float(np.percentile(v, 75)), max=float(v.max()))
    return out
