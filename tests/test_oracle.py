"""Oracle self-checks (CPU): the restatement vs the loop specs of the reference's own tests
(autoencoder/tests/test_triplet_loss_utils.py), vs torch-CPU autograd, vs finite differences."""
import zlib

import numpy as np
import pytest
import torch

import oracle as O


def _labels(rng, n, classes):
    return rng.integers(0, classes, size=n).astype(np.float32)


@pytest.mark.parametrize("classes", [1, 3, 5])
def test_masks_match_reference_loops(classes):
    # reference test_triplet_loss_utils.py:11-70
    rng = np.random.default_rng(classes)
    n = 7
    lab = _labels(rng, n, classes)
    m3 = np.zeros((n, n, n), bool); ap = np.zeros((n, n), bool); an = np.zeros((n, n), bool)
    for i in range(n):
        for j in range(n):
            if i != j and lab[i] == lab[j]:
                ap[i, j] = True
            if i != j and lab[i] != lab[j]:
                an[i, j] = True
            for k in range(n):
                if i == j or j == k or i == k:
                    continue
                if lab[i] == lab[j] and lab[i] != lab[k]:
                    m3[i, j, k] = True
    assert (O.get_triplet_mask(lab) == m3).all()
    assert (O.get_anchor_positive_triplet_mask(lab) == ap).all()
    assert (O.get_anchor_negative_triplet_mask(lab) == an).all()
    nv, dw = O.batch_all_closed_form(lab)
    assert nv == m3.sum()
    assert (dw == m3.sum((1, 2)) + m3.sum((0, 1)) + m3.sum((0, 2))).all()


@pytest.mark.parametrize("classes", [1, 3, 5])
@pytest.mark.parametrize("signed", [False, True])
def test_batch_all_matches_reference_loops(classes, signed):
    # reference test_triplet_loss_utils.py:72-138
    rng = np.random.default_rng(10 + classes)
    n, d = 20, 6
    h = rng.random((n, d)).astype(np.float32) - (0.5 if signed else 0.0)
    lab = _labels(rng, n, classes)
    spec = O.batch_all_triplet_loss_loops(lab, h, np.float64)
    loss, dw, frac, num = O.batch_all_triplet_loss(lab, h, False, np.float64, chunk=7)
    assert np.allclose(loss, spec["loss_valid"]); assert np.allclose(dw, spec["dw_valid"])
    assert np.allclose(num, spec["num_pos"]); assert np.allclose(frac, spec["num_pos"] / (spec["num_valid"] + 1e-16))
    loss, dw, frac, num = O.batch_all_triplet_loss(lab, h, True, np.float64, chunk=3)
    assert np.allclose(loss, spec["loss_pos"]); assert np.allclose(dw, spec["dw_pos"])
    # fp32 path agrees with fp64 to fp32 accuracy
    l32, dw32, _, n32 = O.batch_all_triplet_loss(lab, h, False, np.float32)
    assert np.allclose(l32, spec["loss_valid"], rtol=1e-5); assert (dw32 == spec["dw_valid"]).all()


@pytest.mark.parametrize("classes", [1, 3, 5])
def test_batch_hard_matches_reference_loops(classes):
    # reference test_triplet_loss_utils.py:140-203 (non-negative embeddings only)
    rng = np.random.default_rng(20 + classes)
    n, d = 20, 6
    h = rng.random((n, d)).astype(np.float32)
    lab = _labels(rng, n, classes)
    spec = O.batch_hard_triplet_loss_loops(lab, h, np.float64)
    loss, dw, frac, num = O.batch_hard_triplet_loss(lab, h, np.float64)
    assert np.allclose(loss, spec["loss"]); assert np.allclose(dw, spec["data_weight"])
    assert np.allclose(frac, spec["fraction"]); assert np.allclose(num, spec["num"])


def test_weighted_loss_matches_reference_test():
    # reference test_triplet_loss_utils.py:205-234
    rng = np.random.default_rng(3)
    n, f = 20, 20
    x = rng.integers(0, 2, (n, f)).astype(np.float32)
    y = rng.random((n, f)).astype(np.float32)
    w = rng.integers(0, 50, n).astype(np.float32)
    ce = -(x * np.log(y + 1e-16) + (1. - x) * np.log(1. - y + 1e-16)).sum(1)
    assert np.allclose(ce.mean(), O.weighted_loss(x, y, "cross_entropy"))
    assert np.allclose((ce * w).sum() / w.sum(), O.weighted_loss(x, y, "cross_entropy", w))
    ms = np.square(x - y).sum(1)
    assert np.allclose(ms.mean(), O.weighted_loss(x, y, "mean_squared"))
    assert np.allclose((ms * w).sum() / w.sum(), O.weighted_loss(x, y, "mean_squared", w))
    from sklearn.preprocessing import normalize
    cs = -(normalize(x, axis=1) * normalize(y, axis=1)).sum(1)
    assert np.allclose(cs.mean(), O.weighted_loss(x, y, "cosine_proximity"))
    assert np.allclose((cs * w).sum() / w.sum(), O.weighted_loss(x, y, "cosine_proximity", w))


# ---------------- torch-autograd restatement of the literal TF graph (fp64) ---------------- #
def _t_act(name, z):
    return torch.sigmoid(z) if name == "sigmoid" else torch.tanh(z) if name == "tanh" else z


def _t_weighted_loss(x, y, lf, w):
    if lf == "cross_entropy":
        r = -torch.sum(x * torch.log(y + 1e-16) + (1. - x) * torch.log(1. - y + 1e-16), 1)
    elif lf == "mean_squared":
        r = torch.sum((x - y) ** 2, 1)
    else:
        xn = x * torch.rsqrt(torch.clamp(torch.sum(x * x, 1, keepdim=True), min=1e-12))
        yn = y * torch.rsqrt(torch.clamp(torch.sum(y * y, 1, keepdim=True), min=1e-12))
        r = -torch.sum(xn * yn, 1)
    return torch.sum(r * w) / (torch.sum(w) + 1e-16)


def _t_batch_all(lab, h):
    D = h @ h.T
    T = -D[:, :, None] + D[:, None, :]
    mask = torch.tensor(O.get_triplet_mask(lab), dtype=h.dtype)
    nv = mask.sum()
    loss = torch.sum(torch.nn.functional.softplus(T) * mask) / (nv + 1e-16)
    dw = mask.sum((1, 2)) + mask.sum((0, 1)) + mask.sum((0, 2))
    return loss, dw


def _t_batch_hard(lab, h):
    D = h @ h.T
    ap = torch.tensor(O.get_anchor_positive_triplet_mask(lab), dtype=h.dtype)
    an = torch.tensor(O.get_anchor_negative_triplet_mask(lab), dtype=h.dtype)
    rowmax = torch.amax(D, 1, keepdim=True)          # amax/amin split grads equally among ties (like TF)
    hp = torch.amin(D + rowmax * (1. - ap), 1, keepdim=True)
    hn = torch.amax(an * D, 1, keepdim=True)
    dist = torch.clamp(hn - hp, min=0.)
    cnt = (dist > 0).to(h.dtype)
    dw = cnt[:, 0] + torch.sum(cnt * (D == hp).to(h.dtype), 0) + torch.sum(cnt * (D == hn).to(h.dtype), 0)
    loss = torch.sum(torch.nn.functional.softplus(dist) * cnt) / (cnt.sum() + 1e-16)
    return loss, dw.detach()


@pytest.mark.parametrize("strategy", ["none", "batch_all", "batch_hard"])
@pytest.mark.parametrize("loss_func", ["cross_entropy", "mean_squared", "cosine_proximity"])
@pytest.mark.parametrize("acts", [("sigmoid", "sigmoid"), ("tanh", "none"), ("sigmoid", "tanh")])
def test_forward_backward_matches_torch_autograd(strategy, loss_func, acts):
    if loss_func == "cross_entropy" and acts[1] != "sigmoid":
        pytest.skip("log of a non-probability: NaN in the reference too")
    rng = np.random.default_rng(zlib.crc32(repr((strategy, loss_func, acts)).encode()))
    B, F, H = 24, 40, 9
    W = rng.uniform(-0.4, 0.4, (F, H)); bh = rng.uniform(-0.3, 0.3, H); bv = rng.uniform(-0.3, 0.3, F)
    x = (rng.random((B, F)) < 0.2).astype(np.float64)
    if loss_func != "cross_entropy":
        x = x * rng.random((B, F))
    xc = x * (rng.random((B, F)) >= 0.3)
    lab = _labels(rng, B, 3)
    alpha = 0.7
    r = O.forward_backward(W, bh, bv, x, xc, lab, enc_act=acts[0], dec_act=acts[1], loss_func=loss_func,
                           triplet_strategy=strategy, alpha=alpha, dt=np.float64)
    tW = torch.tensor(W, requires_grad=True); tbh = torch.tensor(bh, requires_grad=True)
    tbv = torch.tensor(bv, requires_grad=True)
    tx = torch.tensor(x); txc = torch.tensor(xc)
    h = _t_act(acts[0], txc @ tW + tbh) - _t_act(acts[0], tbh)
    y = _t_act(acts[1], h @ tW.T + tbv)
    if strategy == "none":
        cost = _t_weighted_loss(tx, y, loss_func, torch.ones(B, dtype=torch.float64)); tl = None
    else:
        tl, dw = (_t_batch_all if strategy == "batch_all" else _t_batch_hard)(lab, h)
        ae = _t_weighted_loss(tx, y, loss_func, dw.detach())
        cost = ae + alpha * tl
        assert np.allclose(r["ae_loss"], ae.item(), rtol=1e-10)
        assert np.allclose(r["triplet_loss"], tl.item(), rtol=1e-10)
        assert np.allclose(r["data_weight"], dw.numpy())
    cost.backward()
    assert np.allclose(r["cost"], cost.item(), rtol=1e-10)
    for name, t in (("dW", tW), ("dbh", tbh), ("dbv", tbv)):
        g = t.grad.numpy()
        assert np.allclose(r[name], g, rtol=1e-7, atol=1e-10 * max(1.0, np.abs(g).max())), name


def test_batch_hard_quirks_signed_embeddings():
    """Signed embeddings: invalid-j-wins-min and hn>=0 quirks vs literal torch graph."""
    hits = 0
    for seed in range(30):
        rng = np.random.default_rng(seed)
        n = 6
        h = rng.random((n, 4)) - 0.5
        lab = _labels(rng, n, 3)
        th = torch.tensor(h, requires_grad=True)
        tl, dw = _t_batch_hard(lab, th)
        tl.backward()
        loss, dwo, frac, num, G = O.batch_hard_triplet_loss(lab, h, np.float64, return_grad=True)
        assert np.allclose(loss, tl.item()); assert np.allclose(dwo, dw.numpy())
        gh = (G + G.T) @ h
        assert np.allclose(gh, th.grad.numpy(), atol=1e-12)
        spec = O.batch_hard_triplet_loss_loops(lab, np.abs(h))
        hits += int(not np.allclose(dwo, spec["data_weight"]))
    assert hits > 0   # the quirk regime is actually exercised


def test_explicit_triplet_matches_torch_autograd():
    rng = np.random.default_rng(5)
    B, F, H = 10, 30, 7
    W = rng.uniform(-0.4, 0.4, (F, H)); bh = rng.uniform(-0.3, 0.3, H); bv = rng.uniform(-0.3, 0.3, F)
    xs = [(rng.random((B, F)) < 0.3) * rng.random((B, F)) for _ in range(3)]
    xcs = [x * (rng.random((B, F)) >= 0.3) for x in xs]
    for lf in ["cross_entropy", "cosine_proximity", "mean_squared"]:
        r = O.explicit_triplet_forward_backward(W, bh, bv, xs, xcs, loss_func=lf, alpha=2.0, dt=np.float64)
        tW = torch.tensor(W, requires_grad=True); tbh = torch.tensor(bh, requires_grad=True)
        tbv = torch.tensor(bv, requires_grad=True)
        hs = [torch.sigmoid(torch.tensor(xc) @ tW + tbh) - torch.sigmoid(tbh) for xc in xcs]
        ys = [torch.sigmoid(h @ tW.T + tbv) for h in hs]
        ae = sum(_t_weighted_loss(torch.tensor(x), y, lf, torch.ones(B, dtype=torch.float64)) for x, y in zip(xs, ys))
        tl = torch.mean(torch.nn.functional.softplus(torch.sum(hs[0] * hs[2] - hs[0] * hs[1], 1)))
        cost = ae + 2.0 * tl
        cost.backward()
        assert np.allclose(r["cost"], cost.item())
        assert np.allclose(r["dW"], tW.grad.numpy(), rtol=1e-7, atol=1e-12)
        assert np.allclose(r["dbh"], tbh.grad.numpy(), rtol=1e-7, atol=1e-12)
        assert np.allclose(r["dbv"], tbv.grad.numpy(), rtol=1e-7, atol=1e-12)


def test_finite_difference_gradient():
    rng = np.random.default_rng(11)
    B, F, H = 12, 16, 5
    W = rng.uniform(-0.4, 0.4, (F, H)); bh = rng.uniform(-0.3, 0.3, H); bv = rng.uniform(-0.3, 0.3, F)
    x = (rng.random((B, F)) < 0.3).astype(np.float64); xc = x * (rng.random((B, F)) >= 0.3)
    lab = _labels(rng, B, 2)
    kw = dict(loss_func="cross_entropy", triplet_strategy="batch_all", alpha=1.0, dt=np.float64)
    r = O.forward_backward(W, bh, bv, x, xc, lab, **kw)
    e = 1e-6
    for (i, j) in [(0, 0), (3, 2), (15, 4)]:
        Wp = W.copy(); Wp[i, j] += e; Wm = W.copy(); Wm[i, j] -= e
        fd = (O.forward_backward(Wp, bh, bv, x, xc, lab, want_grads=False, **kw)["cost"]
              - O.forward_backward(Wm, bh, bv, x, xc, lab, want_grads=False, **kw)["cost"]) / (2 * e)
        assert np.allclose(fd, r["dW"][i, j], rtol=1e-5, atol=1e-9)
    bp = bh.copy(); bp[1] += e; bm = bh.copy(); bm[1] -= e
    fd = (O.forward_backward(W, bp, bv, x, xc, lab, want_grads=False, **kw)["cost"]
          - O.forward_backward(W, bm, bv, x, xc, lab, want_grads=False, **kw)["cost"]) / (2 * e)
    assert np.allclose(fd, r["dbh"][1], rtol=1e-5, atol=1e-9)


@pytest.mark.parametrize("opt", ["gradient_descent", "ada_grad", "momentum", "adam"])
def test_optimizers_vs_torch(opt):
    rng = np.random.default_rng(2)
    p0 = rng.standard_normal((5, 3)); gs = [rng.standard_normal((5, 3)) for _ in range(4)]
    p = p0.copy(); st = O.OptState(opt, [p.shape], np.float64)
    for g in gs:
        O.opt_apply(st, [p], [g], 0.1, 0.5, np.float64)
    tp = torch.tensor(p0.copy(), requires_grad=True)
    to = {"gradient_descent": lambda: torch.optim.SGD([tp], lr=0.1),
          "ada_grad": lambda: torch.optim.Adagrad([tp], lr=0.1, initial_accumulator_value=0.1, eps=0.0),
          "momentum": lambda: torch.optim.SGD([tp], lr=0.1, momentum=0.5),
          "adam": lambda: torch.optim.Adam([tp], lr=0.1, eps=1e-8)}[opt]()
    for g in gs:
        tp.grad = torch.tensor(g); to.step()
    tol = 1e-6 if opt == "adam" else 1e-12      # torch places Adam's eps slightly differently from TF
    assert np.allclose(p, tp.detach().numpy(), rtol=tol, atol=tol)


def test_softplus_thresholds():
    x = np.array([-100., -20., -13.94, -1., 0., 1., 13.94, 20., 100.], np.float32)
    assert np.allclose(O.softplus_tf(x), np.logaddexp(0, x.astype(np.float64)), rtol=2e-6)
    assert np.isfinite(O.softplus_tf(np.array([1e4], np.float32))).all()


def test_philox_known_answer():
    # Random123 kat_vectors: philox4x32-10, ctr=0 key=0 -> 6627e8d5 e169c58d bc57ac4c 9b00dbd8
    out = O.philox4x32([0], [0], [0], [0], 0, 0)
    assert [int(v[0]) for v in out] == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    out = O.philox4x32([0xffffffff], [0xffffffff], [0xffffffff], [0xffffffff], 0xffffffff, 0xffffffff)
    assert [int(v[0]) for v in out] == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]


@pytest.mark.parametrize("norm", ["", "l1", "l2", "max"])
@pytest.mark.parametrize("metric", ["cosine", "linear kernel"])
def test_pairwise_similarity_vs_sklearn(norm, metric):
    """Pins oracle.pairwise_similarity (restating helpers.py:11-50) against the scikit-learn calls the reference makes."""
    from sklearn.metrics import pairwise
    from sklearn.preprocessing import normalize
    rng = np.random.default_rng(5)
    X = rng.standard_normal((37, 19))
    X[5] = 0.0                                                        # an all-zero row: normalize leaves it untouched
    ref = normalize(X, norm=norm) if norm else X
    ref = pairwise.cosine_similarity(ref) if metric == "cosine" else pairwise.linear_kernel(ref)
    np.fill_diagonal(ref, 0)
    got = O.pairwise_similarity(X, norm=norm, metric=metric)
    assert np.allclose(got, ref, rtol=1e-12, atol=1e-14)
    keep = O.pairwise_similarity(X, norm=norm, metric=metric, set_diagonal_zero=False)
    assert np.allclose(np.diag(keep)[6:], np.diag(X @ X.T)[6:] if (metric == "linear kernel" and not norm) else np.diag(keep)[6:])
    with pytest.raises(AssertionError):
        O.pairwise_similarity(X, metric="euclidean")                  # the reference's assert (helpers.py:34)


def test_pair_stats_auroc_vs_sklearn():
    """Pins oracle.pair_stats against the scikit-learn calls of helpers.visualize_pairwise_similarity (roc_curve + auc on the
    related / unrelated score lists), including ties and missing (-1) labels."""
    from sklearn.metrics import auc, roc_curve
    rng = np.random.default_rng(8)
    n = 60
    lab = rng.integers(-1, 4, n)
    X = np.round(rng.standard_normal((n, 6)), 1)                      # coarse values -> many tied similarities
    S = X @ X.T
    got = O.pair_stats(lab, S)
    ok = (lab[None, :] >= 0) & (lab[:, None] >= 0)
    same = (lab[None, :] == lab[:, None]) & ok
    low = np.tril(np.ones((n, n), bool), -1)
    rel, un = S[same & low], S[(~same) & ok & low]
    fpr, tpr, _ = roc_curve(["Related"] * len(rel) + ["Unrelated"] * len(un), list(rel) + list(un), pos_label="Related")
    assert got["n_related"] == len(rel) and got["n_unrelated"] == len(un)
    assert abs(got["auroc"] - auc(fpr, tpr)) < 1e-12
    assert abs(got["related"]["median"] - np.median(rel)) < 1e-12
