"""PyTorch-CPU fp32 restatement of the reference's training step -- the CPU BASELINE of bench.py.

TEST / MEASUREMENT INFRASTRUCTURE ONLY (see oracle/__init__.py): bench.py's ``cpu_baseline`` leg times it on the host
cores of the GPU box; nothing in the product imports it.  It stands in for the reference's TF-CPU path, which cannot
run here (tensorflow==1.12.0 has no wheel for this Python, no network): the graph of
``autoencoder/autoencoder.py:371-477`` and ``autoencoder/triplet_loss_utils.py:79-131,202-277`` written op for op in
torch (sparse matmul, dense matmul, sigmoid, the B x B x B broadcast of batch_all, ``weighted_loss``), differentiated by
autograd and updated by plain SGD, in float32 like TF.

Two forms of batch_all:
  * ``literal``  -- materialises the [B, B, B] tensors exactly as the reference does (2.05 GB each at B = 800, several of
                    them alive for autograd): what a TF-CPU run pays;
  * ``chunked``  -- the same sums anchor-chunk by anchor-chunk (no B^3 tensor), with d loss / d D accumulated by hand and
                    pushed through autograd from D on: arithmetically identical, memory-lean, used for whole epochs.
"""
import time

import numpy as np
import torch


def _sparse(m):
    if isinstance(m, np.ndarray):                      # dense ndarray input (autoencoder.py:143 sparse_input = False)
        return torch.from_numpy(np.ascontiguousarray(m, dtype=np.float32))
    m = m.tocoo()
    i = torch.from_numpy(np.vstack([m.row, m.col]).astype(np.int64))
    return torch.sparse_coo_tensor(i, torch.from_numpy(m.data.astype(np.float32)), m.shape).coalesce()


def _masks(lab):
    eq = lab.unsqueeze(0) == lab.unsqueeze(1)
    B = lab.shape[0]
    ne = ~torch.eye(B, dtype=torch.bool)
    return eq, ne


def batch_all_literal(lab, h):
    """triplet_loss_utils.py:79-131 with the B^3 tensors materialised."""
    D = h @ h.t()
    T = -D.unsqueeze(2) + D.unsqueeze(1)
    eq, ne = _masks(lab)
    distinct = ne.unsqueeze(2) & ne.unsqueeze(1) & ne.unsqueeze(0)
    valid = (distinct & eq.unsqueeze(2) & (~eq).unsqueeze(1)).float()
    nv = valid.sum()
    npos = ((valid * T) > 1e-16).float().sum()
    loss = (torch.nn.functional.softplus(T) * valid).sum() / (nv + 1e-16)
    dw = valid.sum((1, 2)) + valid.sum((0, 1)) + valid.sum((0, 2))
    return loss, dw, npos / (nv + 1e-16), npos


def batch_all_chunked(lab, h, chunk=32):
    """Same sums without a B^3 tensor: loss and d loss/d D per anchor chunk, then one backward from D."""
    D = h @ h.t()
    B = D.shape[0]
    eq, ne = _masks(lab)
    Dd = D.detach()
    G = torch.zeros_like(Dd)
    tot = torch.zeros((), dtype=torch.float64)
    npos = 0.0
    nv = 0.0
    dw = torch.zeros(B)
    for a0 in range(0, B, chunk):
        a1 = min(B, a0 + chunk)
        ap = (eq[a0:a1] & ne[a0:a1]).float()
        an = (~eq[a0:a1]).float()
        T = -Dd[a0:a1].unsqueeze(2) + Dd[a0:a1].unsqueeze(1)
        valid = ap.unsqueeze(2) * an.unsqueeze(1)
        tot += (torch.nn.functional.softplus(T) * valid).sum().double()
        sg = torch.sigmoid(T) * valid
        G[a0:a1] += sg.sum(1) - sg.sum(2)
        npos += float(((valid * T) > 1e-16).sum())
        nv += float(valid.sum())
        dw[a0:a1] += valid.sum((1, 2))
        dw += valid.sum((0, 1)) + valid.sum((0, 2))
    loss_val = (tot / (nv + 1e-16)).float()
    # a scalar with the right value AND the right gradient with respect to D
    loss = loss_val + ((D - Dd) * (G / (nv + 1e-16))).sum()
    return loss, dw, npos / (nv + 1e-16), npos


def batch_hard(lab, h):
    """triplet_loss_utils.py:202-259."""
    D = h @ h.t()
    eq, ne = _masks(lab)
    ap = (eq & ne).float()
    an = (~eq).float()
    rowmax = D.amax(1, keepdim=True)
    hp = (D + rowmax * (1.0 - ap)).amin(1, keepdim=True)
    hn = (an * D).amax(1, keepdim=True)
    dist = torch.clamp(hn - hp, min=0.0)
    cnt = (dist > 0).float()
    dw = cnt.squeeze() + (cnt * (D == hp).float()).sum(0) + (cnt * (D == hn).float()).sum(0)
    loss = (torch.nn.functional.softplus(dist) * cnt).sum() / (cnt.sum() + 1e-16)
    return loss, dw, cnt.sum() / D.shape[0], cnt.sum()


def weighted_ce(x_dense, y, w):
    row = -(x_dense * torch.log(y + 1e-16) + (1.0 - x_dense) * torch.log(1.0 - y + 1e-16)).sum(1)
    return (row * w).sum() / (w.sum() + 1e-16)


def train_step(W, bh, bv, x, xc, labels, strategy, alpha, lr, form="literal"):
    """One mini-batch: forward, cost (autoencoder.py:417-442), autograd, SGD.  x / xc: scipy CSR batches.  In-place update."""
    xs, xcs = _sparse(x), _sparse(xc)
    z1 = torch.sparse.mm(xcs, W) if xcs.is_sparse else xcs @ W            # tf.sparse.matmul / tf.matmul (:377)
    h = torch.sigmoid(z1 + bh) - torch.sigmoid(bh)
    y = torch.sigmoid(h @ W.t() + bv)
    xd = xs.to_dense() if xs.is_sparse else xs
    if strategy == "none":
        cost = weighted_ce(xd, y, torch.ones(xd.shape[0]))
        tl = torch.zeros(())
    else:
        lab = torch.from_numpy(np.asarray(labels).astype(np.int64))
        if strategy == "batch_all":
            tl, dw, _, _ = (batch_all_literal if form == "literal" else batch_all_chunked)(lab, h)
        else:
            tl, dw, _, _ = batch_hard(lab, h)
        cost = weighted_ce(xd, y, dw.detach()) + alpha * tl
    gW, gbh, gbv = torch.autograd.grad(cost, [W, bh, bv])
    with torch.no_grad():
        W -= lr * gW; bh -= lr * gbh; bv -= lr * gbv
    return float(cost.detach()), float(tl.detach())


def time_baseline(m, labels, W0, *, batch, strategy, corr_frac=0.3, lr=0.1, alpha=1.0, literal_steps=2, chunked_steps=10, seed=0,
                  threads=None, budget_s=25.0):
    """Times up to ``literal_steps`` literal steps and ``chunked_steps`` chunked steps (incl. the per-epoch masking + shuffle of
    the reference, utils.py) on ``threads`` host threads; each form stops early once it has used ``budget_s`` seconds (at least
    one step is always timed).  Returns a dict for bench.py's cpu_baseline object."""
    import os
    threads = threads or os.cpu_count() or 1
    torch.set_num_threads(threads)
    np.random.seed(seed)
    N = m.shape[0]
    out = {}
    for form, steps in (("literal", literal_steps), ("chunked", chunked_steps)):
        if steps <= 0 or (form == "literal" and strategy != "batch_all"):
            continue
        W = torch.from_numpy(np.array(W0, np.float32, copy=True)).requires_grad_(True)
        bh = torch.zeros(W.shape[1], requires_grad=True); bv = torch.zeros(W.shape[0], requires_grad=True)
        t0 = time.time()
        if isinstance(m, np.ndarray):                                          # dense masking (utils.py:107-109)
            mc = m * np.random.choice(a=[0, 1], size=m.shape, p=[corr_frac, 1 - corr_frac]).astype(np.float32)
        else:
            keep = np.random.rand(m.nnz) >= corr_frac                         # masking_noise on the whole set (utils.py:111)
            mc = m.copy(); mc.data = mc.data * keep; mc.eliminate_zeros()
        order = np.arange(N); np.random.shuffle(order)                         # gen_batches (utils.py:50-51)
        done = 0
        costs = []
        ran = 0
        for s in range(steps):
            idx = order[(s * batch) % N:(s * batch) % N + batch]
            c, _ = train_step(W, bh, bv, m[idx], mc[idx], None if labels is None else labels[idx], strategy, alpha, lr, form)
            costs.append(c); done += len(idx); ran += 1
            if time.time() - t0 > budget_s:
                break
        dt = time.time() - t0
        out[form] = dict(samples_per_s=done / dt, seconds=dt, steps=ran, first_cost=costs[0], last_cost=costs[-1])
    return out
