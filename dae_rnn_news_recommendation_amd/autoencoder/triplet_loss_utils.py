"""Function surface of the reference's ``autoencoder/triplet_loss_utils.py`` on the MI355X kernels.

Same names, argument order and return tuples as the reference; tensors are NumPy arrays (or CUDA
torch tensors) instead of TF graph nodes and every call evaluates eagerly on the GPU through
libdae_hip's C ABI.  There is no CPU implementation behind these functions: without the built
library / a GPU they raise.
"""
from __future__ import annotations

import numpy as np

from .. import _lib as L


def _labels_to_ids(labels):
    """Any 1-d label container -> dense int32 ids (equality-preserving, like the float32 feed of the
    reference: autoencoder.py:352)."""
    a = np.asarray(labels).reshape(-1)
    _, inv = np.unique(a, return_inverse=True)
    return inv.astype(np.int32)


# --- 2-D / 3-D label masks: pure label logic, evaluated on the host (they are debugging helpers in the
# reference too -- the miners themselves never materialise them on the device) --------------------------------
def _get_anchor_positive_triplet_mask(labels):
    """mask[a, p] = (a != p) and labels[a] == labels[p]   (reference :6-26)."""
    lab = np.asarray(labels).reshape(-1)
    same = lab[:, None] == lab[None, :]
    np.fill_diagonal(same, False)
    return same


def _get_anchor_negative_triplet_mask(labels):
    """mask[a, n] = labels[a] != labels[n]   (reference :29-44)."""
    lab = np.asarray(labels).reshape(-1)
    return lab[:, None] != lab[None, :]


def _get_triplet_mask(labels):
    """mask[a, p, n] = a,p,n distinct and labels[a] == labels[p] != labels[n]   (reference :47-76).
    label[a] != label[n] already implies a != n and p != n, so the product of the two 2-D masks is exact."""
    ap = _get_anchor_positive_triplet_mask(labels)
    an = _get_anchor_negative_triplet_mask(labels)
    return ap[:, :, None] & an[:, None, :]


def _device_encode(encode):
    import torch
    from .. import ops
    if isinstance(encode, torch.Tensor):
        h = encode.detach().to(device="cuda", dtype=torch.float32)
    else:
        h = torch.as_tensor(np.ascontiguousarray(encode, dtype=np.float32)).cuda()
    B, H = h.shape
    hp = torch.zeros((L.pad(B), L.pad(H)), dtype=torch.float32, device="cuda")
    hp[:B, :H] = h
    return ops, torch, hp, B


def _mine(strategy, input_label, encode, pos_triplets_only=False):
    ops, torch, hp, B = _device_encode(encode)
    labels = torch.as_tensor(_labels_to_ids(input_label)).cuda()
    assert labels.numel() == B
    D = ops.gram(hp, splits=1)
    nvalid, dw_i64, cw = ops.label_stats(labels, B, L.TRIPLET["batch_all"])
    if strategy == "batch_all":
        lp, cnt, G, role = ops.triplet_batch_all(D, labels, B, pos_triplets_only)
        tri, dwf = ops.triplet_finalize(L.TRIPLET["batch_all"], pos_triplets_only, B, 1.0, lp, cnt, nvalid, None, role, cw)
        dw = dwf[:B] if pos_triplets_only else dw_i64[:B].to(torch.float32)
    else:
        lp, cnt, dwi, G = ops.triplet_batch_hard(D, labels, B)
        tri, dwf = ops.triplet_finalize(L.TRIPLET["batch_hard"], False, B, 1.0, lp, cnt, nvalid, dwi, None, cw)
        dw = dwf[:B]
    t = tri.cpu().numpy()
    return np.float32(t[1]), dw.cpu().numpy().astype(np.float32), np.float32(t[2]), np.float32(t[3])


def batch_all_triplet_loss(sparse_input, input_label, encode, pos_triplets_only=False):
    """All valid (anchor, positive, negative) triplets of the batch on the dot-product Gram matrix:
    loss = sum softplus(D[a,n]-D[a,p]) * mask / (num + 1e-16)   (reference :79-131).

    Returns (triplet_loss, data_weight[B], fraction_positive, num_positive)."""
    return _mine("batch_all", input_label, encode, pos_triplets_only)


def batch_hard_triplet_loss(sparse_input, input_label, encode):
    """Hardest positive / hardest negative per anchor (reference :202-259, quirks included).

    Returns (triplet_loss, data_weight[B], fraction, num)."""
    return _mine("batch_hard", input_label, encode)


def weighted_loss(sparse_input, input_data, decode, loss_func='cross_entropy', weight=None):
    """sum_i(rowloss_i * w_i) / (sum_i w_i + 1e-16) with rowloss in {cross_entropy, mean_squared,
    cosine_proximity}   (reference :262-277).  ``input_data`` may be dense or scipy-sparse."""
    import torch
    from scipy import sparse
    from .. import ops  # noqa: F401  (forces the library to load / fail loudly)
    x = input_data.toarray() if sparse.issparse(input_data) else np.asarray(input_data)
    x = torch.as_tensor(np.ascontiguousarray(x, dtype=np.float32)).cuda()
    y = decode.detach().to("cuda", torch.float32).contiguous() if isinstance(decode, torch.Tensor) \
        else torch.as_tensor(np.ascontiguousarray(decode, dtype=np.float32)).cuda()
    B, F = x.shape
    Bp = L.pad(B)
    w = np.ones(B, np.float32) if weight is None else np.asarray(weight, np.float32).reshape(-1)
    cw = torch.zeros(Bp, dtype=torch.float32, device="cuda")
    cw[:B] = torch.as_tensor(w / (np.float32(w.sum(dtype=np.float32)) + np.float32(1e-16))).cuda()
    rowloss = torch.zeros((1, Bp), dtype=torch.float32, device="cuda")
    L.call("dae_weighted_loss_rows", L.ptr(x), x.stride(0), L.ptr(y), y.stride(0), B, F, L.LOSS[loss_func], L.ptr(rowloss),
           L.current_stream())
    stats = ops.step_stats(rowloss, cw, B, L.TRIPLET["none"], 0.0, None, None)
    return np.float32(stats[L.STAT_AE].item())
