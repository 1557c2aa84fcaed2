"""Thin torch-tensor wrappers over the per-kernel C-ABI entry points of libdae_hip.so.

torch is plumbing here (device memory + streams); every function enqueues HIP kernels from the
hand-written library on the current stream and returns device tensors.  No function in this module
computes anything on the host or through torch ops: if the library is missing they raise.
"""
from __future__ import annotations

import torch

from . import _lib as L


def tdtype(dtype: int):
    return torch.bfloat16 if dtype == L.BF16 else torch.float32


def _dev(t):
    if t is not None and not t.is_cuda:
        raise RuntimeError("dae ops need CUDA(HIP) tensors: the hot path has no CPU fallback")


def gemm_nt(A0, Bt0, A1=None, Bt1=None, splits=1, K0=None, K1=None, M=None, N=None):
    """C = A0 @ Bt0.T (+ A1 @ Bt1.T); operands [rows x K] row-major (bf16 or fp32), fp32 slabs out."""
    _dev(A0)
    dtype = L.BF16 if A0.dtype == torch.bfloat16 else L.F32
    M = A0.shape[0] if M is None else M
    N = Bt0.shape[0] if N is None else N
    K0 = A0.shape[1] if K0 is None else K0
    K1 = 0 if A1 is None else (A1.shape[1] if K1 is None else K1)
    C = torch.empty((splits, M, N), dtype=torch.float32, device=A0.device)
    L.call("dae_gemm_nt", dtype, M, N, L.ptr(A0), A0.stride(0), L.ptr(Bt0), Bt0.stride(0), K0,
           L.ptr(A1), 0 if A1 is None else A1.stride(0), L.ptr(Bt1), 0 if Bt1 is None else Bt1.stride(0), K1,
           L.ptr(C), N, splits, M * N, L.current_stream())
    return C


def gemm_nt_n(segs, splits=1):
    """C = sum_s A_s @ Bt_s.T over 1..5 K segments [(A, Bt), ...]; operands [rows x K_s] row-major (all bf16 or all fp32)."""
    A0, Bt0 = segs[0]
    _dev(A0)
    dtype = L.BF16 if A0.dtype == torch.bfloat16 else L.F32
    M, N = A0.shape[0], Bt0.shape[0]
    arr = (L.dae_gemm_seg * len(segs))()
    for i, (a, b) in enumerate(segs):
        assert a.shape[0] == M and b.shape[0] == N and a.shape[1] == b.shape[1] and a.dtype == A0.dtype and b.dtype == A0.dtype
        arr[i].A, arr[i].lda, arr[i].Bt, arr[i].ldb, arr[i].K = a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), a.shape[1]
    C = torch.empty((splits, M, N), dtype=torch.float32, device=A0.device)
    L.call("dae_gemm_nt_n", dtype, M, N, arr, len(segs), L.ptr(C), N, splits, M * N, L.current_stream())
    return C


def gather_csr(indptr, indices, values, row_idx, B, F, dtype, *, want_x=True, want_xc=True, want_xct=True,
               want_rowsq=False, corr_mode=L.CORR_NONE, keep_bits=None, seed=0, rng_stream=0, corr_frac=0.0,
               scale=1.0):
    Bp, Fp = L.pad(B), L.pad(F)
    td = tdtype(dtype)
    dev = indptr.device
    x = torch.empty((Bp, Fp), dtype=td, device=dev) if want_x else None
    xc = torch.empty((Bp, Fp), dtype=td, device=dev) if want_xc else None
    xct = torch.zeros((Fp, Bp), dtype=td, device=dev) if want_xct else None
    rowsq = torch.empty(Bp, dtype=torch.float32, device=dev) if want_rowsq else None
    L.call("dae_gather_csr", L.ptr(indptr), L.ptr(indices), L.ptr(values), L.ptr(row_idx), B, F, dtype,
           L.ptr(x), L.ptr(xc), Fp, L.ptr(xct), Bp, L.ptr(rowsq), corr_mode, L.ptr(keep_bits), seed, rng_stream,
           corr_frac, scale, L.current_stream())
    return x, xc, xct, rowsq


def gather_csr_bits(indptr, indices, row_idx, B, F, dtype, *, corr_mode=L.CORR_NONE, keep_bits=None, seed=0,
                    rng_stream=0, corr_frac=0.0):
    """Binary CSR -> (x, x~ bit image [Bp x Fp/32] int32, x~^T); the operand set of the fused corrupt+encode GEMM."""
    Bp, Fp = L.pad(B), L.pad(F)
    td = tdtype(dtype)
    dev = indptr.device
    x = torch.empty((Bp, Fp), dtype=td, device=dev)
    xct = torch.zeros((Fp, Bp), dtype=td, device=dev)
    bits = torch.empty((Bp, Fp // 32), dtype=torch.int32, device=dev)
    L.call("dae_gather_csr_bits", L.ptr(indptr), L.ptr(indices), None, L.ptr(row_idx), B, F, dtype, L.ptr(x), None, Fp,
           L.ptr(xct), Bp, None, corr_mode, L.ptr(keep_bits), seed, rng_stream, corr_frac, 1.0, L.ptr(bits), Fp // 32,
           L.current_stream())
    return x, bits, xct


def encode_bits(bits, Wt_lo, splits=1):
    """slabs[s] = bits(x~) . Wt_lo^T  (bf16 MFMA, A expanded from the bit image in LDS)."""
    Bp, Fp, Hp = bits.shape[0], bits.shape[1] * 32, Wt_lo.shape[0]
    slabs = torch.empty((splits, Bp, Hp), dtype=torch.float32, device=bits.device)
    L.call("dae_encode_bits", L.ptr(bits), bits.stride(0), L.ptr(Wt_lo), Wt_lo.stride(0), Bp, Hp, Fp, L.ptr(slabs), Hp,
           splits, Bp * Hp, L.current_stream())
    return slabs


def gather_dense(data, row_idx, B, F, dtype, *, want_rowsq=False, corr_mode=L.CORR_NONE, keep_bits=None, seed=0,
                 rng_stream=0, corr_frac=0.0, scale=1.0):
    Bp, Fp = L.pad(B), L.pad(F)
    td = tdtype(dtype)
    dev = data.device
    x = torch.empty((Bp, Fp), dtype=td, device=dev)
    xc = torch.empty((Bp, Fp), dtype=td, device=dev)
    xct = torch.zeros((Fp, Bp), dtype=td, device=dev)
    rowsq = torch.empty(Bp, dtype=torch.float32, device=dev) if want_rowsq else None
    scratch = torch.empty((Fp // 64, Bp), dtype=torch.float32, device=dev) if want_rowsq else None
    L.call("dae_gather_dense", L.ptr(data), data.stride(0), L.ptr(row_idx), B, F, dtype, L.ptr(x), L.ptr(xc), Fp,
           L.ptr(xct), Bp, L.ptr(rowsq), L.ptr(scratch), corr_mode, L.ptr(keep_bits), seed, rng_stream, corr_frac,
           scale, L.current_stream())
    return x, xc, xct, rowsq


def encode_finish(slabs, bh, B, H, enc_act, dtype, want_hcat=False):
    S, Bp, Hp = slabs.shape
    dev = slabs.device
    h32 = torch.empty((Bp, Hp), dtype=torch.float32, device=dev)
    hlo = torch.empty((Bp, Hp), dtype=tdtype(dtype), device=dev)
    ht = torch.empty((Hp, Bp), dtype=tdtype(dtype), device=dev)
    ha = torch.empty((Bp, 3 * Hp), dtype=torch.bfloat16, device=dev) if want_hcat else None
    hb = torch.empty((Bp, 3 * Hp), dtype=torch.bfloat16, device=dev) if want_hcat else None
    L.call("dae_encode_finish", L.ptr(slabs), S, Bp * Hp, Hp, L.ptr(bh), B, H, enc_act, dtype, L.ptr(h32), L.ptr(hlo),
           Hp, L.ptr(ht), Bp, L.ptr(ha), L.ptr(hb), L.current_stream())
    if want_hcat:
        return h32, hlo, ht, ha, hb
    return h32, hlo, ht


def decode_loss(h_lo, W_lo, bv, x, cw, B, F, H, dec_act, loss_func, dtype, *, cos_pass=0, cos_stats=None):
    Bp, Fp, Hp = L.pad(B), L.pad(F), L.pad(H)
    dev = h_lo.device
    bn = int(L.load().dae_decode_tile_n(dtype))         # tile width of the decode kernel lays out its partial sums
    ncw, nrw = 2 * Fp // bn, 2 * Bp // 128
    rowloss_part = torch.zeros((ncw, Bp), dtype=torch.float32, device=dev)
    dbv_part = torch.zeros((nrw, Fp), dtype=torch.float32, device=dev)
    cos_part = torch.zeros((2, ncw, Bp), dtype=torch.float32, device=dev) if loss_func == 2 else None
    tile_part = torch.zeros((Bp // 128) * (Fp // bn), dtype=torch.float32, device=dev)
    d2 = torch.zeros((Bp, Fp), dtype=tdtype(dtype), device=dev)
    d2t = torch.zeros((Fp, Bp), dtype=tdtype(dtype), device=dev)
    L.call("dae_decode_loss", dtype, B, F, H, L.ptr(h_lo), Hp, L.ptr(W_lo), Hp, L.ptr(bv), L.ptr(x), Fp, L.ptr(cw),
           dec_act, loss_func, cos_pass, L.ptr(cos_stats), L.ptr(cos_part), L.ptr(rowloss_part), L.ptr(tile_part), L.ptr(dbv_part),
           L.ptr(d2), Fp, L.ptr(d2t), Bp, L.current_stream())
    return dict(rowloss_part=rowloss_part, tile_part=tile_part, dbv_part=dbv_part, cos_part=cos_part, delta2=d2, delta2_t=d2t)


def cos_reduce(cos_part, B, cos_stats):
    _, ncw, Bp = cos_part.shape
    rowloss = torch.empty(Bp, dtype=torch.float32, device=cos_part.device)
    L.call("dae_cos_reduce", L.ptr(cos_part), ncw, B, Bp, L.ptr(cos_stats), L.ptr(rowloss), L.current_stream())
    return rowloss


def gram(h_f32, splits=1):
    Bp, Hp = h_f32.shape
    D = torch.empty((splits, Bp, Bp), dtype=torch.float32, device=h_f32.device)
    L.call("dae_gram", L.ptr(h_f32), Hp, Bp, Hp, L.ptr(D), splits, L.current_stream())
    return D


def label_stats(labels, B, triplet):
    Bp = L.pad(B)
    dev = labels.device if labels is not None else torch.device("cuda")
    n_same = torch.zeros(Bp, dtype=torch.int32, device=dev)
    acc = torch.zeros(2, dtype=torch.int64, device=dev)
    nvalid = torch.zeros(1, dtype=torch.int64, device=dev)
    dw = torch.zeros(Bp, dtype=torch.int64, device=dev)
    cw = torch.zeros(Bp, dtype=torch.float32, device=dev)
    L.call("dae_label_stats", L.ptr(labels), B, Bp, triplet, L.ptr(n_same), L.ptr(acc), L.ptr(nvalid), L.ptr(dw),
           L.ptr(cw), 1.0, None, L.current_stream())
    return nvalid, dw, cw


def triplet_batch_all(D_slabs, labels, B, pos_only=False, fast=False):
    S, Bp, _ = D_slabs.shape
    dev = D_slabs.device
    loss_part = torch.zeros(Bp, dtype=torch.float32, device=dev)
    npos = torch.zeros(Bp, dtype=torch.int32, device=dev)
    G = torch.zeros((Bp, Bp), dtype=torch.float32, device=dev)
    role = torch.zeros((Bp, Bp), dtype=torch.int32, device=dev) if pos_only else None
    L.call("dae_triplet_batch_all", L.ptr(D_slabs), S, Bp * Bp, Bp, L.ptr(labels), B, Bp, int(pos_only) | (2 if fast else 0),
           L.ptr(loss_part), L.ptr(npos), L.ptr(G), L.ptr(role), L.current_stream())
    return loss_part, npos, G, role


def triplet_batch_hard(D_slabs, labels, B):
    S, Bp, _ = D_slabs.shape
    dev = D_slabs.device
    loss_part = torch.zeros(Bp, dtype=torch.float32, device=dev)
    cnt = torch.zeros(Bp, dtype=torch.int32, device=dev)
    dw = torch.zeros(Bp, dtype=torch.int32, device=dev)
    G = torch.zeros((Bp, Bp), dtype=torch.float32, device=dev)
    L.call("dae_triplet_batch_hard", L.ptr(D_slabs), S, Bp * Bp, Bp, L.ptr(labels), B, Bp, L.ptr(loss_part),
           L.ptr(cnt), L.ptr(dw), L.ptr(G), L.current_stream())
    return loss_part, cnt, dw, G


def triplet_finalize(triplet, pos_only, B, alpha, loss_part, cnt_part, nvalid, dw_i32, role_cnt, cw):
    Bp = loss_part.shape[0]
    dev = loss_part.device
    dw_f32 = torch.zeros(Bp, dtype=torch.float32, device=dev)
    tri = torch.zeros(4, dtype=torch.float32, device=dev)
    L.call("dae_triplet_finalize", triplet, int(pos_only), B, Bp, alpha, L.ptr(loss_part), L.ptr(cnt_part),
           L.ptr(nvalid), L.ptr(dw_i32), L.ptr(role_cnt), L.ptr(dw_f32), L.ptr(cw), L.ptr(tri), L.current_stream())
    return tri, dw_f32


def sym_scale(G, B, tri_scalars, dtype):
    Bp = G.shape[0]
    Gs = torch.empty((Bp, Bp), dtype=tdtype(dtype), device=G.device)
    L.call("dae_sym_scale", L.ptr(G), B, Bp, L.ptr(tri_scalars), dtype, L.ptr(Gs), L.current_stream())
    return Gs


def dh_finish(slabs, h_f32, bh, B, H, enc_act, dtype, dh_extra=None):
    S, Bp, Hp = slabs.shape
    dev = slabs.device
    d1t = torch.empty((Hp, Bp), dtype=tdtype(dtype), device=dev)
    colsum = torch.zeros((2, Bp // 32, Hp), dtype=torch.float32, device=dev)
    d1 = torch.empty((Bp, Hp), dtype=torch.float32, device=dev)
    L.call("dae_dh_finish", L.ptr(slabs), S, Bp * Hp, Hp, L.ptr(dh_extra), L.ptr(h_f32), Hp, L.ptr(bh), B, H, enc_act,
           dtype, L.ptr(d1t), Bp, L.ptr(colsum), L.ptr(d1), L.current_stream())
    return d1t, colsum, d1


def bias_grads(dbv_part, colsum_part, bh, H, F, enc_act):
    nrw, Fp = dbv_part.shape
    _, nrb, Hp = colsum_part.shape
    dev = dbv_part.device
    dbh = torch.empty(Hp, dtype=torch.float32, device=dev)
    dbv = torch.empty(Fp, dtype=torch.float32, device=dev)
    L.call("dae_bias_grads", L.ptr(dbv_part), nrw, L.ptr(colsum_part), nrb, L.ptr(bh), H, Hp, F, Fp, enc_act,
           L.ptr(dbh), L.ptr(dbv), 0, 0, 0.0, 0.0, 1.0, None, None, None, L.current_stream())
    return dbh, dbv


def opt_step(opt, lr, momentum, grad_scale, W, bh, bv, grad, s1, s2, dtype, W_lo, Wt_lo, apply=True):
    Fp, Hp = W.shape
    L.call("dae_opt_step", opt, lr, momentum, grad_scale, L.ptr(W), L.ptr(bh), L.ptr(bv), L.ptr(grad), L.ptr(s1),
           L.ptr(s2), Fp, Hp, dtype, L.ptr(W_lo), L.ptr(Wt_lo), int(apply), L.current_stream())


def step_stats(rowloss_part, cw, B, triplet, alpha, tri_scalars, nvalid, tile_part=None):
    ncw, Bp = rowloss_part.shape
    stats = torch.zeros(L.STATS_STRIDE, dtype=torch.float32, device=rowloss_part.device)
    L.call("dae_step_stats", L.ptr(rowloss_part), ncw, L.ptr(tile_part), 0 if tile_part is None else tile_part.numel(),
           L.ptr(cw), B, Bp, triplet, alpha, L.ptr(tri_scalars), L.ptr(nvalid), None, None, L.ptr(stats), L.current_stream())
    return stats


def explicit_triplet(h3, B, H, alpha):
    dev = h3.device
    dh3 = torch.zeros_like(h3)
    loss_part = torch.zeros(B, dtype=torch.float32, device=dev)
    tri = torch.zeros(4, dtype=torch.float32, device=dev)
    L.call("dae_explicit_triplet", L.ptr(h3), h3.stride(0), B, H, alpha, L.ptr(dh3), L.ptr(loss_part), L.ptr(tri),
           L.current_stream())
    return dh3, loss_part, tri
