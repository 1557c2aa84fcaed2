#!/usr/bin/env python3
"""Shader-clock breakdown of the GEMM K loop (dae_gemm_trace) for the step's GEMM shapes.
usage: python tools/gemm_trace.py [--nst 2]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dae_rnn_news_recommendation_amd import _lib as L

ap = argparse.ArgumentParser(); ap.add_argument("--nst", type=int, default=2); a = ap.parse_args()
L.load()
SHAPES = {"encode (896x512xK10112, 8 slices)": (896, 512, 10112, 0, 8),
          "dW (10112x512xK896+896)": (10112, 512, 896, 896, 1),
          "decode-shape (896x10112xK512)": (896, 10112, 512, 0, 1),
          "gram (896x896xK1536, 4 slices)": (896, 896, 1536, 0, 4)}
for name, (M, N, K0, K1, splits) in SHAPES.items():
    g = torch.Generator(device="cuda").manual_seed(1)
    A0 = torch.randn((M, K0), device="cuda", generator=g).to(torch.bfloat16); B0 = torch.randn((N, K0), device="cuda", generator=g).to(torch.bfloat16)
    A1 = torch.randn((M, max(K1, 64)), device="cuda", generator=g).to(torch.bfloat16); B1 = torch.randn((N, max(K1, 64)), device="cuda", generator=g).to(torch.bfloat16)
    C = torch.empty((splits, M, N), dtype=torch.float32, device="cuda")
    nblk = 8 * max((M // 128 + 7) // 8 * (N // 128), (N // 128 + 7) // 8 * (M // 128)) if splits == 1 else (M // 128) * (N // 128) * splits
    tr = torch.zeros((nblk + 64, 4, 8), dtype=torch.int64, device="cuda")
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for it in range(3):
        tr.zero_(); torch.cuda.synchronize(); ev0.record()
        L.call("dae_gemm_trace", L.BF16, M, N, L.ptr(A0), A0.stride(0), L.ptr(B0), B0.stride(0), K0, L.ptr(A1) if K1 else None,
               A1.stride(0) if K1 else 0, L.ptr(B1) if K1 else None, B1.stride(0) if K1 else 0, K1, L.ptr(C), N, splits, M * N, a.nst,
               L.ptr(tr), L.current_stream())
        ev1.record(); torch.cuda.synchronize()
    ref = (A0.float() @ B0.float().T) + ((A1.float() @ B1.float().T) if K1 else 0)
    err = float((C.sum(0) - ref).abs().max() / ref.abs().max())
    t = tr.cpu().numpy().astype(np.float64); live = t[:, :, 4].sum(1) > 0; t = t[live]
    it_ = t[:, :, 4].mean(); per = lambda k: t[:, :, k].mean() / it_
    print(f"{name}: blocks {live.sum()} k-iters {it_:.0f} event {1e3 * ev0.elapsed_time(ev1):.1f} us rel.err {err:.1e}")
    print(f"   per K iteration (shader clocks, mean over waves): mfma-half-a {per(0):.0f} | waits {per(1):.0f} | barrier {per(2):.0f} | reads+mfma-half-b+dma {per(3):.0f}"
          f" | loop total {t[:, :, 5].mean() / it_:.0f}")
    print(f"   epilogue {t[:, :, 6].mean():.0f} clk (s_memtime ticks; the counter is per XCD, so only differences within a wave are meaningful)")
