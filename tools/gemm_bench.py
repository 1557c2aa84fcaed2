#!/usr/bin/env python3
"""Times dae_gemm_nt on the large split-K shapes of the dense-input configs, 256 x 256 / 8-MFMA-wave kernel vs the 128 x 128 kernels,
interleaved in one process (HIP events, random bf16 operands).  usage: python tools/gemm_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dae_rnn_news_recommendation_amd import _lib as L, ops

lib = L.load()
shapes = [("c4 encode x~.W", 896, 1024, 50048, 0), ("c4 dh delta2.W + Gs.h", 896, 1024, 50048, 896), ("B 1792", 1792, 1024, 50048, 0),
          ("square 4096", 4096, 4096, 4096, 0), ("B 896, H 2048", 896, 2048, 50048, 0)]
for name, M, N, K0, K1 in shapes:
    s = lib.dae_gemm_w8_splits(L.BF16, M, N, K0 + K1)
    g = torch.Generator(device="cuda").manual_seed(1)
    A0 = torch.randn(M, K0, device="cuda", generator=g).to(torch.bfloat16)
    B0 = torch.randn(N, K0, device="cuda", generator=g).to(torch.bfloat16)
    A1 = torch.randn(M, K1, device="cuda", generator=g).to(torch.bfloat16) if K1 else None
    B1 = torch.randn(N, K1, device="cuda", generator=g).to(torch.bfloat16) if K1 else None
    flop = 2.0 * M * N * (K0 + K1)
    res = {}
    for rnd in range(3):
        for mode in ("w8", "128"):
            if mode == "w8" and s == 0:
                continue
            lib.dae_set_glds(-7 if mode == "w8" else -6)
            sp = s if s else 8
            if mode == "128" and M % 128:
                continue
            for _ in range(2):
                ops.gemm_nt(A0, B0, A1, B1, splits=sp)
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(10):
                ops.gemm_nt(A0, B0, A1, B1, splits=sp)
            e1.record(); torch.cuda.synchronize()
            res.setdefault(mode, []).append(e0.elapsed_time(e1) * 1e3 / 10)
    lib.dae_set_glds(-7)
    line = f"{name:24s} M {M} N {N} K {K0}+{K1} splits {s}:"
    for mode, v in res.items():
        us = min(v)
        line += f"  {mode}: {us:7.1f} us = {flop / us / 1e6:6.0f} TFLOP/s ({flop / us / 1e6 / 2500:.3f} of 2.5 PF; median {np.median(v):.1f} us)"
    print(line)
