#!/usr/bin/env python3
"""Per-workgroup timeline of the persistent decode kernel (gemm_decode_ast; probe bit 64 of DecodeEpi::dbg: each workgroup leaves 100 MHz timestamps in the
dbv_part buffer): start | ring prologue issued | [A phase done | K loop done | epilogue done] per tile.  usage: python tools/ast_timeline.py [--dbg BITS] [--precision f16x2]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dae_rnn_news_recommendation_amd import _lib as L
from dae_rnn_news_recommendation_amd.engine import Engine
from dae_rnn_news_recommendation_amd.synthetic import synthetic_csr, synthetic_labels, xavier_uniform
ap = argparse.ArgumentParser(); ap.add_argument("--dbg", type=int, default=0); ap.add_argument("--precision", default="f16x2"); a = ap.parse_args()
F, H, B = 10000, 500, 800
m = synthetic_csr(1600, F, seed=1); lab = synthetic_labels(1600, seed=1).astype(np.int32)
eng = Engine(F, H, B, dtype=a.precision, triplet="none", learning_rate=0.1)
eng.upload_csr(m); eng.set_params(xavier_uniform(F, H))
idx = torch.arange(B, dtype=torch.int32, device="cuda"); stats = torch.zeros(8, device="cuda")
kw = dict(corr_mode=L.CORR_PHILOX_MASK, seed=1, rng_stream=0, corr_frac=0.3, phase=3)
for _ in range(5):
    eng.train_step(idx, None, stats, **kw)
eng.lib.dae_set_glds(-500000 - (64 | a.dbg))
rl = eng.buffer("dbv_part", (2 * (eng.Bpm // 128) * eng.Fp,), torch.float32)
rl.zero_(); torch.cuda.synchronize()
eng.train_step(idx, None, stats, **kw)
torch.cuda.synchronize()
t = rl.view(torch.int64)[:512 * 16].cpu().numpy().reshape(512, 16).astype(np.float64)
eng.lib.dae_set_glds(-500000)
nst = (t > 0).sum(axis=1)
t0 = t[:, 0].min()
print(f"dbg={a.dbg}: {int((nst > 0).sum())} workgroups stamped; stamps per workgroup {np.bincount(nst)[1:].tolist()} (index = count - 1)")
us = lambda v: v / 100.0           # 100 MHz -> us
print(f"kernel span (first start -> last stamp): {us(t.max() - t0):.1f} us; workgroup starts spread over {us(t[:, 0].max() - t0):.1f} us")
# stamp order per workgroup: start | ring prologue issued | per tile: [A phase done, only when the tile opens a new panel] K loop done, epilogue done
for k in range(1, 16):
    ok = nst > k
    if ok.sum() == 0:
        break
    d = us(t[ok, k] - t[ok, k - 1])
    print(f"  stamp {k:2d}  n={int(ok.sum()):4d}  since the previous stamp: mean {d.mean():6.2f} us  (min {d.min():6.2f}, max {d.max():6.2f});  reached at {us(t[ok, k] - t0).mean():6.1f} us after the first start (max {us(t[ok, k] - t0).max():6.1f})")
last = np.array([t[i, nst[i] - 1] for i in range(512) if nst[i] > 0])
print(f"  workgroup end: mean {us(last - t0).mean():.1f} us, max {us(last - t0).max():.1f} us")
