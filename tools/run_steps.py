#!/usr/bin/env python3
"""N plain training steps of BASELINE config 2 (no event profiling) -- the workload run under rocprofv3 --pmc."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dae_rnn_news_recommendation_amd import _lib as L
from dae_rnn_news_recommendation_amd.engine import Engine
from dae_rnn_news_recommendation_amd.synthetic import synthetic_csr, synthetic_labels, xavier_uniform
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
strategy = sys.argv[2] if len(sys.argv) > 2 else "batch_all"
m = synthetic_csr(1600, 10000, seed=1); lab = synthetic_labels(1600, seed=1).astype(np.int32)
eng = Engine(10000, 500, 800, dtype="bf16", triplet=strategy, learning_rate=0.1)
eng.upload_csr(m); eng.set_params(xavier_uniform(10000, 500))
stats = torch.zeros(8, device="cuda")
for s in range(steps):
    idx = torch.arange((s % 2) * 800, (s % 2) * 800 + 800, dtype=torch.int32, device="cuda")
    labs = torch.from_numpy(lab[(s % 2) * 800:(s % 2) * 800 + 800]).cuda()
    eng.train_step(idx, labs if strategy != "none" else None, stats, corr_mode=L.CORR_PHILOX_MASK, seed=1, rng_stream=s, corr_frac=0.3, phase=3)
torch.cuda.synchronize()
print("done", stats.cpu().numpy()[:3])
