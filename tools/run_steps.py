#!/usr/bin/env python3
"""N plain training steps (no event profiling) -- the workload run under rocprofv3 --pmc.
usage: python tools/run_steps.py [steps] [strategy] [c2|c4|c5] [precision]   (c2: BASELINE configs[1] CSR step; c4: dense fp32 tf-idf, F = 50000; c5: explicit
(org, pos, neg) triplets, three tf-idf CSR blocks, cosine_proximity, 800 triplets = 2400 rows per step; precision: default = what precision='auto' resolves to for
the strategy | bf16 | fp32 | ...)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dae_rnn_news_recommendation_amd import _lib as L
from dae_rnn_news_recommendation_amd.engine import Engine
from dae_rnn_news_recommendation_amd.synthetic import synthetic_csr, synthetic_labels, xavier_uniform
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
strategy = sys.argv[2] if len(sys.argv) > 2 else "batch_all"
cfg = sys.argv[3] if len(sys.argv) > 3 else "c2"
if cfg == "c5":
    strategy = "explicit"
precision = sys.argv[4] if len(sys.argv) > 4 else L.auto_precision(strategy)
F, H = (50000, 1000) if cfg == "c4" else (10000, 500)
m = synthetic_csr(1600, F, nnz_per_row=300 if cfg == "c4" else 200, seed=1, tfidf=(cfg != "c2")); lab = synthetic_labels(1600, seed=1).astype(np.int32)
eng = Engine(F, H, 2400 if cfg == "c5" else 800, dtype=precision, triplet=strategy, learning_rate=0.1, loss_func="cosine_proximity" if cfg == "c5" else "cross_entropy")
if cfg == "c5":
    from scipy import sparse
    eng.upload_csr(sparse.vstack([m, synthetic_csr(1600, F, seed=2, tfidf=True), synthetic_csr(1600, F, seed=3, tfidf=True)]).tocsr())
elif cfg == "c2":
    eng.upload_csr(m)
else:
    eng.upload_dense(np.ascontiguousarray(m.toarray(), dtype=np.float32))
eng.set_params(xavier_uniform(F, H))
stats = torch.zeros((2, 8), device="cuda")
# class-sorted batches, as fit() and bench.py stage them (utils.class_sort_batches): the miner takes its class-range path
rows = [np.arange(b * 800, b * 800 + 800) for b in range(2)]
rows = [r[np.argsort(lab[r], kind="stable")] for r in rows]
if cfg == "c5":        # a batch = the same 800 rows of the three blocks
    rows = [np.concatenate([np.arange(b * 800, b * 800 + 800) + k * 1600 for k in range(3)]) for b in range(2)]
for s in range(steps):
    idx = torch.from_numpy(rows[s % 2].astype(np.int32)).cuda()
    labs = torch.from_numpy(lab[rows[s % 2] % 1600]).cuda()
    eng.train_step(idx, labs if strategy not in ("none", "explicit") else None, stats[s % 2], corr_mode=L.CORR_PHILOX_MASK, seed=1, rng_stream=s, corr_frac=0.3, phase=3)
torch.cuda.synchronize()
st = stats.cpu().numpy()
print("precision", precision, "done cost/ae/triplet", st[0, :3], "mean_n_valid", float(st[:, 5].mean()))
