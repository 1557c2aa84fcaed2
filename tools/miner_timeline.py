#!/usr/bin/env python3
"""Timeline of the batch_all miner's workgroups from a probe build (-DDAE_MINER_PROBE=4; thread 0 stamps the 100 MHz wall clock
at each stage into the plan's role_cnt buffer).  Build the probe library with
  make -C dae_rnn_news_recommendation_amd/csrc BUILD=build_mp4 OUT=../libdae_mp4.so CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-pass-failed -DDAE_MINER_PROBE=4"
usage: python tools/miner_timeline.py --lib dae_rnn_news_recommendation_amd/libdae_mp4.so"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dae_rnn_news_recommendation_amd import _lib as L
ap = argparse.ArgumentParser()
ap.add_argument("--lib", required=True); ap.add_argument("--batch", type=int, default=800)
ap.add_argument("--unsorted", action="store_true", help="rows in set order instead of class-sorted (fit() sorts every batch by label)")
a = ap.parse_args()
L.LIB_PATH = os.path.abspath(a.lib)
from dae_rnn_news_recommendation_amd.engine import Engine
from dae_rnn_news_recommendation_amd.synthetic import synthetic_csr, synthetic_labels, xavier_uniform
F, H, B = 10000, 500, a.batch
m = synthetic_csr(2 * B, F, seed=1); lab = synthetic_labels(2 * B, seed=1).astype(np.int32)
eng = Engine(F, H, B, dtype="bf16", triplet="batch_all", loss_func="cross_entropy", learning_rate=0.1)
eng.upload_csr(m); eng.set_params(xavier_uniform(F, H))
rows = np.arange(B) if a.unsorted else np.argsort(lab[:B], kind="stable")
idx = torch.from_numpy(rows.astype(np.int32)).cuda(); labs = torch.from_numpy(lab[rows]).cuda(); stats = torch.zeros(8, device="cuda")
for _ in range(4):
    eng.train_step(idx, labs, stats, phase=3, corr_mode=L.CORR_PHILOX_MASK, seed=1, rng_stream=0, corr_frac=0.3)
torch.cuda.synchronize()
nb = B
Bp = eng.info()["Bpm"]
t = eng.buffer("role_cnt", (Bp * Bp // 2,), torch.int64)[: nb * 8].cpu().numpy().reshape(nb, 8)
st = t[:, :5].astype(np.float64) * 0.01          # us
t0 = st[:, 0].min()
print(f"workgroups {nb}; kernel span {st[:, 4].max() - t0:.1f} us (first start -> last end)")
names = ["start (rel.)", "prologue: loads+lists+exp", "count: sort + search", "sweeps", "epilogue"]
d = np.stack([st[:, 0] - t0, st[:, 1] - st[:, 0], st[:, 2] - st[:, 1], st[:, 3] - st[:, 2], st[:, 4] - st[:, 3]], 1)
for i, n in enumerate(names):
    q = np.percentile(d[:, i], [0, 10, 50, 90, 100])
    print(f"  {n:28s} min {q[0]:6.2f}  p10 {q[1]:6.2f}  med {q[2]:6.2f}  p90 {q[3]:6.2f}  max {q[4]:6.2f} us")
end = st[:, 4] - t0
print("  end time percentiles:", np.round(np.percentile(end, [10, 50, 90, 99, 100]), 1))
late = np.argsort(st[:, 0])[-40:]
print("  40 latest starters: start", np.round(d[late, 0].min(), 1), "-", np.round(d[late, 0].max(), 1), "us; their sweeps med", np.round(np.median(d[late, 3]), 1))
nP = t[:, 6] >> 32; nN = t[:, 6] & 0xFFFFFFFF
work = (nP * nN).astype(np.float64)
xcc = (t[:, 5] >> 32) & 0xF; hw = t[:, 5] & 0xFFFFFFFF
kind = (t[:, 5] >> 40) & 0xF
print("  sweep kind per workgroup (8/4/2 = pair sweep with that many factors per log, 1 = per-cell, 15 = direct):", {int(k): int(v) for k, v in zip(*np.unique(kind, return_counts=True))})
rng = (t[:, 7] & 0xFFFFFFFF).astype(np.uint32).view(np.float32)
cyc = (t[:, 7] >> 32).astype(np.float64)
print("  shader clock during the sweeps (s_memtime cycles / wall): med %.2f GHz" % np.median(cyc / np.maximum(d[:, 3], 1e-3) / 1e3))
print("  D-row range (max - min over the anchor's positives and negatives): min %.2f  p10 %.2f  med %.2f  p90 %.2f  max %.2f" % tuple(np.percentile(rng, [0, 10, 50, 90, 100])))
cu = (hw >> 8) & 0xF; se = (hw >> 13) & 0x7
key = xcc * 1000 + se * 16 + cu
u, c = np.unique(key, return_counts=True)
print(f"  distinct (xcc,se,cu) = {len(u)}; workgroups per CU: min {c.min()} max {c.max()}; histogram {np.bincount(c)}")
# does the dispatcher place block b on the CU of block b % 256?  (blocks of one CU: residues mod 256, mod 8)
same256 = sum(len(set(np.nonzero(key == k)[0] % 256)) == 1 for k in u)
print(f"  CUs whose blocks all share b % 256: {same256} of {len(u)};  blocks of the first CUs:", [list(np.nonzero(key == k)[0]) for k in u[:6]])
work_cu = np.array([work[key == k].sum() for k in u]); end_cu = np.array([end[key == k].max() for k in u])
print("  cells per CU: min %.0fk med %.0fk max %.0fk; CU end time vs cells corr %.2f; end of 4-block CUs med %.1f, of 3-block CUs med %.1f" % (
    work_cu.min() / 1e3, np.median(work_cu) / 1e3, work_cu.max() / 1e3, np.corrcoef(work_cu, end_cu)[0, 1], np.median(end_cu[c == 4]) if (c == 4).any() else 0, np.median(end_cu[c == 3]) if (c == 3).any() else 0))

print("  sweep us per 1e5 triplets (med):", np.round(np.median(d[:, 3] / np.maximum(work, 1) * 1e5), 2))
