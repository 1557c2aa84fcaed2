#!/usr/bin/env python3
"""Per-kernel HIP-event timings of the training step (dae_plan_profile) for quick A/B experiments.
usage: python tools/kprof.py [--strategy batch_all] [--precision bf16] [--steps 20]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dae_rnn_news_recommendation_amd import _lib as L
from dae_rnn_news_recommendation_amd.engine import Engine
from dae_rnn_news_recommendation_amd.synthetic import synthetic_csr, synthetic_labels, xavier_uniform

ap = argparse.ArgumentParser()
ap.add_argument("--strategy", default="batch_all"); ap.add_argument("--precision", default="bf16")
ap.add_argument("--steps", type=int, default=20); ap.add_argument("--rows", type=int, default=8000)
ap.add_argument("--features", type=int, default=10000); ap.add_argument("--hidden", type=int, default=500)
ap.add_argument("--batch", type=int, default=800); ap.add_argument("--loss", default="cross_entropy")
ap.add_argument("--enc-splits", type=int, default=0); ap.add_argument("--tag", default="")
ap.add_argument("--nst", type=int, default=-1); ap.add_argument("--phase", type=int, default=3); ap.add_argument("--gram-splits", type=int, default=0)
ap.add_argument("--corr", default="philox", choices=["philox", "bits", "none"])
ap.add_argument("--opt", action="append", default=[], help="plan option name=value (dae_plan_set_option), repeatable")
ap.add_argument("--unsorted", action="store_true", help="batches in row order instead of class-sorted")
ap.add_argument("--lib", default="", help="alternative libdae_hip build (probe variants; tools only)")
ap.add_argument("--lib-f16", default="", help="alternative libdae_hip_f16 build (A/B of two builds on one box; tools only)")
ap.add_argument("--queued", action="store_true", help="queued event pairs (dae_plan_profile mode 2): no host wait between the launches of a step")
ap.add_argument("--stamps", action="store_true", help="pairs stamped by the dispatch itself (dae_plan_profile mode 3): the kernel's own duration")
ap.add_argument("--glds", type=int, action="append", default=[], help="dae_set_glds code(s), e.g. -8 = dW on the producer/consumer kernel")
a = ap.parse_args()
if a.lib:
    L.LIB_PATH = os.path.abspath(a.lib)
if a.lib_f16:
    L.LIB_PATHS["f16"] = os.path.abspath(a.lib_f16)
if a.nst >= 0:
    L.load().dae_set_glds(a.nst)
m = synthetic_csr(a.rows, a.features, seed=1); lab = synthetic_labels(a.rows, seed=1).astype(np.int32)
eng = Engine(a.features, a.hidden, a.batch, dtype=a.precision, triplet=a.strategy, loss_func=a.loss, learning_rate=0.1,
             encode_splits=a.enc_splits, dh_splits=a.enc_splits, gram_splits=a.gram_splits)
for code in a.glds:          # process-wide switches live in the library build this engine runs on
    eng.lib.dae_set_glds(code)
for o in a.opt:
    k, v = o.split("="); eng.set_option(k, int(v))
eng.upload_csr(m); eng.set_params(xavier_uniform(a.features, a.hidden))
# the steps cycle through the rows // batch different batches of the set (fresh rows every step, as in training)
nb = max(1, a.rows // a.batch)
rows_b = [np.arange(b * a.batch, (b + 1) * a.batch) for b in range(nb)]
if not a.unsorted:        # class-sorted batches, as fit() stages them (utils.class_sort_batches)
    rows_b = [r[np.argsort(lab[r], kind="stable")] for r in rows_b]
idxs = [torch.from_numpy(r.astype(np.int32)).cuda() for r in rows_b]
labss = [torch.from_numpy(lab[r]).cuda() for r in rows_b]
stats = torch.zeros(8, device="cuda")
_step = [0]
def one_step():
    b = _step[0] % nb; _step[0] += 1
    eng.train_step(idxs[b], labss[b] if a.strategy != "none" else None, stats, phase=a.phase, **kw)
kw = dict(corr_mode=L.CORR_PHILOX_MASK, seed=1, rng_stream=0, corr_frac=0.3)
if a.corr == "none":
    kw = dict()
elif a.corr == "bits":
    keep = np.random.default_rng(0).random(m.nnz) >= 0.3
    kb = np.packbits(keep, bitorder="little"); kb = np.concatenate([kb, np.zeros((-len(kb)) % 4, np.uint8)]).view(np.int32)
    kw = dict(corr_mode=L.CORR_KEEPBITS, keep_bits=torch.from_numpy(kb.copy()).cuda())
for _ in range(5):
    one_step()
torch.cuda.synchronize()
eng.profile(True, queued=a.queued, stamps=a.stamps)
for _ in range(a.steps):
    one_step()
prof = eng.profile_read(); eng.profile(False)
# the un-profiled step (kernels back to back, side streams allowed): events around a run of steps
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
for _ in range(10):
    one_step()
torch.cuda.synchronize(); e0.record()
for _ in range(200):
    one_step()
e1.record(); torch.cuda.synchronize()
free_us = 1e3 * e0.elapsed_time(e1) / 200
tot = sum(ms for ms, n in prof.values())
print(f"== {'stamps' if a.stamps else 'queued' if a.queued else 'sync'} events corr={a.corr} {a.tag} glds={a.glds} {a.opt} {a.strategy} {a.precision} total {1e3*tot/a.steps:.1f} us/step (bracketed kernels), {free_us:.1f} us/step un-profiled  info={eng.info()}")
for k, (ms, n) in prof.items():
    if n: print(f"   {k:18s} {1e3*ms/n:9.1f} us  x{n/a.steps:.0f}")
