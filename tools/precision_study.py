#!/usr/bin/env python3
"""Which operands of the three gradient GEMMs must carry more than bf16 for the 20-step loss curve to stay within 1e-4 of the
fp32 arithmetic?  CPU study (torch fp32 on the host cores, no GPU): the c2 training step written out by hand with the bf16
mode's data flow (encode from the fp32 master weights, Gram on fp32 h, every other GEMM operand rounded to bf16 when it is
stored) and a per-operand choice between
    f32    the operand as it is            (what precision='fp32' multiplies)
    bf16   round-to-nearest-even bf16      (what precision='bf16' stores and multiplies)
    split  hi = bf16(x), lo = bf16(x - hi), products hi.hi + hi.lo + lo.hi   (what the Gram GEMM already does)
The element-wise parts (sigmoid, the literal cross-entropy with its 1e-16 guards, its derivative, batch_all) run in fp32 torch.
usage: python tools/precision_study.py [--steps 20] [--rows 8000] [--features 10000] [--batch 800]
Prints, per mode, the largest relative deviation of cost / triplet loss from the all-f32 run over the steps."""
import argparse
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from dae_rnn_news_recommendation_amd.synthetic import synthetic_csr, synthetic_labels, xavier_uniform  # noqa: E402


def q(x):
    return x.to(torch.bfloat16).float()


def q16(x):
    """fp16 storage with a per-tensor power-of-two scale (max |x| -> ~2^14): what an fp16 operand image with the scale folded into the consumer's
    epilogue would hold (11 significant bits instead of bf16's 8; the scale keeps delta2 / Gs, whose entries are ~1e-6, out of the subnormals)."""
    m = float(x.abs().max())
    if m == 0.0 or not np.isfinite(m):
        return x.clone()
    sc = 2.0 ** np.floor(np.log2(16384.0 / m))
    return (x * sc).to(torch.float16).float() / sc


def parts(x, mode):
    if mode == "f32":
        return [x]
    if mode in ("f16", "f16split"):
        hi = q16(x)
        return [hi] if mode == "f16" else [hi, q16(x - hi)]
    if mode == "f16raw":                                   # fp16 without the per-tensor scale (small entries fall into the subnormals / to zero)
        return [x.to(torch.float16).float()]
    hi = q(x)
    return [hi] if mode == "bf16" else [hi, q(x - hi)]


def mm(a, b, ma, mb):
    """a @ b with the operands in the given storage modes (fp32 accumulation); lo.lo is dropped as the split GEMM drops it."""
    pa, pb = parts(a, ma), parts(b, mb)
    out = None
    for i, x in enumerate(pa):
        for j, y in enumerate(pb):
            if i + j > 1:
                continue
            t = x @ y
            out = t if out is None else out + t
    return out


def batch_all(lab, h, chunk=32):
    """loss, G = d loss / d D, data weights, N_valid  (triplet_loss_utils.py:79-131, anchor chunks)."""
    D = h @ h.t()
    B = D.shape[0]
    eq = lab.unsqueeze(0) == lab.unsqueeze(1)
    ne = ~torch.eye(B, dtype=torch.bool)
    G = torch.zeros_like(D)
    tot = 0.0
    nv = 0.0
    dw = torch.zeros(B)
    for a0 in range(0, B, chunk):
        a1 = min(B, a0 + chunk)
        ap = (eq[a0:a1] & ne[a0:a1]).float()
        an = (~eq[a0:a1]).float()
        T = -D[a0:a1].unsqueeze(2) + D[a0:a1].unsqueeze(1)
        valid = ap.unsqueeze(2) * an.unsqueeze(1)
        tot += float((torch.nn.functional.softplus(T) * valid).sum().double())
        sg = torch.sigmoid(T) * valid
        G[a0:a1] += sg.sum(1) - sg.sum(2)
        nv += float(valid.sum())
        dw[a0:a1] += valid.sum((1, 2))
        dw += valid.sum((0, 1)) + valid.sum((0, 2))
    return tot / (nv + 1e-16), G / (nv + 1e-16), dw, nv


def run(mode, data, labels, W0, steps, B, lr=0.1, alpha=1.0, seed=7, golden=False):
    """mode: dict operand -> 'f32' | 'bf16' | 'split' for h, W, d2, d1, Gs; a key 'op@gemm' (gemm in dec, dh, dw) overrides 'op' in that GEMM.
    golden: batches and corruption exactly as DenoisingAutoencoder.fit(seed=0, rng='numpy') / the reference stage them -- per epoch the keep
    decisions of the whole set from NumPy's legacy global stream (np.random.rand(nnz) >= v in CSR storage order), then the shuffle -- so the run
    is comparable step by step with tests/golden/full_curve_c2.npz (what tests/test_hip_full_curve.py holds the GPU to)."""
    def M(op, gemm):
        return mode.get(f"{op}@{gemm}", mode[op])
    rng = np.random.default_rng(seed)
    N, F = data.shape
    W = torch.from_numpy(W0.copy()); bh = torch.zeros(W.shape[1]); bv = torch.zeros(F)
    order = rng.permutation(N)
    if golden:
        from dae_rnn_news_recommendation_amd.autoencoder import utils as U
        np.random.seed(0)
        nb = -(-N // B)
    out = []
    for s in range(steps):
        if golden:
            if s % nb == 0:                                  # a new epoch: corruption of the whole set, then the permutation (reference order)
                keep_e = np.random.rand(data.nnz) >= 0.3
                corrupted = data.copy(); corrupted.data = corrupted.data * keep_e
                order = U.epoch_permutation(N)
            idx = order[(s % nb) * B:(s % nb) * B + B]
            x = torch.from_numpy(np.asarray(data[idx].todense(), dtype=np.float32))
            xc = torch.from_numpy(np.asarray(corrupted[idx].todense(), dtype=np.float32))
        else:
            idx = order[(s * B) % N:(s * B) % N + B]
            x = torch.from_numpy(np.asarray(data[idx].todense(), dtype=np.float32))
            keep = torch.from_numpy((rng.random(x.shape) >= 0.3).astype(np.float32))
            xc = x * keep
        lab = torch.from_numpy(labels[idx].astype(np.int64))
        # encode from the fp32 master weights (exact products of 0/1 entries), Gram on fp32 h
        a1 = torch.sigmoid(xc @ W + bh)
        sb = torch.sigmoid(bh)
        h = a1 - sb
        tl, G, dw, nv = batch_all(lab, h)
        # decode on stored operands; the loss and d cost / d z2 literally (autograd over the element-wise part only)
        z2 = (mm(h, W.t(), M("h", "dec"), M("W", "dec")) + bv).requires_grad_(True)
        y = torch.sigmoid(z2)
        row = -(x * torch.log(y + 1e-16) + (1.0 - x) * torch.log(1.0 - y + 1e-16)).sum(1)
        ae = (row * dw).sum() / (dw.sum() + 1e-16)
        (d2,) = torch.autograd.grad(ae, [z2])
        Gs = alpha * (G + G.t())
        dh = mm(d2, W, M("d2", "dh"), M("W", "dh")) + mm(Gs, h, M("Gs", "dh"), M("h", "dh"))
        d1 = dh * a1 * (1.0 - a1)
        dW = mm(xc.t(), d1, "f32", M("d1", "dw")) + mm(d2.t(), h, M("d2", "dw"), M("h", "dw"))
        dbh = d1.sum(0) - sb * (1.0 - sb) * dh.sum(0)
        dbv = d2.sum(0)
        W -= lr * dW; bh -= lr * dbh; bv -= lr * dbv
        out.append((float(ae.detach()) + alpha * tl, float(ae.detach()), tl))
    return np.array(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--rows", type=int, default=8000)
    ap.add_argument("--features", type=int, default=10000)
    ap.add_argument("--batch", type=int, default=800)
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--per-gemm", action="store_true", help="second study: which of the three GEMMs (decode, dh, dW) need their operands split")
    ap.add_argument("--per-term", action="store_true",
                    help="third study: split everywhere (Gs bf16), then ONE operand use (operand@gemm) back to plain bf16 -- i.e. one lo.hi / hi.lo "
                         "product term of one GEMM dropped -- and the combinations of the droppable ones (--drop)")
    ap.add_argument("--golden", action="store_true",
                    help="inputs, batches and corruption of tests/golden/make_full_curve.py (the reference-exact legacy RNG order); deviations are "
                         "reported against the frozen float32 reference curve tests/golden/full_curve_c2.npz -- what the GPU test is held to")
    ap.add_argument("--scheme", action="append", default=[], metavar="OP=MODE[,OP=MODE...]",
                    help="fourth study: evaluate exactly these storage schemes, e.g. --scheme h=f16,W=f16split,d2=f16,d1=f16,Gs=f16 (modes f32 | bf16 | split "
                         "| f16 | f16split; OP may be op@gemm; operands not named are f16)")
    ap.add_argument("--drop", action="append", default=[], metavar="OP@GEMM[,OP@GEMM...]",
                    help="with --per-term: evaluate exactly these combinations of dropped terms instead of the single-term sweep")
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    data = synthetic_csr(a.rows, a.features, seed=1234).tocsr()
    labels = synthetic_labels(a.rows, seed=1234)
    W0 = xavier_uniform(a.features, a.features // 20, seed=42).astype(np.float32)
    gold = None
    if a.golden:
        import os
        g = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
        sys.path.insert(0, g)
        import make_full_curve as MF
        data, labels, W0 = MF.inputs()
        data = data.tocsr(); W0 = W0.astype(np.float32)
        G = np.load(os.path.join(g, "full_curve_c2.npz"))
        gold = np.stack([G["cost"].reshape(-1), G["ae"].reshape(-1), G["triplet"].reshape(-1)], axis=1)[:a.steps]
    ops = ("h", "W", "d2", "d1", "Gs")
    modes = {"all f32 (reference arithmetic)": dict.fromkeys(ops, "f32"),
             "all bf16 (precision='bf16')": dict.fromkeys(ops, "bf16"),
             "all split": dict.fromkeys(ops, "split")}
    for o in ops:
        m = dict.fromkeys(ops, "bf16"); m[o] = "split"
        modes[f"bf16, {o} split"] = m
    for pair in (("W", "d2"), ("W", "h"), ("d2", "h"), ("W", "d2", "h"), ("W", "d2", "h", "d1")):
        m = dict.fromkeys(ops, "bf16")
        for o in pair:
            m[o] = "split"
        modes["bf16, " + " + ".join(pair) + " split"] = m
    if a.per_gemm:
        base = dict.fromkeys(ops, "f32")
        modes = {"all f32 (reference arithmetic)": base}
        for gemms in (("dec",), ("dh",), ("dw",), ("dec", "dh"), ("dec", "dw"), ("dh", "dw")):
            m = dict.fromkeys(ops, "split"); m["Gs"] = "bf16"
            for g in gemms:                                  # these GEMMs fall back to plain bf16 operands
                for o in ops:
                    m[f"{o}@{g}"] = "bf16"
            modes["split everywhere but plain bf16 in " + " + ".join(gemms)] = m
    if a.per_term:
        uses = ("h@dec", "W@dec", "d2@dh", "W@dh", "h@dh", "d1@dw", "d2@dw", "h@dw")
        modes = {"all f32 (reference arithmetic)": dict.fromkeys(ops, "f32")}
        combos = [tuple(c.split(",")) for c in a.drop] if a.drop else [(u,) for u in uses]
        for combo in combos:
            m = dict.fromkeys(ops, "split"); m["Gs"] = "bf16"
            for u in combo:
                assert u in uses, u
                m[u] = "bf16"
            modes["split, lo term dropped: " + " + ".join(combo)] = m
    if a.scheme:
        modes = {"all f32 (reference arithmetic)": dict.fromkeys(ops, "f32")}
        for sch in a.scheme:
            m = dict.fromkeys(ops, "f16")
            for kv in sch.split(","):
                k, v = kv.split("=")
                assert v in ("f32", "bf16", "split", "f16", "f16split", "f16raw"), v
                m[k] = v
            modes["scheme " + sch] = m
    ref = None
    for name, m in modes.items():
        t0 = time.time()
        r = run(m, data, labels, W0, a.steps, a.batch, golden=a.golden)
        if ref is None:
            ref = r
            if gold is not None:                                 # the all-f32 replay against the frozen reference curve, then the curve is the reference
                dev = np.abs(r - gold) / np.abs(gold)
                print(f"all-f32 replay vs tests/golden/full_curve_c2.npz: cost max {dev[:, 0].max():.2e}  triplet max {dev[:, 2].max():.2e}", flush=True)
                ref = gold
            print(f"{name}: cost {r[0, 0]:.4f} -> {r[-1, 0]:.4f}, triplet {r[0, 2]:.5f} -> {r[-1, 2]:.5f}   ({time.time() - t0:.0f} s)", flush=True)
            continue
        dev = np.abs(r - ref) / np.abs(ref)
        print(f"{name:42s} cost max {dev[:, 0].max():.2e} (step {dev[:, 0].argmax() + 1})  AE max {dev[:, 1].max():.2e}  "
              f"triplet max {dev[:, 2].max():.2e} (step {dev[:, 2].argmax() + 1})   ({time.time() - t0:.0f} s)", flush=True)


if __name__ == "__main__":
    main()
