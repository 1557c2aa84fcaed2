#!/usr/bin/env python3
"""bench.py -- training samples/sec of the DAE hot path on MI355X (BASELINE.json's metric).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

A "step" is one mini-batch pass of the hot path (corrupt+gather -> encode -> batch_all miner -> decode +
loss -> backward GEMMs -> SGD update) over B=800 rows of the HBM-resident synthetic train set; every 10th
step also pays the per-epoch host work (new permutation; in --rng numpy mode the reference-exact keep-bit
draw).  Workload at N=1 = BASELINE.json configs[1]: synthetic 8000x10000 binary CSR (~200 nnz/row),
compress_factor 20 (H=500), batch_all, bf16 MFMA operands, masking 0.3, SGD lr 0.1.  N>1: weak scaling --
every rank owns its own 8000-row shard and a B=800 local batch (configs[2]'s sharding), gradients are
all-reduced with RCCL every step, mining is per rank.

Prints ONE JSON line (rank 0).  `roofline` comes from HIP events recorded on the step's stream around each
kernel (dae_plan_profile), in a second pass so that `value` is never measured with profiling on;
`cpu_baseline` times the NumPy oracle ("port" of the reference arithmetic; TF 1.12 cannot run here) on a
bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0      # dense bf16 MFMA, MI355X_MICROARCH.md
PEAK_F32_MFMA_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--rows", type=int, default=8000)
    ap.add_argument("--features", type=int, default=10000)
    ap.add_argument("--compress-factor", type=int, default=20)
    ap.add_argument("--batch", type=int, default=800)
    ap.add_argument("--strategy", default="batch_all", choices=["batch_all", "batch_hard", "none"])
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--rng", default="philox", choices=["philox", "numpy"])
    ap.add_argument("--profile-steps", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend for N>1 (nccl == RCCL; gloo only to exercise the N>1 code path on one GPU)")
    ap.add_argument("--single-device", action="store_true", help="testing aid: every rank uses cuda:0")
    return ap.parse_args()


class Runner:
    """Epoch/step scheduler equal to DenoisingAutoencoder._run_train_step, minus printing."""

    def __init__(self, a, rank, world):
        import torch
        from dae_rnn_news_recommendation_amd import _lib as L
        from dae_rnn_news_recommendation_amd.engine import Engine
        from dae_rnn_news_recommendation_amd.synthetic import synthetic_csr, synthetic_labels, xavier_uniform
        self.L, self.torch, self.a, self.rank, self.world = L, torch, a, rank, world
        F, H = a.features, a.features // a.compress_factor
        self.F, self.H, self.B, self.N = F, H, a.batch, a.rows
        self.m = synthetic_csr(a.rows, F, nnz_per_row=200, seed=1234 + rank)
        self.labels = synthetic_labels(a.rows, kind="category", seed=1234 + rank).astype(np.int32)
        self.eng = Engine(F, H, a.batch, dtype=a.precision, enc_act="sigmoid", dec_act="sigmoid", loss_func="cross_entropy",
                          opt="gradient_descent", learning_rate=0.1, alpha=1.0, triplet=a.strategy)
        self.eng.upload_csr(self.m)
        self.eng.set_params(xavier_uniform(F, H, seed=42))
        self.nb = -(-a.rows // a.batch)
        self.stats = torch.zeros((self.nb, L.STATS_STRIDE), dtype=torch.float32, device=self.eng.device)
        self.step_i = 0
        self.epoch = 0
        np.random.seed(0)
        self._prep_epoch()

    def _prep_epoch(self):
        torch, L = self.torch, self.L
        from dae_rnn_news_recommendation_amd.autoencoder import utils
        if self.a.rng == "numpy":
            keep = utils.masking_keep(self.m.nnz, 0.3)
            self.bits = torch.from_numpy(utils.pack_keep_bits(keep).view(np.int32)).to(self.eng.device, non_blocking=True)
            self.plan = dict(corr_mode=L.CORR_KEEPBITS, keep_bits=self.bits)
        else:
            self.plan = dict(corr_mode=L.CORR_PHILOX_MASK, seed=1234, rng_stream=self.epoch, corr_frac=0.3)
        order = utils.epoch_permutation(self.N)
        self.order = torch.from_numpy(order.astype(np.int32)).to(self.eng.device, non_blocking=True)
        self.lab = torch.from_numpy(self.labels[order]).to(self.eng.device, non_blocking=True)

    def step(self):
        b = self.step_i % self.nb
        if b == 0 and self.step_i > 0:
            self.epoch += 1
            self._prep_epoch()
        lo = b * self.B
        hi = min(self.N, lo + self.B)
        rows, labs = self.order[lo:hi], (self.lab[lo:hi] if self.a.strategy != "none" else None)
        if self.world > 1:
            from dae_rnn_news_recommendation_amd import dp
            self.eng.train_step(rows, labs, self.stats[b], phase=1, **self.plan)
            dp.allreduce_sum_(self.eng.grad)
            self.eng.apply(grad_scale=1.0 / self.world)
        else:
            self.eng.train_step(rows, labs, self.stats[b], phase=3, **self.plan)
        self.step_i += 1


def cpu_baseline(a):
    """The NumPy oracle (restated reference arithmetic) on a bounded sample: ONE step of the same workload."""
    import oracle as O
    from threadpoolctl import threadpool_limits
    from dae_rnn_news_recommendation_amd.synthetic import synthetic_csr, synthetic_labels, xavier_uniform
    F, H, B = a.features, a.features // a.compress_factor, a.batch
    m = synthetic_csr(2 * B, F, nnz_per_row=200, seed=1234)
    lab = synthetic_labels(2 * B, seed=1234)
    W = xavier_uniform(F, H, seed=42); bh = np.zeros(H, np.float32); bv = np.zeros(F, np.float32)
    np.random.seed(0)
    threads = os.cpu_count() or 1
    with threadpool_limits(limits=threads):
        t0 = time.time()
        xc = O.masking_noise(m, 0.3)
        idx = O.gen_batches_index(2 * B, B)[0]
        r = O.forward_backward(W, bh, bv, m[idx], xc[idx], lab[idx], triplet_strategy=a.strategy, alpha=1.0, dt=np.float32)
        st = O.OptState("gradient_descent", [W.shape, bh.shape, bv.shape])
        O.opt_apply(st, [W, bh, bv], [r["dW"], r["dbh"], r["dbv"]], 0.1)
        dt = time.time() - t0
    return {"value": B / dt, "unit": "samples/s", "cores": threads, "kind": "port",
            "sample": f"1 step of the same workload (B={B}, {F}x{H}, {a.strategy}, fp32 NumPy oracle incl. masking+shuffle "
                      f"of a {2 * B}-row set); BLAS GEMMs use {threads} threads, the B^3 miner sweep is single-threaded NumPy; "
                      f"{dt:.1f} s",
            "seconds": dt, "cost": float(r["cost"])}


def committed_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc pass (profiles/*_pmc_traffic.json; FETCH_SIZE is
    doubled per the gfx950 correction of MI355X_MICROARCH.md).  PMC counters cannot be read from inside this process, so
    this is the last committed measurement of the same workload, or None."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))
    if not files:
        return None
    try:
        t = json.load(open(files[-1])).get(kernel)
        return None if t is None else t["fetch_bytes"] + t["write_bytes"]
    except Exception:
        return None


def main():
    a = parse()
    import torch
    from dae_rnn_news_recommendation_amd import dp
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1:
        if a.single_device:
            os.environ["LOCAL_RANK"] = "0"
        dp.init_from_env(a.backend)
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    else:
        torch.cuda.set_device(0)
    assert a.gpus == world, f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
    run = Runner(a, rank, world)

    for _ in range(a.warmup):
        run.step()
    dp.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        run.step()
    torch.cuda.synchronize(); dp.barrier(); torch.cuda.synchronize()
    dt = dp.allreduce_max_float(time.perf_counter() - t0)
    last = run.stats.cpu().numpy()
    value = a.steps * a.batch * world / dt

    out = {
        "metric": "training samples/sec (8000x10000 batch_all)", "value": value, "unit": "samples/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.precision, "data": "synthetic",
        "config": {"workload": f"synthetic {a.rows}x{a.features} binary CSR (~200 nnz/row) per GPU, compress_factor "
                               f"{a.compress_factor} (H={a.features // a.compress_factor}), B={a.batch}/GPU, {a.strategy}, "
                               f"masking 0.3, cross_entropy, SGD lr 0.1, {a.precision} MFMA operands + fp32 accumulate/master "
                               f"weights; BASELINE.json configs[1]" + ("" if world == 1 else " sharded as configs[2] (weak)"),
                   "global_batch": a.batch * world, "parallelism": f"dp{world}", "rng": a.rng,
                   "collective": None if world == 1 else "RCCL all-reduce of the flat fp32 gradient per step"},
        "final_losses": {"cost": float(last[:, 0].mean()), "autoencoder": float(last[:, 1].mean()),
                         "triplet": float(last[:, 2].mean()), "fraction": float(last[:, 3].mean()),
                         "note": "means over the last epoch's batches, as the reference prints them (autoencoder.py:283-294)"},
    }

    if rank == 0 and not a.no_roofline:
        eng = run.eng
        eng.profile(True)
        for _ in range(a.profile_steps):
            if world == 1:
                run.step()
            else:          # profile the local step only (no collective inside the event brackets)
                run.eng.train_step(run.order[:a.batch], run.lab[:a.batch] if a.strategy != "none" else None, run.stats[0],
                                   phase=1, **run.plan)
        prof = eng.profile_read()
        eng.profile(False)
        B, F, H = a.batch, a.features, a.features // a.compress_factor
        flops = {"encode_gemm": 2.0 * B * F * H, "decode_loss": 2.0 * B * F * H, "dh_gemm": 2.0 * B * F * H + 2.0 * B * B * H,
                 "dw_gemm": 4.0 * B * F * H, "gram": 2.0 * B * B * H}
        kern = {}
        tot = sum(ms for ms, n in prof.values())
        for k, (ms, n) in prof.items():
            if n == 0:
                continue
            us = 1e3 * ms / n
            e = {"avg_us": us, "launches_per_step": n / a.profile_steps, "time_share": ms / tot if tot else 0.0}
            if k in flops:
                peak = PEAK_F32_MFMA_TFLOPS if a.precision == "fp32" else PEAK_BF16_TFLOPS   # bf16 mode: the Gram matrix is a split-bf16 GEMM
                e.update(bound="mfma", achieved_tflops=flops[k] / (us * 1e-6) / 1e12, peak_tflops=peak)
                e["frac"] = e["achieved_tflops"] / peak
            kern[k] = e
        out["kernels"] = kern
        out["profiled_step_us"] = 1e3 * tot / a.profile_steps
        # headline roofline: the fused encode GEMM named by BASELINE.json's north_star
        e = kern.get("encode_gemm")
        if e:
            out["roofline"] = {"kernel": "encode_gemm (gemm_nt_pc<bf16, 4, ENCODE>: x~[BxF].W[FxH], split-K 8, 8-wave producer/consumer)", "bound": "mfma",
                               "achieved": e["achieved_tflops"], "peak": e["peak_tflops"], "unit": "TFLOP/s", "frac": e["frac"],
                               "traffic": committed_traffic("encode_gemm"),
                               "algorithmic": f"2*B*F*H = {2.0 * B * F * H / 1e9:.2f} GFLOP per launch (dense accounting)"}
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(a)
        out["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist
        dp.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
