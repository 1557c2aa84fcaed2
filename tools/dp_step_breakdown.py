#!/usr/bin/env python3
"""Where the data-parallel step form spends its time WITHOUT communication: a one-rank RCCL group on one GPU runs
phase-1 step -> reduce-scatter -> sharded apply -> all-gather -> transpose rebuild, each piece bracketed by events.
usage: python tools/dp_step_breakdown.py [--grad-dtype fp32|bf16] [--precision f16x2|bf16x3|bf16]
(the split modes -- f16x2, the product default, and bf16x3: fp32 gradients, fp32 master rows all-gathered, the shadow images rebuilt per rank in
dp.ShardedExchange's form; the all-reduce form those modes default to is timed beside it)"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as dist
from dae_rnn_news_recommendation_amd import _lib as L, dp
from dae_rnn_news_recommendation_amd.engine import Engine
from dae_rnn_news_recommendation_amd.synthetic import synthetic_csr, synthetic_labels, xavier_uniform
ap = argparse.ArgumentParser(); ap.add_argument("--grad-dtype", default="fp32"); ap.add_argument("--precision", default=L.AUTO_PRECISION); a = ap.parse_args()
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29534")
torch.cuda.set_device(0); dist.init_process_group("nccl", rank=0, world_size=1); dp.quiet_first_collective()
F, H, B = 10000, 500, 800
m = synthetic_csr(1600, F, seed=1); lab = synthetic_labels(1600, seed=1).astype(np.int32)
eng = Engine(F, H, B, dtype=a.precision, triplet="batch_all", learning_rate=0.1, dp_world=1)
eng.upload_csr(m); eng.set_params(xavier_uniform(F, H))
ex = dp.ShardedExchange(eng, grad_dtype=a.grad_dtype)      # (the split mode's default is dp.AllReduceExchange: timed below beside this form)
idx = torch.arange(B, dtype=torch.int32, device="cuda"); labs = torch.from_numpy(lab[:B]).cuda(); stats = torch.zeros(8, device="cuda")
kw = dict(corr_mode=L.CORR_PHILOX_MASK, seed=1, rng_stream=0, corr_frac=0.3)
def timed(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    import time; t0 = time.perf_counter(); e0.record()
    for _ in range(n): fn()
    e1.record(); host = (time.perf_counter() - t0) / n * 1e6; torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n, host
Hp, c = eng.Hp, eng.chunk_rows
gw = eng.grad[:ex.n_w]; bias = eng.grad[eng.Fp * Hp:eng.Fp * Hp + Hp + eng.Fp]
if eng.x3:          # the split modes (f16x2 = the product default, bf16x3): fp32 gradients in, fp32-accurate weights out
    my_w = torch.zeros((c, Hp), dtype=torch.float32, device="cuda")
    exa = dp.AllReduceExchange(eng, buckets=1)
    exb = dp.AllReduceExchange(eng, buckets=4)       # torch.distributed form of the bands (25 us of host time per collective call)
    nx1 = dp.NativeAllReduceExchange(eng, buckets=1)               # round 6: the collective in the C ABI, on the step's own stream (dae_dp_exchange)
    nx4 = dp.NativeAllReduceExchange(eng, buckets=4, comm=nx1.comm)
    print(f"native communicator: RCCL from {nx1.comm.library}, ranks_seen {nx1.comm.ranks_seen}")
    def step_and3(after_dw):
        eng.train_step(idx, labs, stats, phase=1, **kw); ex.step(grad_scale=1.0, grad_ready_after_dw=after_dw)
    rows = [("fused single-GPU step (phase 3)", lambda: eng.train_step(idx, labs, stats, phase=3, **kw)),
            ("phase-1 step (fp32 gradient to memory)", lambda: eng.train_step(idx, labs, stats, phase=1, **kw)),
            ("reduce_scatter fp32 (1 rank)", lambda: dist.reduce_scatter_tensor(ex.rs_out, gw)),
            ("all_reduce bias (1 rank)", lambda: dist.all_reduce(bias)),
            ("apply_rows (all rows at 1 rank)", lambda: eng.apply_rows(ex.rs_f32, ex.f0, ex.f1, grad_scale=1.0, update_bias=True)),
            ("copy of my fp32 master rows", lambda: my_w.copy_(eng.W_full[:c])),
            ("all_gather of the master rows (1 rank)", lambda: dist.all_gather_into_tensor(eng.W_full.view(-1), my_w.view(-1))),
            ("sync_shadows (four images)", lambda: eng.sync_shadows()),
            ("whole exchange.step (sharded form)", lambda: ex.step(grad_scale=1.0)),
            ("phase-1 step + sharded exchange", lambda: step_and3(False)),
            ("all_reduce of the flat fp32 gradient", lambda: dist.all_reduce(exa.flat)),
            ("dae_plan_apply (whole W + 4 images)", lambda: eng.apply(grad_scale=1.0)),
            ("whole AllReduceExchange.step, 1 bucket", lambda: exa.step(grad_scale=1.0)),
            ("phase-1 step + all-reduce exchange (default)", lambda: (eng.train_step(idx, labs, stats, phase=1, **kw), exa.step(grad_scale=1.0))),
            ("apply in 4 row bands (dae_plan_apply_band)", lambda: (eng.begin_apply(), [eng.apply_band(exb.bounds[k], exb.bounds[k + 1]) for k in range(4)])),
            ("whole AllReduceExchange.step, 4 buckets", lambda: exb.step(grad_scale=1.0)),
            ("phase-1 step + bucketed exchange (4 buckets)", lambda: (eng.train_step(idx, labs, stats, phase=1, **kw), exb.step(grad_scale=1.0))),
            ("NATIVE dae_allreduce_grads (1 rank)", lambda: L.check(eng.lib.dae_allreduce_grads(eng.plan, nx1.comm.handle, L.current_stream()), "ar", eng.lib)),
            ("NATIVE whole dae_dp_exchange, 1 bucket", lambda: nx1.step(grad_scale=1.0)),
            ("phase-1 step + NATIVE exchange, 1 bucket", lambda: (eng.train_step(idx, labs, stats, phase=1, **kw), nx1.step(grad_scale=1.0))),
            ("NATIVE whole dae_dp_exchange, 4 buckets", lambda: nx4.step(grad_scale=1.0)),
            ("phase-1 step + NATIVE exchange, 4 buckets", lambda: (eng.train_step(idx, labs, stats, phase=1, **kw), nx4.step(grad_scale=1.0)))]
    print(f"precision {a.precision}  {'piece':40s} {'GPU us':>9s} {'host us/call':>13s}")
    for name, fn in rows:
        g, h = timed(fn)
        print(f"{name:40s} {g:9.1f} {h:13.1f}")
    dist.destroy_process_group()
    sys.exit(0)
ex3 = dp.ShardedExchange(eng, grad_dtype=a.grad_dtype, packed=False)
exn = dp.ShardedExchange(eng, grad_dtype=a.grad_dtype, packed=True, overlap=False)
def step_and(exch, after_dw):
    eng.train_step(idx, labs, stats, phase=1, **kw); exch.step(grad_scale=1.0, grad_ready_after_dw=after_dw)
rows = [("fused single-GPU step (phase 3)", lambda: eng.train_step(idx, labs, stats, phase=3, **kw)),
        ("phase-1 step (gradient to memory)", lambda: eng.train_step(idx, labs, stats, phase=1, **kw)),
        ("reduce_scatter (1 rank)", lambda: dist.reduce_scatter_tensor(ex.rs_out, gw if a.grad_dtype == "fp32" else gw.to(torch.bfloat16))),
        ("apply_rows_packed (all rows at 1 rank)", lambda: eng.apply_rows_packed(ex.rs_f32, ex.f0, ex.f1, ex.send, ex.bias_off, grad_scale=1.0)),
        ("all_gather of the packed chunk (1 rank)", lambda: dist.all_gather_into_tensor(ex.recv, ex.send)),
        ("dp_unpack (W_lo + Wt_lo + biases)", lambda: eng.dp_unpack(ex.recv, 1, ex.chunk_stride, ex.bias_off, grad_scale=1.0)),
        ("whole exchange.step, packed", lambda: ex.step(grad_scale=1.0)),
        ("phase-1 step + packed exchange", lambda: step_and(ex, False)),
        ("  ... reduce-scatter beside the tail", lambda: step_and(ex, True)),
        ("packed, everything on the step's stream", lambda: exn.step(grad_scale=1.0)),
        ("phase-1 step + that", lambda: step_and(exn, False)),
        ("former form: all_reduce bias (1 rank)", lambda: dist.all_reduce(bias)),
        ("former form: apply_rows", lambda: eng.apply_rows(ex3.rs_f32, ex3.f0, ex3.f1, grad_scale=1.0, update_bias=True)),
        ("former form: copy of my W_lo rows", lambda: ex3.my_lo.copy_(eng.W_lo_full[:c])),
        ("former form: refresh_wt", lambda: eng.refresh_wt()),
        ("former form: whole exchange.step", lambda: ex3.step(grad_scale=1.0)),
        ("former form: phase-1 step + exchange", lambda: step_and(ex3, False))]
print(f"{'piece':40s} {'GPU us':>9s} {'host us/call':>13s}")
for name, fn in rows:
    g, h = timed(fn)
    print(f"{name:40s} {g:9.1f} {h:13.1f}")
dist.destroy_process_group()
