#!/usr/bin/env python3
"""Turn one `tools/make_profile_report.sh <tag>` run (gpurun_out/<tag>/) into the committed round artefacts under profiles/:
bench JSON lines, the rocprofv3 kernel-trace and PMC tables, the per-launch HBM traffic JSON bench.py reads back, and a summary.

usage: python tools/build_profile_summary.py gpurun_out/r01c r01"""
import json, os, re, shutil, sys

src, rnd = sys.argv[1], sys.argv[2]
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
SLOT_KERNEL = [("gather", "gather_csr_kernel"), ("encode_gemm", "gemm_nt_pc<unsigned short, 4, 1>"), ("encode_finish", "encode_finish_kernel"),
               ("gram", "gemm_nt_pc<unsigned short, 4, 4>"), ("miner", "batch_all_kernel"), ("sym_scale", "sym_scale_kernel"),
               ("decode_loss", "gemm_decode_loss"), ("dh_gemm", "gemm_nt_pc<unsigned short, 4, 2>"), ("dh_finish", "dh_finish_kernel"),
               ("dw_gemm", "gemm_dw_opt"), ("bias_grads", "step_tail_kernel")]
NOTE = {"gather": "CSR rows -> x~ tile, x bit image, x~^T scatter", "encode_gemm": "8-wave producer/consumer, split-K 8; + label statistics workgroup",
        "encode_finish": "slab reduction, bias, act, h / h^T / split-bf16 images", "gram": "split-bf16 (3 products), split-K 4",
        "miner": "batch_all, pair-packed sweep", "sym_scale": "Gs = a/Nv (G + G^T) -> bf16", "decode_loss": "GEMM + loss + delta2 (two layouts), x from bits",
        "dh_gemm": "delta2.W + Gs.h, split-K 8", "dh_finish": "slab reduction, act', delta1^T, column sums",
        "dw_gemm": "dW GEMM + SGD update of W and both bf16 shadows", "bias_grads": "step tail: bias grads + update, statistics, x~^T un-scatter"}

def copy(a, b):
    if os.path.exists(os.path.join(src, a)):
        shutil.copy(os.path.join(src, a), os.path.join(dst, f"{rnd}_{b}"))

copy("bench.json", "bench_n1.json"); copy("bench_numpy_rng.json", "bench_n1_numpy_rng.json"); copy("bench_none.json", "bench_n1_strategy_none.json")
copy("bench_batch_hard.json", "bench_n1_batch_hard.json"); copy("bench_under_rocprof.json", "bench_under_rocprof.json")
copy("kernel_stats.md", "rocprofv3_kernel_stats.md"); copy("pmc_counters.md", "rocprofv3_pmc_counters.md"); copy("host.txt", "host.txt")
copy("gemm_trace.txt", "gemm_trace.txt"); copy("kprof.txt", "kprof.txt")

def table(path):
    rows = []
    for line in open(path):
        c = [x.strip() for x in line.strip().strip("|").split("|")]
        if len(c) >= 4 and c[0].startswith("`"):
            rows.append(c)
    return rows

stats = {r[0].strip("`"): float(r[3]) for r in table(os.path.join(src, "kernel_stats.md"))}
pmc = {}
for r in table(os.path.join(src, "pmc_counters.md")):
    pmc.setdefault(r[0].strip("`"), {})[r[1]] = float(r[3])

def find(d, key):
    for k, v in d.items():
        if k.startswith(key):
            return v
    return None

traffic = {"_note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), per launch, KB->bytes, FETCH_SIZE doubled (gfx950 "
                    "correction, MI355X_MICROARCH.md); workload = tools/run_steps.py (BASELINE configs[1] step)"}
for slot, kern in SLOT_KERNEL:
    c = find(pmc, kern)
    if c and "FETCH_SIZE" in c:
        traffic[slot] = {"fetch_bytes": int(c["FETCH_SIZE"] * 1024 * 2), "write_bytes": int(c.get("WRITE_SIZE", 0) * 1024)}
json.dump(traffic, open(os.path.join(dst, f"{rnd}_pmc_traffic.json"), "w"), indent=1)

b = json.load(open(os.path.join(src, "bench.json")))
def val(name):
    p = os.path.join(src, name)
    return json.load(open(p)) if os.path.exists(p) else None
bn, b0, bh = val("bench_numpy_rng.json"), val("bench_none.json"), val("bench_batch_hard.json")
host = open(os.path.join(src, "host.txt")).read().split("\n")
L = []
L.append(f"# Round {rnd[1:]} -- measured on 1x MI355X (gfx950), ROCm 7.2, host: {host[1].split(':')[-1].strip() if len(host) > 1 else '?'} ({host[0]} hw threads)\n")
L.append("Workload: BASELINE.json configs[1] -- synthetic 8000x10000 binary CSR (~200 nnz/row), H=500, B=800, batch_all, masking 0.3,\n"
         "cross_entropy, SGD lr 0.1, bf16 MFMA operands + fp32 accumulate/master weights.  Command: `python bench.py --steps 300 --warmup 30`\n"
         "(all files of this set come from ONE `tools/make_profile_report.sh` run, i.e. one box; boxes of the pool differ by up to 1.5x on the\n"
         "memory-bound kernels -- the same binary measured 210 us/step on this box and ~300 us/step on the slowest one seen).\n")
L.append("\n## Headline\n\n| metric | value |\n|---|---|")
L.append(f"| training samples/s (device-Philox masking, `value`) | **{b['value']:,.0f}** ({1e3 * b['ms_per_step']:.1f} us/step) |")
if bn: L.append(f"| same, reference-exact NumPy legacy RNG stream (`--rng numpy`) | {bn['value']:,.0f} ({1e3 * bn['ms_per_step']:.1f} us/step; host RNG per epoch) |")
if b0: L.append(f"| `--strategy none` (BASELINE configs[0] shape on the GPU) | {b0['value']:,.0f} ({1e3 * b0['ms_per_step']:.1f} us/step) |")
if bh: L.append(f"| `--strategy batch_hard` | {bh['value']:,.0f} ({1e3 * bh['ms_per_step']:.1f} us/step) |")
cb = b.get("cpu_baseline")
if cb: L.append(f"| CPU baseline: NumPy oracle ('{cb['kind']}'; TF 1.12 cannot run here), 1 step, {cb['cores']} host threads | {cb['value']:.1f} samples/s ({cb['seconds']:.1f} s/step) |")
fl = b["final_losses"]
L.append(f"| final losses (means over the last epoch's batches) | cost {fl['cost']:.2f}, AE {fl['autoencoder']:.2f}, triplet {fl['triplet']:.4f}, fraction {fl['fraction']:.4f} |")
r = b["roofline"]
L.append(f"| roofline of the fused-encode GEMM (2BFH = 8.0 GFLOP / launch) | {r['achieved']:.0f} TFLOP/s = **{100 * r['frac']:.1f} %** of 2.5 PFLOP/s dense bf16 (HIP events); "
         f"{8.0e9 / (find(stats, 'gemm_nt_pc<unsigned short, 4, 1>') * 1e-6) / 1e12:.0f} TFLOP/s by the rocprofv3 duration |")
L.append("\n## Per-kernel breakdown of one step (HIP events on the step's stream vs rocprofv3 --kernel-trace of the same bench)\n")
L.append("| step slot | kernel symbol | HIP-event avg us | rocprofv3 avg us | MFMA frac of peak (events) | HBM read MB (FETCH_SIZE x2) | HBM write MB | what it does |")
L.append("|---|---|---:|---:|---:|---:|---:|---|")
tot_ev = 0.0; tot_rp = 0.0
for slot, kern in SLOT_KERNEL:
    k = b["kernels"].get(slot)
    if not k: continue
    rp = find(stats, kern); t = traffic.get(slot)
    tot_ev += k["avg_us"]; tot_rp += rp or 0.0
    L.append(f"| {slot} | `{kern}` | {k['avg_us']:.1f} | {rp:.1f} | {('%.1f %%' % (100 * k['frac'])) if 'frac' in k else ''} | "
             f"{(t['fetch_bytes'] / 1e6 if t else float('nan')):.1f} | {(t['write_bytes'] / 1e6 if t else float('nan')):.1f} | {NOTE[slot]} |")
L.append(f"\nSums: {tot_ev:.0f} us with event brackets (each bracket adds a host sync and ~2 us), {tot_rp:.0f} us of rocprofv3 kernel time; the\n"
         f"un-profiled step is {1e3 * b['ms_per_step']:.1f} us -- the stream is back-to-back kernels, there is no launch gap left to remove.\n")
L.append("PMC detail (MFMA busy, wave cycles, waits, LDS bank conflicts per kernel): `%s_rocprofv3_pmc_counters.md`; K-loop phase clocks of the\n"
         "4-wave GEMM kernel: `%s_gemm_trace.txt`; what was tried and what it bought: `%s_experiments.md`.\n" % (rnd, rnd, rnd))
open(os.path.join(dst, f"{rnd}_summary.md"), "w").write("\n".join(L))
print("\n".join(L))
