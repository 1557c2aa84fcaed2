#!/usr/bin/env python3
"""Turn one `tools/make_profile_report.sh <tag>` run (gpurun_out/<tag>/) into the committed round artefacts under profiles/:
the bench JSON lines of the five configs, the rocprofv3 kernel-trace and PMC tables, the per-launch HBM traffic JSON bench.py
reads back (stamped with the hash of the kernel sources it was measured on) and a summary.

usage: python tools/build_profile_summary.py gpurun_out/r02 r02 [--traffic-only]
(--traffic-only: just the PMC traffic JSON -- make_profile_report.sh calls it on the GPU box between the PMC passes and the bench
runs, so that the bench lines of the same run can quote the traffic of the kernels they are timing)"""
import json, os, shutil, sys

src, rnd = sys.argv[1], sys.argv[2]
TRAFFIC_ONLY = "--traffic-only" in sys.argv
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(root, "profiles")
# step slot (bench.py `kernels` key) -> kernel symbol prefix in the rocprofv3 tables
SLOT_KERNEL = [("encode_gemm", "encode_csr_kernel"), ("gram", "gram64f_kernel"), ("miner", "batch_all_tile_kernel"),
               ("sym_scale", "sym_scale_kernel"), ("decode_loss", "gemm_decode_loss<unsigned short"), ("dh_gemm", "gemm_nt_pc<unsigned short, 4, 2>"),
               ("dh_finish", "dh_finish_kernel<unsigned short"), ("dw_gemm", "gemm_dw_pc"), ("bias_grads", "step_tail_kernel")]
SLOT_KERNEL_C4 = [("gather", "gather_dense_kernel"), ("encode_gemm", "gemm_nt_w8<1>"), ("encode_finish", "encode_finish_kernel"),
                  ("decode_loss", "gemm_decode_loss"), ("dh_gemm", "gemm_nt_w8<2>"), ("dw_gemm", "gemm_dw")]
NOTE = {"encode_gemm": "corrupt + gather + sparse x~.W (fp32 master W) + bias + act, all images of h, x bit image, x~^T scatter, label statistics",
        "gram": "split 16-bit Gram (3 products) on 64 x 64 tiles over the whole K, one slab", "miner": "batch_all on a 16 x 16 lane grid (FAST pair sweep), positive-triplet count from sorted runs", "sym_scale": "Gs = a/Nv (G + G^T) -> bf16",
        "decode_loss": "128 x 64 tiles: GEMM (f16x2h: h.W_hi + h.W_lo + h_lo.W_hi) + loss + delta2 (two layouts), x from bits", "dh_gemm": "delta2.W_hi + delta2.W_lo + Gs.h, split-K 8",
        "dh_finish": "slab reduction, act', delta1^T, column sums", "dw_gemm": "160 x 128 tiles: dW GEMM (f16x2h: x~^T.[d1_hi ; d1_lo] + d2^T.[h_hi ; h_lo], paired stages) + SGD update of W and the hi + lo images of both 16-bit shadows",
        "bias_grads": "step tail: bias grads + update, statistics, x~^T un-scatter"}


def copy(a, b):
    if os.path.exists(os.path.join(src, a)):
        shutil.copy(os.path.join(src, a), os.path.join(dst, f"{rnd}_{b}"))


for c in (() if TRAFFIC_ONLY else ("c1", "c2", "c3", "c4", "c5")):
    copy(f"bench_{c}.json", f"bench_n1_{c}.json")
copy("bench_under_rocprof.json", "bench_under_rocprof.json")
copy("kernel_stats.md", "rocprofv3_kernel_stats.md"); copy("pmc_counters.md", "rocprofv3_pmc_counters.md")
copy("pmc_counters_c4.md", "rocprofv3_pmc_counters_c4.md"); copy("pmc_counters_c5.md", "rocprofv3_pmc_counters_c5.md"); copy("pmc_counters_c1.md", "rocprofv3_pmc_counters_c1.md"); copy("pmc_counters_c3.md", "rocprofv3_pmc_counters_c3.md"); copy("bench_driver_form.json", "bench_driver_form.json"); copy("host.txt", "host.txt")
copy("kprof.txt", "kprof.txt"); copy("miner_timeline.txt", "miner_timeline.txt"); copy("dp_step_breakdown.txt", "dp_step_breakdown.txt")


def table(path):
    rows = []
    if not os.path.exists(path):
        return rows
    for line in open(path):
        c = [x.strip() for x in line.strip().strip("|").split("|")]
        if len(c) >= 4 and c[0].startswith("`"):
            rows.append(c)
    return rows


def find(d, key):
    for k, v in d.items():
        if k.startswith(key):
            return v
    return None


def load(name):
    p = os.path.join(src, name)
    try:
        return json.loads(open(p).read().strip().splitlines()[-1])
    except Exception:       # noqa: BLE001
        return None


stats = {r[0].strip("`"): float(r[3]) for r in table(os.path.join(src, "kernel_stats.md"))}


def pmc_of(fname):
    out = {}
    for r in table(os.path.join(src, fname)):
        out.setdefault(r[0].strip("`"), {})[r[1]] = float(r[3])
    return out


pmc, pmc4, pmc5 = pmc_of("pmc_counters.md"), pmc_of("pmc_counters_c4.md"), pmc_of("pmc_counters_c5.md")
pmc1, pmc3 = pmc_of("pmc_counters_c1.md"), pmc_of("pmc_counters_c3.md")
traffic = {"_note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), per launch, KB->bytes, FETCH_SIZE doubled (gfx950 "
                    "correction, MI355X_MICROARCH.md); workload = tools/run_steps.py (c2: BASELINE configs[1] step; c4: dense F = 50000)",
           "_source_hash": open(os.path.join(src, "source_hash.txt")).read().strip()}
for cfg, slots, tab in (("c2", SLOT_KERNEL, pmc), ("c4", SLOT_KERNEL_C4, pmc4), ("c5", SLOT_KERNEL, pmc5), ("c1", SLOT_KERNEL, pmc1), ("c3", SLOT_KERNEL, pmc3)):
    traffic[cfg] = {}
    for slot, kern in slots:
        c = find(tab, kern)
        if c and "FETCH_SIZE" in c:
            traffic[cfg][slot] = {"fetch_bytes": int(c["FETCH_SIZE"] * 1024 * 2), "write_bytes": int(c.get("WRITE_SIZE", 0) * 1024)}
# miner: VALU instructions per triplet-lane (SQ_INSTS_VALU counts wave instructions)
m = find(pmc, "batch_all_tile_kernel")
try:
    nv = float(open(os.path.join(src, "run_steps.txt")).read().split("mean_n_valid")[1].split()[0])
except Exception:       # noqa: BLE001
    nv = None
if m and nv and "SQ_INSTS_VALU" in m:
    traffic["miner_valu"] = {"wave_insts_per_launch": m["SQ_INSTS_VALU"], "n_valid": nv, "insts_per_triplet_lane": m["SQ_INSTS_VALU"] * 64.0 / nv,
                             "active_quad_cycles": m.get("SQ_ACTIVE_INST_VALU")}
json.dump(traffic, open(os.path.join(dst, f"{rnd}_pmc_traffic.json"), "w"), indent=1)
if TRAFFIC_ONLY:
    print("wrote", os.path.join(dst, f"{rnd}_pmc_traffic.json"))
    sys.exit(0)

b = load("bench_c2.json")
host = open(os.path.join(src, "host.txt")).read().split("\n")
L = []
L.append(f"# Round {rnd[1:]} -- measured on 1x MI355X (gfx950), ROCm 7.2, host: {host[1].split(':')[-1].strip() if len(host) > 1 else '?'} ({host[0]} hw threads)\n")
L.append("All files of this set come from ONE `tools/make_profile_report.sh` run (one box; boxes of the pool differ by up to 1.4x on the same\n"
         f"binary).  Kernel sources hash `{traffic['_source_hash']}` (bench.source_hash).\n")
L.append(f"\n## Headline (`python bench.py --gpus 1 --steps {b['steps']} --warmup {b['warmup']}`, default config c2 = BASELINE configs[1], precision {b.get('dtype')})\n\n| metric | value |\n|---|---|")
L.append(f"| training samples/s (device-Philox masking, `value`) | **{b['value']:,.0f}** ({1e3 * b['ms_per_step']:.1f} us/step) |")
bd = load("bench_driver_form.json")
if bd:
    L.append(f"| the driver's command, `python bench.py --gpus 1 --steps 20 --warmup 5` (`value` over exactly 20 steps) | {bd['value']:,.0f} ({1e3 * bd['ms_per_step']:.1f} us/step"
             + (f"; >= 50 ms `long_run` of the same process: {1e3 * bd['long_run']['ms_per_step']:.1f} us/step" if bd.get("long_run") else "") + ") |")
f = b.get("fit", {})
if "philox" in f: L.append(f"| through `DenoisingAutoencoder.fit()` (N x timed epochs / wall, first epoch excluded), Philox | {f['philox']['samples_per_s']:,.0f} |")
if "numpy" in f: L.append(f"| same, reference-exact NumPy legacy stream (native MT19937 continuation, feeder thread) | {f['numpy']['samples_per_s']:,.0f} |")
cb = b.get("cpu_baseline")
if cb and cb.get("value"):
    L.append(f"| CPU baseline: PyTorch-CPU fp32 restatement, literal B^3 batch_all, {cb['cores']} threads | {cb['value']:.1f} samples/s |")
    if cb.get("chunked"): L.append(f"| CPU baseline, chunked (memory-lean) form | {cb['chunked']['samples_per_s']:.1f} samples/s |")
fl = b["final_losses"]
for mode, what in (("f16x2h", "fp16 images; W, h, delta1 hi + lo: inside the 1e-4 gate over 100 steps"), ("f16x2d", "fp16 images; W, delta2 hi + lo"),
                   ("f16x2", "fp16 images, W alone hi + lo: holds 20 steps, leaves 1e-4 at step 37"), ("bf16x3", "split-bf16, three terms: inside the gate"),
                   ("fp32", "exact-fp32 MFMA: inside the gate"), ("bf16", "plain bf16: OUTSIDE the gate")):
    if b.get(mode) and b[mode].get("value"):
        L.append(f"| `precision='{mode}'` ({what}), same K steps | {b[mode]['value']:,.0f} ({1e3 * b[mode]['ms_per_step']:.1f} us/step) |")
L.append(f"| final losses (means over the last epoch's batches) | cost {fl['cost']:.2f}, AE {fl['autoencoder']:.2f}, triplet {fl['triplet']:.4f}, fraction {fl['fraction']:.4f} |")
r = b.get("roofline")
if r:
    both = f" (SURVEY 8(d)-strict accounting; {100 * r['frac_min_bytes']:.1f} % of HBM peak by this data flow's minimum bytes, {100 * r['mfma_frac']:.1f} % of the MFMA peak by dense FLOPs)" if "mfma_frac" in r else ""
    L.append(f"| `roofline` ({r['kernel'].split(' (')[0]}) | {r['achieved']:.0f} {r['unit']} = **{100 * r['frac']:.1f} %** of {r['peak']:.0f}{both}; traffic {r['traffic']} B ({r.get('traffic_source')}) |")
sr = b.get("step_roofline")
if sr: L.append(f"| whole step | {100 * sr['mfma_frac_dense_accounting']:.1f} % of bf16 MFMA peak by dense accounting (10.B.F.H), {100 * sr['hbm_frac_min_bytes']:.1f} % of HBM peak by minimum bytes |")
L.append("\n## The other named configs (one JSON line each in this directory)\n\n| config | samples/s | us/step | fit() Philox | fit() numpy RNG | roofline |\n|---|---:|---:|---:|---:|---|")
for c in ("c1", "c2", "c3", "c4", "c5"):
    x = load(f"bench_{c}.json")
    if not x: continue
    ff = x.get("fit", {})
    rr = x.get("roofline") or {}
    sps = lambda k: (ff.get(k) or {}).get('samples_per_s') or float('nan')         # (c4: the numpy-RNG leg is skipped with a stated reason)
    L.append(f"| {c} ({x.get('dtype')}) | {x['value']:,.0f} | {1e3 * x['ms_per_step']:.1f} | {sps('philox'):,.0f} | "
             f"{sps('numpy'):,.0f} | {rr.get('bound', '')} {100 * rr.get('frac', 0):.1f} % ({rr.get('kernel', '').split(' (')[0]}; traffic {rr.get('traffic')}) |")
L.append("\n## Per-kernel breakdown of one c2 step (HIP events on the step's stream vs rocprofv3 --kernel-trace of the same bench)\n")
L.append("| step slot | kernel symbol | HIP-event avg us | rocprofv3 avg us | roofline (events) | HBM read MB (FETCH_SIZE x2) | HBM write MB | what it does |")
L.append("|---|---|---:|---:|---|---:|---:|---|")
tot_ev = 0.0; tot_rp = 0.0
for slot, kern in SLOT_KERNEL:
    k = b.get("kernels", {}).get(slot)
    if not k: continue
    rp = find(stats, kern) or float("nan"); t = traffic["c2"].get(slot)
    tot_ev += k["avg_us"]; tot_rp += rp
    roof = f"{k['bound']} {100 * k['frac']:.1f} %" if "frac" in k else ""
    if "mfma_frac" in k and k["bound"] == "hbm": roof += f" (mfma {100 * k['mfma_frac']:.1f} %)"
    L.append(f"| {slot} | `{kern}` | {k['avg_us']:.1f} | {rp:.1f} | {roof} | "
             f"{(t['fetch_bytes'] / 1e6 if t else float('nan')):.1f} | {(t['write_bytes'] / 1e6 if t else float('nan')):.1f} | {NOTE[slot]} |")
tb = sum(v["fetch_bytes"] + v["write_bytes"] for v in traffic["c2"].values())
L.append(f"\nSums: {tot_ev:.0f} us by HIP events ({b.get('kernel_timing', 'host wait behind every launch')}), {tot_rp:.0f} us of rocprofv3 kernel time; the un-profiled\n"
         f"step is {1e3 * b['ms_per_step']:.1f} us.  PMC bytes per step: {tb / 1e6:.0f} MB.  The three event forms side by side: `r06_event_forms.txt`.\n")
if "miner_valu" in traffic:
    mv = traffic["miner_valu"]
    L.append(f"Miner VALU accounting (PMC): {mv['wave_insts_per_launch']:.3g} wave instructions per launch for N_valid = {mv['n_valid']:.3g} triplets = "
             f"**{mv['insts_per_triplet_lane']:.1f} VALU instructions per triplet-lane**.\n")
if os.path.exists(os.path.join(src, "dp_step_breakdown.txt")):
    L.append("## Data-parallel step form without communication (one-rank RCCL group, `tools/dp_step_breakdown.py`)\n\n```\n"
             + "".join(l for l in open(os.path.join(src, "dp_step_breakdown.txt")) if "us" in l or "step" in l or "rank" in l) + "```\n")
# MFMA busy share of the dense-input GEMM kernels (c4): tools/pmc_c4_sq.sh, an own PMC pass
if os.path.exists(os.path.join(src, "pmc_counters_c4_sq.md")):
    copy("pmc_counters_c4_sq.md", "rocprofv3_pmc_counters_c4_sq.md")
    sq4 = pmc_of("pmc_counters_c4_sq.md")
    rows = []
    for k, v in sq4.items():
        if "SQ_VALU_MFMA_BUSY_CYCLES" in v and v.get("GRBM_GUI_ACTIVE") and v["SQ_VALU_MFMA_BUSY_CYCLES"] > 0:
            cyc = v["GRBM_GUI_ACTIVE"] / 8.0            # the counter is summed over the 8 XCDs
            rows.append((v["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc), k, cyc))
    if rows:
        L.append(f"## MFMA busy share of the dense-input kernels (c4; `{rnd}_rocprofv3_pmc_counters_c4_sq.md`, own PMC pass of `tools/pmc_c4_sq.sh`)\n\n"
                 "`SQ_VALU_MFMA_BUSY_CYCLES` / (1024 SIMDs x kernel cycles), kernel cycles = `GRBM_GUI_ACTIVE` / 8 (summed over the XCDs); the 256-row tiles of "
                 "`gemm_nt_w8` spend 12.5 % of their MFMA work on the 896 -> 1024 row padding:\n\n| kernel | MFMA busy | kernel cycles |\n|---|---:|---:|")
        for share, k, cyc in sorted(rows, reverse=True):
            L.append(f"| `{k}` | {100 * share:.1f} % | {cyc:,.0f} |")
        L.append("")
fb2 = os.path.join(dst, f"{rnd}_bench_n1_c2_fast_box.json")
if os.path.exists(fb2):
    x = json.loads(open(fb2).read().strip().splitlines()[-1])
    L.append(f"## Box variance\n\nThe c2 step of the same kernels on a fast box of the pool, two hours earlier (`{rnd}_bench_n1_c2_fast_box.json`, `--no-fit --no-fp32`): "
             f"**{x['value']:,.0f} samples/s** ({1e3 * x['ms_per_step']:.1f} us/step); per kernel (HIP events): "
             + ", ".join(f"{k} {v['avg_us']:.1f}" for k, v in x.get("kernels", {}).items()) + ".\n")
fb = os.path.join(dst, f"{rnd}_bench_n1_c2_fastbox.json")
if os.path.exists(fb):
    x = json.loads(open(fb).read().strip().splitlines()[-1])
    L.append(f"## Box variance\n\nThe same kernels (source hash above) on one of the fast boxes of the pool, 20 minutes earlier: **{x['value']:,.0f} samples/s** "
             f"({1e3 * x['ms_per_step']:.1f} us/step), `{rnd}_bench_n1_c2_fastbox.json` / `{rnd}_kprof_fastbox.txt`.  That line was produced by the bench.py revision before "
             "the per-epoch staging fix: its `fit.numpy` figure (0.71 M) shows the bug fixed afterwards -- torch's parallel pinned copy woke 128 OpenMP workers per "
             "epoch and the container's 16-CPU cgroup quota throttled the whole process; with the single-threaded staging the reference-exact RNG mode "
             "runs at the Philox rate (table above).\n")
_mt = f"`{rnd}_miner_timeline.txt`" if os.path.exists(os.path.join(dst, f"{rnd}_miner_timeline.txt")) else "`r03_miner_timeline.txt` (the probe build was not re-run since)"
L.append(f"PMC detail: `{rnd}_rocprofv3_pmc_counters.md` (c4: `{rnd}_rocprofv3_pmc_counters_c4.md`); per-workgroup timeline of the miner: {_mt};\n"
         f"`tools/kprof.py` output: `{rnd}_kprof.txt`; round-6 evidence files: `{rnd}_curve_modes.txt` (which mode holds which curve), `{rnd}_curve_tests.txt`, `{rnd}_decode_ast.txt` (the persistent decode kernel: built, slower), `{rnd}_region_trace.txt`.\n")
open(os.path.join(dst, f"{rnd}_summary.md"), "w").write("\n".join(L))
print("\n".join(L))
