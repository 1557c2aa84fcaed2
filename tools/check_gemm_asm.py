#!/usr/bin/env python3
"""Static check of the hand-placed LDS waits in the GEMM K loops (csrc/dae_gemm.hip).

The fragment reads are inline-asm ds_read_b128 and the waits are inline-asm `s_waitcnt lgkmcnt(N)`; hipcc's own
scoreboard does not see them, and register-only MFMAs may be moved across an asm wait.  This script compiles the
file to gfx950 assembly and, for every basic block that contains MFMAs fed by ds_read_b128 / ds_read_b64_tr_b16 results, replays the
in-order LDS return rule: after `s_waitcnt lgkmcnt(N)` all but the youngest N reads have landed.  Every MFMA
source register must come from a read that has landed.  Exit code 1 on any violation.

usage: python tools/check_gemm_asm.py [path/to/dae_gemm.hip]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "dae_rnn_news_recommendation_amd", "csrc", "dae_gemm.hip")


def regs(tok):
    m = re.match(r"[va]\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"[va](\d+)$", tok)
    return {int(m.group(1))} if m else set()


def main():
    with tempfile.TemporaryDirectory() as d:
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-S",
                               "--cuda-device-only", src, "-o", os.path.join(d, "k.s")], stderr=subprocess.DEVNULL)
        text = open(os.path.join(d, "k.s")).read()
    bad = 0
    checked = 0
    func = "?"
    pending = []          # list of (dest regs) for ds_read_b128 in issue order
    landed = 0            # number of reads known complete
    for line in text.split("\n"):
        t = line.strip()
        if re.match(r"_ZN3dae\w+:", t):
            func = t.split(":")[0]
            pending, landed = [], 0
            continue
        if t.startswith(".LBB") or t.startswith("s_barrier"):
            # a new block: conservatively forget nothing (reads issued in a predecessor stay pending in order)
            continue
        tok = t.replace(",", " ").split()
        if not tok:
            continue
        if tok[0] in ("ds_read_b128", "ds_read_b64_tr_b16"):      # (the transposing reads of gemm_dw_pc<TRA>: two 64-bit halves per A fragment)
            pending.append(regs(tok[1]))
        elif tok[0] == "s_waitcnt":
            m = re.search(r"lgkmcnt\((\d+)\)", t)
            if m:
                landed = max(landed, len(pending) - int(m.group(1)))
            elif "vmcnt" not in t and "expcnt" not in t:
                landed = len(pending)
        elif tok[0].startswith("v_mfma"):
            srcs = regs(tok[2]) | regs(tok[3])
            for i, dst in enumerate(pending):
                if dst & srcs:
                    # only the most recent producer of these registers matters
                    last = max(j for j, dd in enumerate(pending) if dd & srcs & dst)
                    if i == last:
                        checked += 1
                        if i >= landed:
                            bad += 1
                            print(f"VIOLATION in {func}: {t}  (needs read #{i}, only {landed} landed of {len(pending)})")
    print(f"checked {checked} MFMA operand/read pairs, {bad} violations")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
