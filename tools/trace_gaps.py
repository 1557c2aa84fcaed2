#!/usr/bin/env python3
"""Launches per step and inter-kernel gaps from a rocprofv3 --kernel-trace run (rocpd SQLite output).
A "step" is the run of dispatches between two consecutive launches of the step's first kernel (default: the kernel with the
most frequent name among the encode kernels).  Printed: launches per step, per-step wall (first start -> last end), sum of
kernel durations, and the gap end(i) -> start(i+1) between consecutive kernels (negative = the next kernel's first workgroups
started while the previous one drained).
usage: python tools/trace_gaps.py <results.db> [first-kernel-substring]"""
import re
import sqlite3
import sys

import numpy as np

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = db.execute(f"select {name_col}, start, end from kernels order by start").fetchall()
first = sys.argv[2] if len(sys.argv) > 2 else "encode_csr_kernel"


def short(n):
    n = re.sub(r"\(.*", "", n)
    return n.replace("void dae::", "").replace("dae::", "")[:48]


starts = [i for i, r in enumerate(rows) if first in r[0]]
steps = [rows[a:b] for a, b in zip(starts[:-1], starts[1:])]
steps = [s for s in steps if len(s) == np.bincount([len(t) for t in steps]).argmax()]          # the steady-state steps
steps = steps[len(steps) // 4:]                                                                # past the warm-up
n = len(steps[0])
wall = np.array([s[-1][2] - s[0][1] for s in steps]) / 1e3
ksum = np.array([sum(r[2] - r[1] for r in s) for s in steps]) / 1e3
period = np.diff([s[0][1] for s in steps]) / 1e3
gaps = np.array([[s[i + 1][1] - s[i][2] for i in range(n - 1)] for s in steps]) / 1e3
print(f"{len(steps)} steady-state steps of {n} kernel launches each (first kernel: {short(steps[0][0][0])})")
print(f"step period (start to start)  median {np.median(period):7.1f} us")
print(f"first start -> last end       median {np.median(wall):7.1f} us")
print(f"sum of kernel durations       median {np.median(ksum):7.1f} us")
print(f"sum of the {n - 1} gaps             median {np.median(gaps.sum(1)):7.1f} us   (negative: consecutive kernels overlap ramp and tail)")
print(f"last end -> next step's start median {np.median(period - wall[:-1]):7.1f} us   (host launch path between steps)")
print("| after kernel | gap to the next launch, median us | p90 |")
print("|---|---:|---:|")
for i in range(n - 1):
    print(f"| `{short(steps[0][i][0])}` | {np.median(gaps[:, i]):.2f} | {np.percentile(gaps[:, i], 90):.2f} |")
