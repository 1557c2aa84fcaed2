#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace run (rocpd SQLite output) as a per-kernel table:
calls, total / average / min / max duration (us) and share of GPU kernel time.
usage: python tools/rocprof_summary.py <results.db> [> profiles/xxx_kernel_stats.md]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = db.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                  f"from kernels group by {name_col} order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)


def short(n):
    n = re.sub(r"\(.*", "", n)
    n = n.replace("void dae::", "").replace("dae::", "")
    return n[:70]


print(f"| kernel | calls | total us | avg us | min us | max us | share |")
print("|---|---:|---:|---:|---:|---:|---:|")
for n, c, s, a, mn, mx in rows:
    print(f"| `{short(n)}` | {c} | {s/1e3:.1f} | {a/1e3:.2f} | {mn/1e3:.2f} | {mx/1e3:.2f} | {100*s/tot:.1f}% |")
print(f"\ntotal GPU kernel time: {tot/1e3:.1f} us over {sum(r[1] for r in rows)} dispatches")
