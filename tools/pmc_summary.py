#!/usr/bin/env python3
"""Per-kernel mean of every PMC counter in a rocprofv3 --pmc run (rocpd SQLite output).
usage: python tools/pmc_summary.py <results.db> [more.db ...]"""
import re, sqlite3, sys
def short(n):
    n = re.sub(r"\(.*", "", n).replace("void dae::", "").replace("dae::", "")
    return n[:58]
print("| kernel | counter | dispatches | mean value | mean duration us |"); print("|---|---|---:|---:|---:|")
for path in sys.argv[1:]:
    db = sqlite3.connect(path)
    rows = db.execute("select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection "
                      "group by kernel_name, counter_name order by kernel_name, counter_name").fetchall()
    for n, c, cnt, v, d in rows:
        if "at::native" in n or "rocclr" in n: continue
        print(f"| `{short(n)}` | {c} | {cnt} | {v:.5g} | {d/1e3:.1f} |")
