#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc output (rocpd sqlite): per kernel name (+ workgroup count) the mean of each counter.
    python tools/rocpd_pmc_summary.py <db> """
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*$", "", name).replace("void ", "").replace("stgcn::", "")
    return name[:60]


def main(path):
    db = sqlite3.connect(path)
    names = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
    cand = [n for n in names if "pmc" in n.lower() or "counter" in n.lower()]
    print(f"# PMC summary of {path}\n\npmc-related tables/views: {cand}\n")
    view = None
    for v in ("counters_collection", "pmc_events"):
        if v in names:
            view = v
            break
    if view is None:
        for n in cand:
            print(n, [d[1] for d in db.execute(f"pragma table_info({n})")])
        return
    cols = [d[1] for d in db.execute(f"pragma table_info({view})")]
    print(f"using view {view}: columns {cols}\n")
    kcol = "kernel_name" if "kernel_name" in cols else ("name" if "name" in cols else None)
    ccol = "counter_name" if "counter_name" in cols else ("pmc_name" if "pmc_name" in cols else ("symbol" if "symbol" in cols else None))
    vcol = "value" if "value" in cols else ("counter_value" if "counter_value" in cols else None)
    gcol = "grid_size" if "grid_size" in cols else ("grid_x" if "grid_x" in cols else None)
    if not (kcol and ccol and vcol):
        print("unrecognised schema; first rows:")
        for r in db.execute(f"select * from {view} limit 5"):
            print(r)
        return
    q = f"select {kcol}, {gcol or '0'}, {ccol}, avg({vcol}), count(*) from {view} group by {kcol}, {gcol or '0'}, {ccol}"
    agg = {}
    for k, gsz, c, v, n in db.execute(q):
        agg.setdefault((short(k), gsz), {})[c] = (v, n)
    counters = sorted({c for d in agg.values() for c in d})
    print("| kernel | grid | n | " + " | ".join(counters) + " |")
    print("|---|---|---|" + "---|" * len(counters))
    for (k, gsz), d in sorted(agg.items()):
        if k.startswith(("at::", "void at::")) or "elementwise_kernel" in k:      # ATen fills / copies around the library's kernels
            continue
        n = max(x[1] for x in d.values())
        print(f"| {k} | {gsz} | {n} | " + " | ".join(f"{d[c][0]:.4g}" if c in d else "" for c in counters) + " |")


if __name__ == "__main__":
    main(sys.argv[1])
