#!/usr/bin/env python3
"""HBM-side traffic per launch of every labelled kernel, from two SEPARATE rocprofv3 --pmc passes
(FETCH_SIZE and WRITE_SIZE do not fit one pass: MI355X_MICROARCH.md "rocprofv3 PMC slots").

    python tools/pmc_traffic.py <pmc_fetch_results.db> <pmc_write_results.db> <launch_log_fetch> [<launch_log_write>] > profiles/rNN_pmc_traffic.json

Units / corrections (MI355X_MICROARCH.md "HBM"): both counters are in KiB-sized units of 1024 B as rocprofv3 reports
them; on gfx950 FETCH_SIZE tallies 128-B requests at 64 B, i.e. HALF the bytes of 16-B/lane coalesced streaming reads
(all of this library's global loads are float4) -> doubled.  WRITE_SIZE is used as reported (it matched the known
store volume of the gated conv, 2 x rows x C x 4 B, within 10 % on this library's kernels).  Infinity-Cache hits are
included in both, so this is fabric (memory-side of L2) traffic: an upper bound on HBM bytes.

Join: every kernel of the library is launched through STGCN_LAUNCH, which appends "label@tag <kernel> <workgroups> <threads>" to
STGCN_LAUNCH_LOG in launch order; the i-th line therefore belongs to the i-th `stgcn::` dispatch of the profiled process (both in
program order on one stream).  Each label gets the mean over ITS OWN dispatches -- two launches of one kernel symbol with equal
grids (tmp_conv2 of block 0 / 1, head vs block LayerNorm) no longer share a number.  If the counter table has no dispatch-order
column the old (kernel symbol, grid) join is used and the ambiguous labels are dropped."""
import json
import re
import sqlite3
import sys


def norm(name):
    name = re.sub(r"\(stgcn::.*$", "", name)
    name = name.replace("void ", "").replace("stgcn::", "").replace("(", "").replace(")", "")
    return name.replace(" ", "")


def dispatch_rows(path, counter):
    """[(kernel_name, grid_size, value)] of every stgcn:: dispatch in dispatch order, or None if the order is not recorded."""
    db = sqlite3.connect(path)
    cols = [d[1] for d in db.execute("pragma table_info(counters_collection)")]
    order = next((c for c in ("dispatch_id", "id", "start", "timestamp", "correlation_id") if c in cols), None)
    if order is None or "kernel_name" not in cols:
        return None
    q = f"select kernel_name, grid_size, sum(value), {order} from counters_collection where counter_name = ? group by {order}, kernel_name, grid_size order by {order}"
    rows = [(norm(k), int(g), float(v)) for k, g, v, _ in db.execute(q, (counter,)) if "stgcn::" in k]
    return rows


def read_log(path):
    out = []
    for line in open(path):
        parts = line.rstrip("\n").split("\t")
        if len(parts) == 4:
            label, kernel, wgs, threads = parts
            out.append((label, norm(kernel), int(wgs) * int(threads), int(wgs)))
    return out


def by_order(rows, log):
    """{label: (mean value, count)}: i-th log line <-> i-th dispatch; None if the two sequences do not line up."""
    if rows is None or len(rows) != len(log):
        return None
    agg = {}
    for (label, _, grid, wgs), (_, g, v) in zip(log, rows):
        if g != grid:
            return None
        a = agg.setdefault(label, [0.0, 0, wgs])
        a[0] += v
        a[1] += 1
    return {k: (a[0] / a[1], a[1], a[2]) for k, a in agg.items()}


def by_symbol(path, counter, log):
    db = sqlite3.connect(path)
    table = {}
    q = "select kernel_name, grid_size, avg(value), count(*) from counters_collection where counter_name = ? group by kernel_name, grid_size"
    for k, g, v, n in db.execute(q, (counter,)):
        table[(norm(k), int(g))] = (float(v), int(n))
    users = {}
    for label, kname, grid, wgs in log:
        users.setdefault((kname.split("<")[0], grid), set()).add(label)
    out = {}
    for label, kname, grid, wgs in log:
        if len(users[(kname.split("<")[0], grid)]) > 1:
            continue                                   # ambiguous: several labels share this (symbol, grid)
        hit = table.get((kname, grid))
        if hit is None:
            cands = [v for k, v in table.items() if k[1] == grid and k[0].split("<")[0] == kname.split("<")[0]]
            hit = cands[0] if len(cands) == 1 else None
        if hit is not None:
            out[label] = (hit[0], hit[1], wgs)
    return out


def main(fetch_db, write_db, log_fetch, log_write=None):
    lf = read_log(log_fetch)
    lw = read_log(log_write) if log_write else lf
    f = by_order(dispatch_rows(fetch_db, "FETCH_SIZE"), lf)
    w = by_order(dispatch_rows(write_db, "WRITE_SIZE"), lw)
    how = "dispatch order (STGCN_LAUNCH_LOG line i <-> i-th stgcn:: dispatch)"
    if f is None or w is None:
        f, w = by_symbol(fetch_db, "FETCH_SIZE", lf), by_symbol(write_db, "WRITE_SIZE", lw)
        how = "(kernel symbol, grid) -- dispatch order not recorded; labels sharing a symbol and grid are omitted"
    res = {}
    for label in sorted(set(f) & set(w)):
        res[label] = {"workgroups": f[label][2], "launches_sampled": f[label][1],
                      "FETCH_SIZE_raw_KiB": round(f[label][0], 1), "WRITE_SIZE_raw_KiB": round(w[label][0], 1),
                      "read_bytes": int(2 * f[label][0] * 1024), "write_bytes": int(w[label][0] * 1024),
                      "hbm_bytes": int(2 * f[label][0] * 1024 + w[label][0] * 1024)}
    json.dump({"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes), FETCH_SIZE x2 (gfx950)", "join": how,
               "per_launch": res}, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main(*sys.argv[1:5])
