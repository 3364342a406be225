#!/usr/bin/env python3
"""HBM-side traffic per launch of every labelled kernel, from two SEPARATE rocprofv3 --pmc passes
(FETCH_SIZE and WRITE_SIZE do not fit one pass: MI355X_MICROARCH.md "rocprofv3 PMC slots").

    python tools/pmc_traffic.py <pmc_fetch_results.db> <pmc_write_results.db> <launch_log> > profiles/rNN_pmc_traffic.json

Units / corrections (MI355X_MICROARCH.md "HBM"): both counters are in KiB-sized units of 1024 B as rocprofv3 reports
them; on gfx950 FETCH_SIZE tallies 128-B requests at 64 B, i.e. HALF the bytes of 16-B/lane coalesced streaming reads
(all of this library's global loads are float4) -> doubled.  WRITE_SIZE is used as reported (it matched the known
store volume of the gated conv, 2 x rows x C x 4 B, within 10 % on this library's kernels).  Infinity-Cache hits are
included in both, so this is fabric (memory-side of L2) traffic: an upper bound on HBM bytes.

The launch log (STGCN_LAUNCH_LOG=<file>) joins the library's labels with rocprof's (kernel symbol, grid size)."""
import json
import re
import sqlite3
import sys


def norm(name):
    name = re.sub(r"\(stgcn::.*$", "", name)
    name = name.replace("void ", "").replace("stgcn::", "").replace("(", "").replace(")", "")
    return name.replace(" ", "")


def counter_means(path, counter):
    db = sqlite3.connect(path)
    out = {}
    q = "select kernel_name, grid_size, avg(value), count(*) from counters_collection where counter_name = ? group by kernel_name, grid_size"
    for k, g, v, n in db.execute(q, (counter,)):
        out[(norm(k), int(g))] = (float(v), int(n))
    return out


def main(fetch_db, write_db, launch_log):
    fetch = counter_means(fetch_db, "FETCH_SIZE")
    write = counter_means(write_db, "WRITE_SIZE")
    labels = {}
    for line in open(launch_log):
        parts = line.rstrip("\n").split("\t")
        if len(parts) != 4:
            continue
        label, kernel, wgs, threads = parts
        labels.setdefault(label, (norm(kernel), int(wgs) * int(threads), int(wgs)))
    def lookup(table, kname, grid):
        """exact (symbol, grid) match, else the unique kernel with the same base name and grid (the launch log shows
        template arguments as written in the source, e.g. <MTW, 2, true>, rocprof shows them instantiated)"""
        hit = table.get((kname, grid))
        if hit is not None:
            return hit, kname
        base = kname.split("<")[0]
        cands = [(k, v) for k, v in table.items() if k[1] == grid and k[0].split("<")[0] == base]
        if len(cands) == 1:
            return cands[0][1], cands[0][0][0]
        return None, kname

    res = {}
    for label, (kname, grid, wgs) in sorted(labels.items()):
        f, kname_f = lookup(fetch, kname, grid)
        w, _ = lookup(write, kname, grid)
        if f is None or w is None:
            continue
        kname = kname_f
        res[label] = {
            "kernel": kname, "workgroups": wgs, "launches_sampled": f[1],
            "FETCH_SIZE_raw_KiB": round(f[0], 1), "WRITE_SIZE_raw_KiB": round(w[0], 1),
            "read_bytes": int(2 * f[0] * 1024), "write_bytes": int(w[0] * 1024),
            "hbm_bytes": int(2 * f[0] * 1024 + w[0] * 1024),
        }
    json.dump({"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes), FETCH_SIZE x2 (gfx950)",
               "per_launch": res}, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main(*sys.argv[1:4])
