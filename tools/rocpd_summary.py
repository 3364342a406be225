#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per kernel name AND launch geometry -> calls, avg/min/max us.
    python tools/rocpd_summary.py gpurun_out/prof/r01_results.db > profiles/r01_kernel_stats.md"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*$", "", name)
    name = name.replace("void ", "").replace("stgcn::", "")
    return name[:70]


def main(path, skip_first=0):
    db = sqlite3.connect(path)
    rows = db.execute("select name, grid_x, grid_y, workgroup_x, lds_size, vgpr_count, accum_vgpr_count, duration from kernels").fetchall()
    agg = {}
    for name, gx, gy, wx, lds, vg, ag, dur in rows:
        key = (short(name), gx // max(wx, 1), gy, lds, vg, ag)
        a = agg.setdefault(key, [0, 0.0, 1e30, 0.0])
        a[0] += 1
        a[1] += dur
        a[2] = min(a[2], dur)
        a[3] = max(a[3], dur)
    total = sum(a[1] for a in agg.values())
    print(f"# rocprofv3 --kernel-trace summary of {path}\n")
    print(f"total kernel time {total / 1e6:.3f} ms over {sum(a[0] for a in agg.values())} dispatches\n")
    print("| kernel | workgroups | grid_y | LDS B | VGPR | AGPR | calls | avg us | min us | max us | % time |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    for key, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        n, gx, gy, lds, vg, ag = key
        print(f"| {n} | {gx} | {gy} | {lds} | {vg} | {ag} | {a[0]} | {a[1] / a[0] / 1e3:.2f} | {a[2] / 1e3:.2f} | {a[3] / 1e3:.2f} | {100 * a[1] / total:.1f} |")


if __name__ == "__main__":
    main(sys.argv[1])
