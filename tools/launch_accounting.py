#!/usr/bin/env python3
"""Per-launch accounting of the ST-block kernels of one bench line: time (the hipEvent pass of bench.py), algorithmic FLOPs and bytes (the
closed forms of bench.py: SURVEY.md section 8d per kernel), the fabric-side bytes the PMC passes measured, and both roofline fractions.

    python tools/launch_accounting.py profiles/rNN_bench.json [profiles/rNN_pmc_traffic.json] > profiles/rNN_launch_accounting.md
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    rec = json.load(open(sys.argv[1]))
    traffic = json.load(open(sys.argv[2]))["per_launch"] if len(sys.argv) > 2 else {}
    wl = rec["config"]["workload"]
    B = rec["config"]["global_batch"] // max(1, rec.get("n_gpus", 1))
    N = 207 if "207 nodes" in wl else 325 if "325 nodes" in wl else 8192
    Ks = 5 if N > 1024 else 3
    bf = rec["dtype"] == "bf16"
    e = 2 if bf else 4
    flops = bench.stblock_flops_by_label(B, N, Ks)
    nbytes = bench.stblock_bytes_by_label(B, N, Ks, e)
    peak_tf = bench.PEAK_BF16_MFMA_TFLOPS if bf else bench.PEAK_FP32_MFMA_TFLOPS
    peak_gbs = bench.PEAK_HBM_GBS
    per = rec["roofline"]["per_kernel_us_per_step"]
    print(f"# per-launch accounting: {wl}, {rec['dtype']}, step {rec['ms_per_step']} ms (hipGraph replay); launch times: bench.py's hipEvent pass (eager)\n")
    print(f"peaks: {peak_tf:g} TFLOP/s ({'bf16' if bf else 'fp32'} MFMA, dense), {peak_gbs:g} GB/s HBM; `measured MB` = 2 x FETCH_SIZE + WRITE_SIZE of the launch "
          "(fabric side of L2, Infinity-Cache hits included)\n")
    print("| launch | us | GFLOP | TFLOP/s | % of MFMA peak | algorithmic MB | GB/s (algorithmic) | % of HBM peak | measured MB |")
    print("|---|---|---|---|---|---|---|---|---|")
    tot_us = tot_f = 0.0
    for k in sorted(per, key=lambda k: -per[k]):
        us = per[k]
        f = flops.get(k)
        b = nbytes.get(k)
        t = traffic.get(k)
        tb = None
        if isinstance(t, dict):
            tb = t.get("hbm_bytes")
        fs = f"{f / 1e9:.3f} | {f / us / 1e6:.1f} | {100 * f / us / 1e6 / peak_tf:.1f}" if f else "- | - | -"
        bs = f"{b / 1e6:.1f} | {b / us / 1e3:.0f} | {100 * b / us / 1e3 / peak_gbs:.1f}" if b else "- | - | -"
        print(f"| {k} | {us:.1f} | {fs} | {bs} | {tb / 1e6:.1f} |" if tb else f"| {k} | {us:.1f} | {fs} | {bs} | - |")
        if f and not k.startswith(("head.", "adamw", "prepack", "reduce")):
            tot_us += us
            tot_f += f
    if tot_us:
        print(f"\nST-block launches with matrix work: {tot_f / 1e9:.2f} GFLOP in {tot_us:.1f} us = {tot_f / tot_us / 1e6:.1f} TFLOP/s = "
              f"{100 * tot_f / tot_us / 1e6 / peak_tf:.1f} % of the MFMA peak")


if __name__ == "__main__":
    main()
