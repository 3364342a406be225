#!/usr/bin/env python3
"""Side measurements on one MI355X for the BASELINE.json configs that are NOT the headline (bench.py is C2):
  c3: PEMS-BAY size, 325 nodes, ChebConv Ks=3, bs 64      (fp32 path; bf16 storage is not built)
  c5: synthetic dense 8192-node graph, ChebConv Ks=5, bs 16 (tiled graph conv: one GEMM launch per operator term)
Full training step of the drop-in model (zero_grad + forward + MSE + backward + AdamW, dropout on), eager launches, the
library's own hipEvent timers for the per-kernel split.  One JSON line per config on stdout.

  python tools/gpu_side_configs.py c5 [c3] [--steps K] [--precision fp32 bf16x3 bf16]
"""
import argparse
import ctypes as C
import json
import os
import sys
import time
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CONFIGS = {"c3": dict(N=325, Ks=3, B=64), "c5": dict(N=8192, Ks=5, B=16)}
N_HIS, KT = 12, 3
BLOCKS = [[1], [64, 16, 64], [64, 16, 64], [128, 128], [1]]
PEAK_FP32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32
PEAK_BF16_MFMA_TFLOPS = 2516.6     # 16 x the fp32 MFMA rate (dense; AMD's 5 PF figure is 2:1 sparse)


def synthetic_operator(N, dev):
    """Symmetric, ~40 % dense, infinity norm 1 (spectrum inside [-1, 1] like the rescaled Laplacian); values do not matter
    for timing, only the dense N x N shape does."""
    g = torch.Generator(device="cpu").manual_seed(0)
    a = torch.rand(N, N, generator=g) * (torch.rand(N, N, generator=g) < 0.4)
    a = torch.maximum(a, a.t())
    a = a / a.sum(1).max()
    return a.to(dev)


def run(name, steps, precision="fp32", ldpad=None):
    from stgcn_amd import DropoutStream, _lib, models, ops
    from stgcn_amd.train import make_optimizer, train_step
    cfg = CONFIGS[name]
    N, Ks, B = cfg["N"], cfg["Ks"], cfg["B"]
    dev = torch.device("cuda", 0)
    L = _lib.lib()
    assert L.backend == "hip-gfx950"
    ops.set_gc_precision(precision)          # operator products of the tiled graph conv (graphs beyond 512 nodes only)
    if ldpad is not None:
        ops.set_gc_ld_pad(ldpad)             # row padding of the 16-bit planes (tuning knob)
    torch.cuda.reset_peak_memory_stats()
    gso = synthetic_operator(N, dev)
    args = types.SimpleNamespace(Kt=KT, Ks=Ks, act_func="glu", graph_conv_type="cheb_graph_conv", gso=gso, enable_bias=True, droprate=0.5,
                                 n_his=N_HIS)
    torch.manual_seed(42)
    model = models.STGCNChebGraphConv(args, BLOCKS, N).to(dev)
    DropoutStream.manual_seed(1234)
    opt = make_optimizer(model, lr=1e-3, weight_decay=1e-3, capturable=False)
    g = torch.Generator(device="cpu").manual_seed(7)
    x = torch.randn(B, 1, N_HIS, N, generator=g).to(dev)
    y = torch.randn(B, N, generator=g).to(dev)
    model.train()
    for _ in range(2):
        loss = train_step(model, opt, x, y, None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = train_step(model, opt, x, y, None)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    L.dll.stgcn_profile_enable(1)
    ksteps = min(steps, 3)
    for _ in range(ksteps):
        train_step(model, opt, x, y, None)
    torch.cuda.synchronize()
    buf = C.create_string_buffer(1 << 15)
    L.check(L.dll.stgcn_profile_collect(buf, len(buf)), "stgcn_profile_collect")
    L.dll.stgcn_profile_enable(0)
    prof = json.loads(buf.value.decode())
    out = {"config": name, "N": N, "Ks": Ks, "batch": B, "dtype": "f32", "operator_products": precision if N > 512 else "fp32", "ld_pad": ops.set_gc_ld_pad(-1), "steps": steps, "ms_per_step": round(1e3 * el / steps, 3),
           "windows_per_s": round(B * steps / el, 1), "final_loss": round(float(loss.item()), 5),
           "mem_GB": round(torch.cuda.max_memory_allocated() / 2**30, 2),
           "per_kernel_ms_per_step": {k: round(v["total_ms"] / ksteps, 4) for k, v in sorted(prof.items())}}
    # operator GEMMs of the tiled path: 2 N^2 (slabs * 16) FLOPs per launch
    gemm = {}
    for blk, T1 in ((0, N_HIS - KT + 1), (1, N_HIS - 3 * (KT - 1))):
        for lab in ("gso_gemm_fwd", "gso_gemm_bwd"):
            rec = prof.get(f"{lab}@{blk}")
            if rec:
                fl = 2.0 * N * N * B * T1 * 16                      # algorithmic FLOPs of one operator product
                issued = fl * {"fp32": 1, "bf16x3": 3, "bf16": 1}[precision]      # bf16x3 issues three bf16 MFMAs per product
                peak = PEAK_FP32_MFMA_TFLOPS if precision == "fp32" else PEAK_BF16_MFMA_TFLOPS
                us = max(1e3 * rec["total_ms"] / rec["calls"], 1e-6)
                gemm[f"{lab}@{blk}"] = {"launches_per_step": rec["calls"] // ksteps, "avg_us": round(us, 1), "gflop_per_launch": round(fl / 1e9, 2),
                                        "algorithmic_tflops": round(fl / (us * 1e-6) / 1e12, 2),
                                        "issued_mfma_tflops": round(issued / (us * 1e-6) / 1e12, 2), "mfma_peak_tflops": peak,
                                        "frac_of_mfma_peak": round(issued / (us * 1e-6) / 1e12 / peak, 4)}
    if gemm:
        out["operator_gemm"] = gemm
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("configs", nargs="+", choices=sorted(CONFIGS))
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--precision", nargs="+", default=["fp32"], choices=["fp32", "bf16x3", "bf16"])
    ap.add_argument("--ldpad", nargs="+", type=int, default=[None], help="row padding(s) of the 16-bit planes to try (bf16 elements)")
    a = ap.parse_args()
    for c in a.configs:
        for pad in a.ldpad:
            for prec in (a.precision if CONFIGS[c]["N"] > 512 else ["fp32"]):
                run(c, a.steps, prec, pad)
