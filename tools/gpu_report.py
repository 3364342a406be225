#!/usr/bin/env python3
"""One-shot GPU report (run on the MI355X box): per-stage parity errors at C2 shapes, per-kernel timings of
the fused path, and the same training step executed by stock PyTorch-ROCm ops (the "unfused GPU" baseline of
SURVEY.md section 8d) -- printed as JSON lines.  Diagnostic only; nothing here is imported by the product."""
import ctypes as C
import json
import os
import sys
import time
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
from stgcn_amd import DropoutStream, _lib, layers, models  # noqa: E402
from stgcn_amd.train import make_optimizer, train_step  # noqa: E402


class EagerSTConvBlock(layers.STConvBlock):
    """The same parameters, executed op by op through PyTorch-ROCm (what the reference does on a GPU)."""

    def forward(self, x):
        x = self.tmp_conv1(x)
        x = self.graph_conv(x)
        x = torch.relu(x)
        x = self.tmp_conv2(x)
        x = self.tc2_ln(x.permute(0, 2, 3, 1)).permute(0, 3, 1, 2)
        return self.dropout(x)


def time_steps(model, opt, x, y, steps=50, warmup=10):
    for _ in range(warmup):
        train_step(model, opt, x, y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        train_step(model, opt, x, y)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def main():
    dev = torch.device("cuda:0")
    L = _lib.lib()
    print(json.dumps({"device": torch.cuda.get_device_name(0), "backend": L.backend, "torch": torch.__version__}))

    # 1. parity report at C2 full size (no asserts)
    from tests.helpers import real_gso
    gso_np = real_gso("metr_la.cheb_sym_norm_lap")
    quick = "--quick" in sys.argv
    for blk, (c_in, T) in enumerate(() if quick else ((1, 12), (64, 8))):
        from tests.gpu_util import run_block_case
        try:
            err = run_block_case(c_in, (64, 16, 64), 3, 3, "cheb_graph_conv", "glu", 207, 32, T, True, gso=gso_np)
            print(json.dumps({"parity_c2_block": blk, "errors": {k: float(f"{v:.3e}") for k, v in err.items()}}))
        except Exception as e:  # noqa: BLE001
            print(json.dumps({"parity_c2_block": blk, "exception": repr(e)}))

    # 2. fused training step: time + per-kernel profile
    gso = torch.from_numpy(gso_np).to(dev)
    torch.manual_seed(42)
    model = models.STGCNChebGraphConv(bench.make_args(gso), bench.BLOCKS, 207).to(dev)
    opt = make_optimizer(model)
    x = torch.randn(32, 1, 12, 207, device=dev)
    y = torch.randn(32, 207, device=dev)
    model.train()
    ms = time_steps(model, opt, x, y)
    print(json.dumps({"fused_step_ms": round(ms, 4), "windows_per_s": round(32 / ms * 1e3, 1)}))
    L.dll.stgcn_profile_enable(1)
    for _ in range(20):
        train_step(model, opt, x, y)
    torch.cuda.synchronize()
    buf = C.create_string_buffer(1 << 14)
    L.dll.stgcn_profile_collect(buf, len(buf))
    L.dll.stgcn_profile_enable(0)
    prof = json.loads(buf.value.decode())
    flops = bench.stblock_flops_by_label(32, 207)
    rep = {}
    for k, v in sorted(prof.items()):
        us = v["total_ms"] / 20 * 1e3
        rep[k] = {"us_per_step": round(us, 2), "calls_per_step": v["calls"] / 20}
        if flops.get(k):
            rep[k]["tflops"] = round(flops[k] / (us * 1e-6) / 1e12, 2)
            rep[k]["mfma_frac"] = round(rep[k]["tflops"] / 157.3, 3)
    print(json.dumps({"kernel_profile": rep, "sum_us": round(sum(r["us_per_step"] for r in rep.values()), 1)}))

    if quick:
        return
    # forward-only / eval timing
    model.eval()
    with torch.no_grad():
        for _ in range(5):
            model(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            model(x)
        torch.cuda.synchronize()
    print(json.dumps({"fused_eval_fwd_ms": round((time.perf_counter() - t0) / 50 * 1e3, 4)}))

    # 3. unfused baseline: identical parameters, stock PyTorch-ROCm ops
    torch.manual_seed(42)
    eager = models.STGCNChebGraphConv(bench.make_args(gso), bench.BLOCKS, 207)
    for i, b in enumerate(eager.st_blocks):
        b.__class__ = EagerSTConvBlock
    eager = eager.to(dev)
    for b in eager.st_blocks:           # the eager sub-layers read self.gso directly: must be on the device
        b.graph_conv.cheb_graph_conv.gso = gso
    opt2 = make_optimizer(eager)
    eager.train()
    ms2 = time_steps(eager, opt2, x, y, steps=30, warmup=5)
    print(json.dumps({"unfused_pytorch_rocm_step_ms": round(ms2, 4), "windows_per_s": round(32 / ms2 * 1e3, 1)}))


if __name__ == "__main__":
    main()
