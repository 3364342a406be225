#!/usr/bin/env python3
"""Per-kernel times (library hipEvent timer, eager launches) of ONE ST block fwd+bwd at the C2 shapes, for A/B runs of kernel variants:
    python tools/gpu_block_times.py [--block 0|1] [--eval] [--B 32] [--iters 30]
Prints one JSON line {label: us_per_call}."""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stgcn_amd import _lib, ops  # noqa: E402
from tests.emu_util import block_case, params_in_field_order  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--block", type=int, default=1)
    ap.add_argument("--eval", action="store_true")
    ap.add_argument("--B", type=int, default=32)
    ap.add_argument("--N", type=int, default=207)
    ap.add_argument("--iters", type=int, default=30)
    a = ap.parse_args()
    dev = "cuda:0"
    c_in, T = (1, 12) if a.block == 0 else (64, 8)
    channels, Kt, Ks, gct = (64, 16, 64), 3, 3, "cheb_graph_conv"
    cfg, p = block_case(c_in, channels, Kt, Ks, gct, "glu", a.N, a.B, T)
    gso = np.load(os.path.join(ROOT, "tests", "golden", "gso_real.npz"))["metr_la.cheb_sym_norm_lap"] if a.N == 207 else None
    if gso is None:
        rs = np.random.RandomState(0)
        gso = (rs.uniform(-1, 1, (a.N, a.N)) / a.N).astype(np.float32)
    bcfg = ops.BlockConfig(Kt=Kt, Ks=Ks, n_vertex=a.N, c_in=c_in, channels=channels, act_func="glu", graph_conv_type=gct, droprate=0.5, tag=a.block)
    gp, gt = ops.gso_prepare(torch.from_numpy(gso).to(dev), ops.graph_terms(bcfg))
    params = [None if t is None else t.clone().to(dev).requires_grad_(True) for t in params_in_field_order(p, "st_blocks.0.", gct)]
    g = torch.Generator().manual_seed(0)
    x = torch.randn(a.B, c_in, T, a.N, generator=g).to(dev).requires_grad_(c_in > 1)
    dy = torch.randn(a.B, channels[2], T - 4, a.N, generator=g).to(dev)
    wsc = ops.WorkspaceCache()
    L = _lib.lib()

    def step(i):
        y = ops.st_conv_block(x, gp, gt, bcfg, params, not a.eval, 7, i + 1, wsc)
        y.backward(dy)

    for i in range(5):
        step(i)
    torch.cuda.synchronize()
    L.dll.stgcn_profile_enable(1)
    for i in range(a.iters):
        step(i)
    torch.cuda.synchronize()
    buf = C.create_string_buffer(1 << 14)
    L.check(L.dll.stgcn_profile_collect(buf, len(buf)), "collect")
    L.dll.stgcn_profile_enable(0)
    prof = json.loads(buf.value.decode())
    out = {k: round(1e3 * v["total_ms"] / v["calls"], 2) for k, v in sorted(prof.items())}
    out["_sum"] = round(sum(1e3 * v["total_ms"] for v in prof.values()) / a.iters, 2)
    out["_cfg"] = {"block": a.block, "eval": a.eval, "B": a.B, "fuse": os.environ.get("STGCN_FUSE", "all")}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
