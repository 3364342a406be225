#!/usr/bin/env python3
"""Per-phase cycle stamps of the main kernels (diagnostic build -DSTGCN_PHASE_TIMING) at C2 shapes.
Prints, per kernel: number of workgroups that stamped, the mean duration (cycles) between consecutive phase
stamps, and the span first-start -> last-end (= kernel duration in shader cycles)."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.environ.get("STGCN_PHASE_LIB", "/tmp/libstgcn_phase.so")   # (a prebuilt library travels with the snapshot: no compile on the GPU box)
if not os.path.exists(LIB):
  subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", '-DSTGCN_BACKEND_NAME="hip-gfx950"',
                "-DSTGCN_PHASE_TIMING", *os.environ.get("STGCN_EXTRA_FLAGS", "").split(), os.path.join(ROOT, "stgcn_amd/csrc/stgcn_capi.hip"), "-o", LIB], check=True)
os.environ["STGCN_AMD_LIB"] = LIB
from stgcn_amd import _lib, ops  # noqa: E402
from tests.emu_util import block_case, params_in_field_order  # noqa: E402
from tests.helpers import real_gso  # noqa: E402

L = _lib.lib()
if os.environ.get("STGCN_BENCH_TC2LN_PP"):
    ops.set_tc2ln_peers(int(os.environ["STGCN_BENCH_TC2LN_PP"]))
dev = "cuda:0"
NAMES = {1: "tconv_fwd", 2: "tconv_bwd_data", 3: "tconv_bwd_weight", 4: "gconv_fwd", 6: "align_gate_bwd", 7: "tconv_fwd(tc2, v3)",
         8: "tc2_bwd", 9: "tc2_ln_fwd", 10: "tc1_bwd", 11: "tc1_fwd"}


def run_block(c_in, T):
    cfg, p = block_case(c_in, (64, 16, 64), 3, 3, "cheb_graph_conv", "glu", 207, 32, T)
    gso = real_gso("metr_la.cheb_sym_norm_lap")
    bcfg = ops.BlockConfig(Kt=3, Ks=3, n_vertex=207, c_in=c_in, channels=(64, 16, 64), act_func="glu", graph_conv_type="cheb_graph_conv",
                           droprate=0.5)
    gp, gt = ops.gso_prepare(torch.from_numpy(gso).to(dev), ops.graph_terms(bcfg))
    params = [None if t is None else t.clone().to(dev).requires_grad_(True) for t in params_in_field_order(p, "st_blocks.0.", "cheb_graph_conv")]
    x = torch.randn(int(os.environ.get("STGCN_PHASE_B", "32")), c_in, T, 207, device=dev).requires_grad_(c_in > 1)
    wsc = ops.WorkspaceCache()

    def go():
        y = ops.st_conv_block(x, gp, gt, bcfg, params, True, 1, 1, wsc)
        y.backward(torch.randn_like(y))
        torch.cuda.synchronize()
    return go


def report(kid, which, go):
    for _ in range(2):
        go()
    L.dll.stgcn_debug_phase_select(kid)
    go()
    buf = (C.c_longlong * (4096 * 16))()
    L.dll.stgcn_debug_phase_read(buf)
    a = np.frombuffer(buf, dtype=np.int64).reshape(4096, 16)
    used = a[(a != 0).any(axis=1)]
    if len(used) == 0:
        print(NAMES[kid], which, "no stamps")
        return
    stamps = used[:, :8]                                  # (columns 8.. hold accumulators, not clock values)
    t0 = stamps[stamps > 0].min()
    span = stamps.max() - t0
    cols = [i for i in range(8) if (used[:, i] != 0).mean() > 0.5]
    busy = [f"acc{i}: {used[:, i].mean():.0f}" for i in range(8, 16) if (used[:, i] != 0).mean() > 0.5]   # role busy-time accumulators
    if busy:
        print(f"{NAMES[kid]:18s} {which:6s} busy cycles per role (mean over workgroups): " + "  ".join(busy))
    line = []
    prev = None
    for i in cols:
        v = used[:, i]
        if prev is not None:
            ok = (v != 0) & (used[:, prev] != 0)
            line.append(f"p{prev}->p{i}: {np.mean(v[ok] - used[:, prev][ok]):.0f}")
        prev = i
    start_spread = (used[:, cols[0]] - t0)
    if "STGCN_PHASE_WALL" in os.environ.get("STGCN_EXTRA_FLAGS", ""):
        st = (used[:, cols[0]] - t0) / 100.0
        en = (used[:, cols[-1]] - t0) / 100.0
        q = lambda v: " ".join(f"{np.percentile(v, p):.1f}" for p in (0, 10, 25, 50, 75, 90, 100))
        print(f"{NAMES[kid]:18s} {which:6s} wgs={len(used):5d} WALL us: start pct[0,10,25,50,75,90,100] = {q(st)} | end = {q(en)} | lifetime = {q(en - st)}")
        return
    # per-XCD timelines (every XCD has its own s_memtime base): when do workgroups start / end relative to the XCD's first start
    idx = np.nonzero((a != 0).any(axis=1))[0]
    st_rel, en_rel = [], []
    for x in range(8):
        rows = a[idx[idx % 8 == x]]
        if len(rows) == 0:
            continue
        first = rows[:, cols[0]]
        last = rows[:, cols[-1]]
        ok = (first != 0) & (last != 0)
        if ok.sum() == 0:
            continue
        base = first[ok].min()
        st_rel.append(first[ok] - base)
        en_rel.append(last[ok] - base)
    if st_rel:
        st_rel = np.concatenate(st_rel); en_rel = np.concatenate(en_rel)
        q = lambda v: " ".join(f"{np.percentile(v, p):.0f}" for p in (0, 25, 50, 75, 100))
        print(f"   per-XCD timeline (cycles): start pct[0,25,50,75,100] = {q(st_rel)} | end = {q(en_rel)} | lifetime = {q(en_rel - st_rel)}")
    print(f"{NAMES[kid]:18s} {which:10s} wgs={len(used):5d} span={span} cyc  start mean={start_spread.mean():.0f} max={start_spread.max()}  | " + "  ".join(line))
    print(f"   (the LAST of several launches of this kernel in the step overwrote earlier ones)")


go1 = run_block(64, 8)
go0 = run_block(1, 12)
KIDS = [int(k) for k in os.environ.get("STGCN_PHASE_KIDS", "1,2,3,4,6").split(",")]
for kid in KIDS:
    report(kid, "blk1", go1)
for kid in KIDS:
    if kid in (1, 3, 4, 6, 7, 8, 9, 11):
        report(kid, "blk0", go0)
