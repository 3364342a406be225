#!/usr/bin/env python3
"""Per-phase cycle stamps of ONE kernel id over a whole-model training step (diagnostic build -DSTGCN_PHASE_TIMING):
the stamps of the LAST launch of that kernel id survive for its workgroup indices (e.g. the head's conv is the last
tconv_fwd of the forward).  STGCN_PHASE_KID=<id> STGCN_PHASE_WGS=<n> (only workgroups < n are summarised)."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = "/tmp/libstgcn_phase.so"
subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", '-DSTGCN_BACKEND_NAME="hip-gfx950"',
                "-DSTGCN_PHASE_TIMING", *os.environ.get("STGCN_EXTRA_FLAGS", "").split(), os.path.join(ROOT, "stgcn_amd/csrc/stgcn_capi.hip"), "-o", LIB], check=True)
os.environ["STGCN_AMD_LIB"] = LIB
import bench  # noqa: E402
from stgcn_amd import _lib, models  # noqa: E402
from stgcn_amd.train import make_optimizer, train_step  # noqa: E402

L = _lib.lib()
dev = torch.device("cuda:0")
gso_np, _ = bench.load_gso()
B = int(os.environ.get("STGCN_BENCH_B", "32"))
model = models.STGCNChebGraphConv(bench.make_args(torch.from_numpy(gso_np).to(dev)), bench.BLOCKS, 207).to(dev)
opt = make_optimizer(model)
x = torch.randn(B, 1, 12, 207, device=dev)
y = torch.randn(B, 207, device=dev)
model.train()
for _ in range(3):
    train_step(model, opt, x, y)
torch.cuda.synchronize()
kid = int(os.environ.get("STGCN_PHASE_KID", "1"))
nw = int(os.environ.get("STGCN_PHASE_WGS", "4096"))
L.dll.stgcn_debug_phase_select(kid)
train_step(model, opt, x, y)
torch.cuda.synchronize()
buf = (C.c_longlong * (4096 * 16))()
L.dll.stgcn_debug_phase_read(buf)
a = np.frombuffer(buf, dtype=np.int64).reshape(4096, 16)[:nw]
used = a[(a != 0).any(axis=1)]
cols = [i for i in range(16) if (used[:, i] != 0).mean() > 0.5]
line, prev = [], None
for i in cols:
    if prev is not None:
        ok = (used[:, i] != 0) & (used[:, prev] != 0)
        line.append(f"p{prev}->p{i}: {np.median(used[:, i][ok] - used[:, prev][ok]):.0f}")
    prev = i
life = used[:, cols[-1]] - used[:, cols[0]]
unit = "x10ns" if "STGCN_PHASE_WALL" in os.environ.get("STGCN_EXTRA_FLAGS", "") else "cyc"
print(f"kid {kid} B={B} wgs={len(used)} lifetime median {np.median(life):.0f} max {life.max():.0f} {unit} | " + "  ".join(line))
