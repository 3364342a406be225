"""Binds stgcn_amd to the CPU-emulated twin of the HIP library (tests only)."""
import numpy as np
import torch

from oracle import stblock_stages as st
from oracle import stgcn_oracle as orc
from stgcn_amd import _lib
from tests.emu.build_emu import build


def bind_emulator():
    import os
    # STGCN_EMU_LIB: emulated build of an experiment variant; STGCN_EMU_ASAN=1: the AddressSanitizer build (tools/emu_asan.sh preloads its runtime)
    # STGCN_EMU_RACE=1: the LDS race-check build (tools/emu_race.sh)
    L = _lib.use_library(os.environ.get("STGCN_EMU_LIB") or build(asan=os.environ.get("STGCN_EMU_ASAN") == "1", race=os.environ.get("STGCN_EMU_RACE") == "1"))
    assert L.is_emulator
    return L


def nonsym_gso(n, seed):
    rs = np.random.RandomState(seed)
    a = rs.uniform(-1, 1, (n, n)) * (rs.uniform(size=(n, n)) < 0.6)
    return (a / max(1.0, np.abs(np.linalg.eigvals(a)).max())).astype(np.float32)


def block_case(c_in, channels, Kt, Ks, gct, act, N, B, T, seed=3):
    """Parameters (reference state_dict names, fp32 torch) + numpy views for the stage oracle."""
    cfg = orc.OracleConfig(Kt=Kt, Ks=Ks, n_his=T, act_func=act, graph_conv_type=gct, droprate=0.5,
                           blocks=[[c_in], list(channels), [128, 128], [1]])
    full = orc.random_params(cfg, N, seed=seed, dtype=torch.float32)
    p = {k: v for k, v in full.items() if k.startswith("st_blocks.0.")}
    return cfg, p


def params_in_field_order(p, prefix, gct):
    gc = "graph_conv.cheb_graph_conv." if gct == "cheb_graph_conv" else "graph_conv.graph_conv."
    names = ["tmp_conv1.causal_conv.weight", "tmp_conv1.causal_conv.bias", "tmp_conv1.align.align_conv.weight",
             "tmp_conv1.align.align_conv.bias", "graph_conv.align.align_conv.weight", "graph_conv.align.align_conv.bias",
             gc + "weight", gc + "bias", "tmp_conv2.causal_conv.weight", "tmp_conv2.causal_conv.bias",
             "tmp_conv2.align.align_conv.weight", "tmp_conv2.align.align_conv.bias", "tc2_ln.weight", "tc2_ln.bias"]
    return [p.get(prefix + n) for n in names]


def big_gso(n, seed, density=0.4):
    """Dense non-symmetric operator for large graphs without an eigen-decomposition: rows scaled so that the infinity norm
    (hence the spectral radius) is <= 1, like the rescaled Laplacian the reference feeds the Chebyshev recursion."""
    rs = np.random.RandomState(seed)
    a = rs.uniform(-1, 1, (n, n)).astype(np.float32)
    a *= rs.uniform(size=(n, n)) < density
    return (a / np.abs(a).sum(1).max()).astype(np.float32)
