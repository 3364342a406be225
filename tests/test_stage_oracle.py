"""Validates the kernel-by-kernel restatement (oracle/stblock_stages.py, explicit backward,
folded residual weights) against the module-by-module oracle (autograd), in float64.  CPU only."""
import numpy as np
import pytest
import torch

from oracle import stblock_stages as st
from oracle import stgcn_oracle as orc


def _nonsym_gso(n, seed):
    rs = np.random.RandomState(seed)
    a = rs.uniform(-1, 1, (n, n)) * (rs.uniform(size=(n, n)) < 0.6)
    return a / max(1.0, np.abs(np.linalg.eigvals(a)).max())


CASES = [
    # (c_in, channels, Kt, Ks, gct, act, N, B, T)
    (1, (64, 16, 64), 3, 3, "cheb_graph_conv", "glu", 13, 2, 8),
    (64, (64, 16, 64), 3, 3, "cheb_graph_conv", "glu", 9, 2, 6),
    (64, (64, 16, 64), 3, 2, "graph_conv", "glu", 10, 3, 5),
    (8, (6, 4, 6), 2, 5, "cheb_graph_conv", "gtu", 7, 2, 6),      # align conv inside tconv (8 > 6), Ks = 5
    (4, (8, 8, 8), 3, 1, "cheb_graph_conv", "glu", 5, 1, 7),      # identity align before gc, Ks = 1
    (4, (8, 16, 8), 3, 4, "cheb_graph_conv", "glu", 6, 2, 7),     # zero-pad align before gc (8 < 16)
]


@pytest.mark.parametrize("c_in,channels,Kt,Ks,gct,act,N,B,T", CASES)
def test_stage_pipeline_matches_autograd(c_in, channels, Kt, Ks, gct, act, N, B, T):
    rs = np.random.RandomState(7)
    cfg = orc.OracleConfig(Kt=Kt, Ks=Ks, n_his=T, act_func=act, graph_conv_type=gct, droprate=0.3,
                           blocks=[[c_in], list(channels), [128, 128], [1]])
    full = orc.random_params(cfg, N, seed=3, dtype=torch.float64)
    p = {k: v for k, v in full.items() if k.startswith("st_blocks.0.")}
    gso = _nonsym_gso(N, 5)
    x = rs.standard_normal((B, c_in, T, N))
    T2 = T - 2 * (Kt - 1)
    keep = (rs.uniform(size=(B, channels[2], T2, N)) > 0.3).astype(np.float64)
    dy = rs.standard_normal((B, channels[2], T2, N))

    # module-level oracle + autograd
    xt = torch.from_numpy(x).requires_grad_(True)
    leaves = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    y = orc.st_conv_block(xt, torch.from_numpy(gso), leaves, "st_blocks.0.", cfg, c_in, channels,
                          torch.from_numpy(keep))
    names = list(leaves)
    grads = torch.autograd.grad(y, [xt] + [leaves[n] for n in names], torch.from_numpy(dy), allow_unused=True)
    dx_ref, g_ref = grads[0], dict(zip(names, grads[1:]))

    # stage-level restatement (channels-last)
    bp = st.block_params_np(p, "st_blocks.0.", gct, np.float64)
    cl = lambda a: np.ascontiguousarray(a.transpose(0, 2, 3, 1))
    y2, sv = st.stblock_fwd(cl(x), gso, bp, Kt, c_in, channels, gct, act, cl(keep), 0.3)
    assert np.abs(y2 - cl(y.detach().numpy())).max() < 1e-11
    dx2, g2 = st.stblock_bwd(cl(dy), sv, gso, bp, Kt, c_in, channels, gct, act, 0.3)
    assert np.abs(dx2 - cl(dx_ref.numpy())).max() < 1e-10

    gcname = "cheb_graph_conv" if gct == "cheb_graph_conv" else "graph_conv"
    keymap = {"tc1_w": "tmp_conv1.causal_conv.weight", "tc1_b": "tmp_conv1.causal_conv.bias",
              "tc1_aw": "tmp_conv1.align.align_conv.weight", "tc1_ab": "tmp_conv1.align.align_conv.bias",
              "al_w": "graph_conv.align.align_conv.weight", "al_b": "graph_conv.align.align_conv.bias",
              "gc_w": f"graph_conv.{gcname}.weight", "gc_b": f"graph_conv.{gcname}.bias",
              "tc2_w": "tmp_conv2.causal_conv.weight", "tc2_b": "tmp_conv2.causal_conv.bias",
              "tc2_aw": "tmp_conv2.align.align_conv.weight", "tc2_ab": "tmp_conv2.align.align_conv.bias",
              "ln_w": "tc2_ln.weight", "ln_b": "tc2_ln.bias"}
    for short, key in keymap.items():
        ref = g_ref["st_blocks.0." + key]
        mine = g2[short]
        if ref is None:
            assert mine is None, f"{key}: reference has no grad, stage oracle produced one"
        else:
            assert mine is not None, key
            assert np.abs(mine - ref.numpy()).max() < 1e-9 * max(1.0, np.abs(ref.numpy()).max()), key
