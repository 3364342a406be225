"""Backward kernels of fp32 blocks with "bf16x3" matrix products (ops.set_bwd_precision: split operands, three bf16 MFMAs per product,
~2^-16 relative) on the CPU emulator: forward results are unchanged (exact fp32 products), every gradient stays inside the 1e-3 bar of
the fp32 configurations against the float64 stage oracle."""
import numpy as np
import pytest
import torch

from oracle import stblock_stages as st
from stgcn_amd import ops
from tests.emu_util import bind_emulator, block_case, nonsym_gso, params_in_field_order

CASES = [
    (1, (64, 16, 64), 3, 3, "cheb_graph_conv", "glu", 21, 2, 7, True),
    (64, (64, 16, 64), 3, 3, "cheb_graph_conv", "glu", 17, 2, 6, True),
    (64, (64, 16, 64), 3, 3, "graph_conv", "gtu", 35, 1, 5, False),
]


def run(c_in, channels, Kt, Ks, gct, act, N, B, T, training, dev="cpu"):
    cfg, p = block_case(c_in, channels, Kt, Ks, gct, act, N, B, T)
    gso = nonsym_gso(N, 5)
    rs = np.random.RandomState(11)
    x_np = rs.standard_normal((B, c_in, T, N)).astype(np.float32)
    T2 = T - 2 * (Kt - 1)
    dy_np = rs.standard_normal((B, channels[2], T2, N)).astype(np.float32)
    bcfg = ops.BlockConfig(Kt=Kt, Ks=Ks, n_vertex=N, c_in=c_in, channels=tuple(channels), act_func=act, graph_conv_type=gct, droprate=0.5)
    gp, gt = ops.gso_prepare(torch.from_numpy(gso).to(dev), ops.graph_terms(bcfg))
    params = [None if t is None else t.clone().to(dev).requires_grad_(True) for t in params_in_field_order(p, "st_blocks.0.", gct)]
    x = torch.from_numpy(x_np).to(dev).requires_grad_(c_in > 1)
    y = ops.st_conv_block(x, gp, gt, bcfg, params, training, 99, 3, ops.WorkspaceCache())
    y.backward(torch.from_numpy(dy_np).to(dev))
    cl = lambda a: np.ascontiguousarray(a.transpose(0, 2, 3, 1)).astype(np.float64)
    keep = None
    if training:
        ks = ops.dropout_mask(B * T2 * N * channels[2], 0.5, 99, 3, dev).cpu().numpy().reshape(B, T2, N, channels[2])
        keep = (ks > 0).astype(np.float64)
    bp = st.block_params_np(p, "st_blocks.0.", gct, np.float64)
    y_ref, sv = st.stblock_fwd(cl(x_np), gso.astype(np.float64), bp, Kt, c_in, channels, gct, act, keep, 0.5)
    dx_ref, g_ref = st.stblock_bwd(cl(dy_np), sv, gso.astype(np.float64), bp, Kt, c_in, channels, gct, act, 0.5, need_dx=c_in > 1)
    rel = lambda got, ref: float(np.abs(got - ref).max() / max(1e-30, np.abs(ref).max()))
    err = {"fwd.y": float(np.abs(cl(y.detach().cpu().numpy()) - y_ref).max())}
    if c_in > 1:
        err["bwd.dx"] = rel(cl(x.grad.cpu().numpy()), dx_ref)
    from stgcn_amd import _lib
    for name, prm in zip(_lib.PARAM_FIELDS, params):
        if prm is not None and g_ref[name] is not None:
            err["grad." + name] = rel(prm.grad.cpu().numpy().astype(np.float64), g_ref[name].reshape(prm.shape))
    return err


@pytest.mark.parametrize("c_in,channels,Kt,Ks,gct,act,N,B,T,training", CASES)
def test_bf16x3_backward_stays_inside_the_gradient_bar(c_in, channels, Kt, Ks, gct, act, N, B, T, training):
    bind_emulator()
    prev = ops.set_bwd_precision("bf16x3")
    try:
        err = run(c_in, channels, Kt, Ks, gct, act, N, B, T, training)
    finally:
        assert ops.set_bwd_precision(prev) == "bf16x3"
    assert err.pop("fwd.y") <= 5e-5           # the forward keeps exact fp32 products
    worst = max(err.values())
    assert worst <= 1e-3, err                 # gradient bar of the fp32 configurations (measured: ~3e-5)
    assert worst >= 1e-7, err                 # ... and it is not the exact path that ran
