"""Backward kernels on the CPU emulator, through the real autograd.Function, checked against the numpy
stage oracle (explicit backward) -- intermediates in `ws` first, then dx and every parameter gradient."""
import numpy as np
import pytest
import torch

from oracle import stblock_stages as st
from stgcn_amd import _lib, ops
from tests.emu_util import bind_emulator, block_case, nonsym_gso, params_in_field_order

CASES = [
    # c_in, channels, Kt, Ks, gct, act, N, B, T, training
    (1, (64, 16, 64), 3, 3, "cheb_graph_conv", "glu", 21, 2, 7, True),
    (64, (64, 16, 64), 3, 3, "cheb_graph_conv", "glu", 17, 2, 6, True),
    (64, (64, 16, 64), 3, 3, "graph_conv", "gtu", 35, 1, 5, False),
    (16, (128, 16, 64), 2, 5, "cheb_graph_conv", "glu", 9, 2, 5, True),
    (32, (64, 16, 128), 3, 1, "cheb_graph_conv", "glu", 16, 1, 5, False),
    (128, (64, 16, 64), 3, 2, "cheb_graph_conv", "glu", 10, 1, 5, True),
    (4, (64, 16, 64), 3, 3, "cheb_graph_conv", "gtu", 10, 1, 6, False),     # thin first layer (K = 12) WITH an input gradient
    (2, (64, 16, 64), 2, 3, "cheb_graph_conv", "glu", 19, 2, 5, True),      # K = Kt * c_in = 4: the wave-per-tile thin kernels with all four taps, dZ1 written for the input gradient
]


@pytest.mark.parametrize("c_in,channels,Kt,Ks,gct,act,N,B,T,training", CASES)
def test_block_backward(c_in, channels, Kt, Ks, gct, act, N, B, T, training):
    bind_emulator()
    ops.set_debug_stages(True)      # fused kernels also write the intermediates they keep on chip (dZ2)
    cfg, p = block_case(c_in, channels, Kt, Ks, gct, act, N, B, T)
    gso = nonsym_gso(N, 5)
    rs = np.random.RandomState(11)
    x_np = rs.standard_normal((B, c_in, T, N)).astype(np.float32)       # logical NCHW
    T2 = T - 2 * (Kt - 1)
    dy_np = rs.standard_normal((B, channels[2], T2, N)).astype(np.float32)
    pdrop = 0.5

    bcfg = ops.BlockConfig(Kt=Kt, Ks=Ks, n_vertex=N, c_in=c_in, channels=tuple(channels), act_func=act,
                           graph_conv_type=gct, droprate=pdrop)
    gp, gt = ops.gso_prepare(torch.from_numpy(gso), ops.graph_terms(bcfg))
    params = [None if t is None else t.clone().requires_grad_(True) for t in params_in_field_order(p, "st_blocks.0.", gct)]
    x = torch.from_numpy(x_np).requires_grad_(c_in > 1)
    wsc = ops.WorkspaceCache()
    seed, offset = 99, 3
    y = ops.st_conv_block(x, gp, gt, bcfg, params, training, seed, offset, wsc)
    assert y.shape == (B, channels[2], T2, N)
    y.backward(torch.from_numpy(dy_np))

    # ---- oracle ------------------------------------------------------------------------------------
    cl = lambda a: np.ascontiguousarray(a.transpose(0, 2, 3, 1)).astype(np.float64)
    keep = None
    if training:
        ks = ops.dropout_mask(B * T2 * N * channels[2], pdrop, seed, offset, "cpu").numpy().reshape(B, T2, N, channels[2])
        keep = (ks > 0).astype(np.float64)
    bp = st.block_params_np(p, "st_blocks.0.", gct, np.float64)
    y_ref, sv = st.stblock_fwd(cl(x_np), gso.astype(np.float64), bp, Kt, c_in, channels, gct, act, keep, pdrop)
    dx_ref, g_ref = st.stblock_bwd(cl(dy_np), sv, gso.astype(np.float64), bp, Kt, c_in, channels, gct, act, pdrop,
                                   need_dx=c_in > 1)
    assert np.abs(cl(y.detach().numpy()) - y_ref).max() < 5e-5

    # ---- intermediates kept in the workspace ----------------------------------------------------------
    desc = ops.make_desc(bcfg, B, T, training, c_in > 1)
    plan = ops.query_plan(desc)
    ws = wsc.buf.numpy()
    T1 = plan.T1
    c0, c1, c2 = channels
    dH2, _, _ = st.ln_dropout_bwd(cl(dy_np), sv["H2"], bp["ln_w"], sv["mean"], sv["rstd"], keep, pdrop)
    dZ2 = st.gate_bwd(dH2, sv["U2"], sv["S2"], act)
    got = ws[plan.ws_dZ2:plan.ws_dZ2 + dZ2.size].reshape(dZ2.shape)
    assert np.abs(got - dZ2).max() < 1e-4 * max(1.0, np.abs(dZ2).max()), "dZ2"
    dG = st.tconv_bwd_data(dZ2, sv["W2"], Kt, c1)
    dYg = dG * (sv["G"] > 0)
    got = ws[plan.ws_dYg:plan.ws_dYg + dYg.size].reshape(dYg.shape)
    assert np.abs(got - dYg).max() < 1e-4 * max(1.0, np.abs(dYg).max()), "dYg"
    dA, _, _ = st.gconv_bwd(dG, sv["G"], sv["Xs"], gso.astype(np.float64), sv["Wk"])
    got = ws[plan.ws_dA:plan.ws_dA + dA.size].reshape(dA.shape)
    assert np.abs(got - dA).max() < 1e-4 * max(1.0, np.abs(dA).max()), "dA"
    dZ1 = st.gate_bwd(dA @ sv["Wa"].T, sv["U1"], sv["S1"], act)
    if not (plan.thin_tc1 and c_in == 1):      # the thin first layer keeps dZ1 on chip unless dx is needed
        got = ws[plan.ws_dZ1:plan.ws_dZ1 + dZ1.size].reshape(dZ1.shape)
        assert np.abs(got - dZ1).max() < 1e-4 * max(1.0, np.abs(dZ1).max()), "dZ1"

    # ---- outputs -----------------------------------------------------------------------------------------
    if c_in > 1:
        assert np.abs(cl(x.grad.numpy()) - dx_ref).max() < 1e-4 * max(1.0, np.abs(dx_ref).max()), "dx"
    else:
        assert x.grad is None
    for name, prm in zip(_lib.PARAM_FIELDS, params):
        ref = g_ref[name]
        if prm is None:
            continue
        if ref is None:
            assert prm.grad is None, f"{name}: reference leaves .grad None"
            continue
        assert prm.grad is not None, name
        err = np.abs(prm.grad.numpy().astype(np.float64) - ref.reshape(prm.shape)).max()
        assert err < 2e-4 * max(1.0, np.abs(ref).max()), f"{name}: {err}"


@pytest.mark.parametrize("wgs", [1, 3, 5])
def test_tc1_bwd_ranges_cut_inside_items(wgs):
    """tc1_bwd_kernel hands every workgroup an equal-weight range of the (window, node tile, output step) sequence; with fewer workgroups
    than items the cuts fall inside items (halo tiles re-formed, weight gradients owned by the range of the tile's output step): same
    results as one workgroup per item."""
    bind_emulator()
    prev = ops.set_tc1_bwd_wgs(wgs)
    try:
        test_block_backward(64, (64, 16, 64), 3, 3, "cheb_graph_conv", "glu", 17, 2, 6, True)
        test_block_backward(32, (64, 16, 128), 3, 1, "cheb_graph_conv", "glu", 16, 1, 5, False)
    finally:
        ops.set_tc1_bwd_wgs(prev)


def test_gconv_bwd2_job_waves_fit_the_launch_bounds(monkeypatch):
    """with Ks >= 4 the split graph-conv backward asked for one job wave per parameter-gradient job and, with 8 tile waves,
    for more than the 12 waves its __launch_bounds__(768) admits.  The job-wave count is clamped (the job loop strides by it); forcing
    two parts on a 256-node-tile graph with Ks = 8 gives 8 tile waves + 5 wanted job waves -> 4."""
    import tests.test_emu_backward as me

    def sym_gso(n, seed):      # symmetric, spectrum in [-1, 1]: T_7 of it stays bounded (a non-normal operator's polynomials grow)
        rs = np.random.RandomState(seed)
        a = rs.uniform(-1, 1, (n, n)) * (rs.uniform(size=(n, n)) < 0.6)
        a = 0.5 * (a + a.T)
        return (a / np.abs(np.linalg.eigvalsh(a)).max()).astype(np.float32)

    monkeypatch.setattr(me, "nonsym_gso", sym_gso)
    monkeypatch.setenv("STGCN_GCBWD2_PARTS", "2")
    test_block_backward(64, (64, 16, 64), 3, 8, "cheb_graph_conv", "glu", 250, 1, 5, True)


@pytest.mark.parametrize("mask", ["y", "philox"])
def test_zero_initialised_layernorm_affine_still_gets_gradients(mask, monkeypatch):
    """with gamma = beta = 0 every element of the block output is an exact zero, kept or dropped.  The backward reads the dropout
    mask off the output; the forward therefore stores dropped elements as -0.0 and kept zeros as +0.0 (drop_encode) -- d gamma / d beta must
    come out as the oracle's (a test on "y != 0" would make them identically zero and gamma could never leave zero).  Both mask sources."""
    import tests.test_emu_backward as me
    orig = me.block_case

    def zeroed(*a, **k):
        cfg, p = orig(*a, **k)
        p["st_blocks.0.tc2_ln.weight"] = torch.zeros_like(p["st_blocks.0.tc2_ln.weight"])
        p["st_blocks.0.tc2_ln.bias"] = torch.zeros_like(p["st_blocks.0.tc2_ln.bias"])
        return cfg, p

    monkeypatch.setattr(me, "block_case", zeroed)
    monkeypatch.setenv("STGCN_HOOK_MASK", mask)
    test_block_backward(64, (64, 16, 64), 3, 3, "cheb_graph_conv", "glu", 17, 2, 6, True)
