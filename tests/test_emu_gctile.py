"""Tiled graph convolution (stgcn_kernels_gctile.hip.h: one GEMM launch per operator term + row passes, Clenshaw backward)
on the CPU emulator.  The node threshold of the tiled path is lowered so that small graphs run it, every stage and
gradient is checked against the numpy stage oracle through the same test bodies as the slab-resident kernels, and both
paths are compared with each other; one case runs above the real threshold (N > 512)."""
import numpy as np
import pytest
import torch

from stgcn_amd import ops
from tests.emu_util import bind_emulator, block_case, nonsym_gso, params_in_field_order
from tests.test_emu_backward import test_block_backward as run_backward_case


@pytest.fixture
def tiled_everywhere():
    bind_emulator()
    prev = ops.set_gc_tiled_min_nodes(1)
    try:
        yield
    finally:
        ops.set_gc_tiled_min_nodes(prev)
        assert ops.set_gc_tiled_min_nodes(0) == prev


CASES = [
    # c_in, channels, Kt, Ks, gct, act, N, B, T, training
    (1, (64, 16, 64), 3, 3, "cheb_graph_conv", "glu", 21, 2, 7, True),
    (64, (64, 16, 64), 3, 3, "graph_conv", "gtu", 35, 1, 5, False),           # Kipf: X_1 = A_hat X_0, one weight
    (16, (128, 16, 64), 2, 5, "cheb_graph_conv", "glu", 9, 2, 5, True),       # 5 terms: three Clenshaw steps
    (32, (64, 16, 128), 3, 1, "cheb_graph_conv", "glu", 16, 1, 5, False),     # single term: no GEMM at all
    (128, (64, 16, 64), 3, 2, "cheb_graph_conv", "glu", 10, 1, 5, True),
    (64, (64, 16, 64), 3, 4, "cheb_graph_conv", "glu", 150, 3, 7, True),      # 2 operator row tiles, 15 slabs = 2 column tiles (ragged)
]


@pytest.mark.parametrize("c_in,channels,Kt,Ks,gct,act,N,B,T,training", CASES)
def test_tiled_block_against_stage_oracle(tiled_everywhere, c_in, channels, Kt, Ks, gct, act, N, B, T, training):
    bcfg = ops.BlockConfig(Kt=Kt, Ks=Ks, n_vertex=N, c_in=c_in, channels=tuple(channels), act_func=act, graph_conv_type=gct, droprate=0.5)
    plan = ops.query_plan(ops.make_desc(bcfg, B, T, training, c_in > 1))
    assert plan.tiled_gc == 1 and plan.NP % 128 == 0 and plan.NP >= N
    run_backward_case(c_in, channels, Kt, Ks, gct, act, N, B, T, training)


def test_tiled_operator_layout(tiled_everywhere):
    gso = nonsym_gso(37, 2)
    gp, gt = ops.gso_prepare(torch.from_numpy(gso), 3)
    assert gp.shape == (1, 128, 128) and gt.shape == (1, 128, 128)
    assert np.array_equal(gp[0, :37, :37].numpy(), gso) and np.array_equal(gt[0, :37, :37].numpy(), gso.T)
    assert gp[0, 37:].abs().sum() == 0 and gp[0, :, 37:].abs().sum() == 0 and gt[0, 37:].abs().sum() == 0 and gt[0, :, 37:].abs().sum() == 0


def _run(c_in, channels, Kt, Ks, gct, act, N, B, T, x_np, dy_np, p, gso):
    bcfg = ops.BlockConfig(Kt=Kt, Ks=Ks, n_vertex=N, c_in=c_in, channels=tuple(channels), act_func=act, graph_conv_type=gct, droprate=0.5)
    gp, gt = ops.gso_prepare(torch.from_numpy(gso), ops.graph_terms(bcfg))
    params = [None if t is None else t.clone().requires_grad_(True) for t in params_in_field_order(p, "st_blocks.0.", gct)]
    x = torch.from_numpy(x_np).requires_grad_(True)
    y = ops.st_conv_block(x, gp, gt, bcfg, params, True, 5, 1, ops.WorkspaceCache())
    y.backward(torch.from_numpy(dy_np))
    return y.detach().numpy(), x.grad.numpy(), [None if q is None or q.grad is None else q.grad.numpy() for q in params]


@pytest.mark.parametrize("gct,Ks", [("cheb_graph_conv", 3), ("graph_conv", 1)])
def test_tiled_equals_slab_resident(gct, Ks):
    """Same block, same inputs, same dropout stream through both graph-conv implementations."""
    bind_emulator()
    c_in, channels, Kt, act, N, B, T = 64, (64, 16, 64), 3, "glu", 45, 2, 6
    _, p = block_case(c_in, channels, Kt, Ks, gct, act, N, B, T)
    gso = nonsym_gso(N, 9)
    rs = np.random.RandomState(4)
    x_np = rs.standard_normal((B, c_in, T, N)).astype(np.float32)
    dy_np = rs.standard_normal((B, channels[2], T - 2 * (Kt - 1), N)).astype(np.float32)
    ya, dxa, ga = _run(c_in, channels, Kt, Ks, gct, act, N, B, T, x_np, dy_np, p, gso)
    prev = ops.set_gc_tiled_min_nodes(1)
    try:
        yb, dxb, gb = _run(c_in, channels, Kt, Ks, gct, act, N, B, T, x_np, dy_np, p, gso)
    finally:
        ops.set_gc_tiled_min_nodes(prev)
    assert np.abs(ya - yb).max() < 2e-5
    assert np.abs(dxa - dxb).max() < 2e-5 * max(1.0, np.abs(dxa).max())
    for a, b in zip(ga, gb):
        assert (a is None) == (b is None)
        if a is not None:
            assert np.abs(a - b).max() < 2e-5 * max(1.0, np.abs(a).max())


def test_default_threshold_selects_tiled_path_above_512_nodes():
    """N = 530 runs the tiled path without any knob (the slab-resident kernels stop at 512 nodes)."""
    bind_emulator()
    assert ops.set_gc_tiled_min_nodes(0) == 513
    bcfg = ops.BlockConfig(Kt=3, Ks=3, n_vertex=530, c_in=1, channels=(64, 16, 64), act_func="glu", graph_conv_type="cheb_graph_conv",
                           droprate=0.5)
    assert ops.query_plan(ops.make_desc(bcfg, 1, 5, True, False)).tiled_gc == 1
    bcfg = ops.BlockConfig(Kt=3, Ks=3, n_vertex=512, c_in=1, channels=(64, 16, 64), act_func="glu", graph_conv_type="cheb_graph_conv",
                           droprate=0.5)
    assert ops.query_plan(ops.make_desc(bcfg, 1, 5, True, False)).tiled_gc == 0
    run_backward_case(1, (64, 16, 64), 3, 3, "cheb_graph_conv", "glu", 530, 1, 5, True)


@pytest.mark.parametrize("name", ["tiny_cheb_f32", "tiny_gc_f32", "tiny_ks5_f32"])
def test_tiled_model_matches_reference_golden(tiled_everywhere, name):
    """Whole drop-in model through the tiled graph conv against the golden fixtures the reference itself produced."""
    from tests.test_emu_model import test_model_matches_reference_golden as run_model_case
    run_model_case(name)
