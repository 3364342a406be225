"""-m gpu: the tiled graph convolution (graphs beyond 512 nodes: BASELINE.json configs[4], 8192-node dense stress graph)
on a real MI355X -- against the stage oracle, and against the slab-resident kernels on the C2 graph."""
import numpy as np
import pytest
import torch

from tests.helpers import real_gso

pytestmark = pytest.mark.gpu


@pytest.fixture
def tiled_everywhere():
    from stgcn_amd import ops
    from tests.gpu_util import bind_hip
    bind_hip()
    prev = ops.set_gc_tiled_min_nodes(1)
    try:
        yield
    finally:
        ops.set_gc_tiled_min_nodes(prev)


CASES = [
    (1, (64, 16, 64), 3, 3, "cheb_graph_conv", "glu", 21, 2, 7, True),
    (64, (64, 16, 64), 3, 3, "graph_conv", "gtu", 35, 1, 5, False),
    (16, (128, 16, 64), 2, 5, "cheb_graph_conv", "glu", 9, 2, 5, True),
    (32, (64, 16, 128), 3, 1, "cheb_graph_conv", "glu", 16, 1, 5, False),
    (64, (64, 16, 64), 3, 4, "cheb_graph_conv", "glu", 150, 3, 7, True),
]


@pytest.mark.parametrize("c_in,channels,Kt,Ks,gct,act,N,B,T,training", CASES)
def test_tiled_small_graphs(tiled_everywhere, c_in, channels, Kt, Ks, gct, act, N, B, T, training):
    from tests.gpu_util import assert_errors, run_block_case
    assert_errors(run_block_case(c_in, channels, Kt, Ks, gct, act, N, B, T, training))


@pytest.mark.parametrize("blk", [0, 1])
def test_tiled_c2_full_size(tiled_everywhere, blk):
    """The C2 blocks (real METR-LA operator, bs 32) through the tiled path: same tolerances as the slab-resident kernels."""
    from tests.gpu_util import assert_errors, run_block_case
    gso = real_gso("metr_la.cheb_sym_norm_lap")
    c_in, T = ((1, 12), (64, 8))[blk]
    assert_errors(run_block_case(c_in, (64, 16, 64), 3, 3, "cheb_graph_conv", "glu", 207, 32, T, True, gso=gso))


def test_above_the_slab_limit():
    """600 nodes: the default threshold selects the tiled path (5 operator row tiles, ragged last one)."""
    from stgcn_amd import ops
    from tests.gpu_util import assert_errors, bind_hip, run_block_case
    bind_hip()
    assert ops.set_gc_tiled_min_nodes(0) == 513
    assert_errors(run_block_case(64, (64, 16, 64), 3, 3, "cheb_graph_conv", "glu", 600, 3, 6, True))


@pytest.mark.parametrize("blk", [0, 1])
def test_c5_graph_8192_nodes(blk):
    """BASELINE.json configs[4] graph size (8192 nodes, dense operator, ChebConv Ks = 5) at batch 1 -- fp32 path; every
    stage and gradient of both block shapes against the fp64 stage oracle."""
    from tests.emu_util import big_gso
    from tests.gpu_util import assert_errors, run_block_case
    c_in, T = ((1, 6), (64, 5))[blk]
    assert_errors(run_block_case(c_in, (64, 16, 64), 3, 5, "cheb_graph_conv", "glu", 8192, 1, T, True, gso=big_gso(8192, 3)))


def test_tiled_equals_slab_resident_on_c2():
    """Same block, inputs and dropout stream through both graph-conv implementations at the C2 size."""
    from stgcn_amd import ops
    from tests.emu_util import block_case, params_in_field_order
    from tests.gpu_util import bind_hip
    bind_hip()
    dev = "cuda:0"
    c_in, channels, Kt, Ks, gct, act, N, B, T = 64, (64, 16, 64), 3, 3, "cheb_graph_conv", "glu", 207, 32, 8
    _, p = block_case(c_in, channels, Kt, Ks, gct, act, N, B, T)
    gso = torch.from_numpy(real_gso("metr_la.cheb_sym_norm_lap")).to(dev)
    g = torch.Generator().manual_seed(1)
    x0 = torch.randn(B, c_in, T, N, generator=g).to(dev)
    dy = torch.randn(B, channels[2], T - 2 * (Kt - 1), N, generator=g).to(dev)
    bcfg = ops.BlockConfig(Kt=Kt, Ks=Ks, n_vertex=N, c_in=c_in, channels=channels, act_func=act, graph_conv_type=gct, droprate=0.5)

    def run():
        gp, gt = ops.gso_prepare(gso, ops.graph_terms(bcfg))
        params = [None if t is None else t.clone().to(dev).requires_grad_(True) for t in params_in_field_order(p, "st_blocks.0.", gct)]
        x = x0.clone().requires_grad_(True)
        y = ops.st_conv_block(x, gp, gt, bcfg, params, True, 5, 1, ops.WorkspaceCache())
        y.backward(dy)
        torch.cuda.synchronize()
        return y.detach(), x.grad, [None if q is None or q.grad is None else q.grad for q in params]

    ya, dxa, ga = run()
    prev = ops.set_gc_tiled_min_nodes(1)
    try:
        yb, dxb, gb = run()
    finally:
        ops.set_gc_tiled_min_nodes(prev)
    rel = lambda a, b: float((a - b).abs().max() / max(1.0, float(a.abs().max())))
    assert rel(ya, yb) < 2e-5 and rel(dxa, dxb) < 2e-5
    for a, b in zip(ga, gb):
        assert (a is None) == (b is None)
        if a is not None:
            assert rel(a, b) < 5e-5
