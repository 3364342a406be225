"""-m gpu: the fused ST-Conv block on a real MI355X, every stage against the CPU oracle (through the C ABI)."""
import numpy as np
import pytest
import torch

from tests.helpers import real_gso

pytestmark = pytest.mark.gpu

SMALL = [
    (1, (64, 16, 64), 3, 3, "cheb_graph_conv", "glu", 21, 2, 7, True),
    (64, (64, 16, 64), 3, 3, "cheb_graph_conv", "glu", 17, 2, 6, True),
    (64, (64, 16, 64), 3, 3, "graph_conv", "gtu", 35, 1, 5, False),
    (16, (128, 16, 64), 2, 5, "cheb_graph_conv", "glu", 9, 2, 5, True),
    (32, (64, 16, 128), 3, 1, "cheb_graph_conv", "glu", 16, 1, 5, False),
    (128, (64, 16, 64), 3, 2, "cheb_graph_conv", "glu", 10, 1, 5, True),
    (64, (64, 16, 64), 3, 3, "cheb_graph_conv", "glu", 1, 1, 5, True),        # single vertex
    (1, (64, 16, 64), 3, 3, "cheb_graph_conv", "glu", 300, 1, 5, False),      # 19 node tiles (MAXQ 6 path), ragged rows
    (4, (64, 16, 64), 3, 3, "cheb_graph_conv", "gtu", 33, 2, 6, True),        # thin first layer (K = 12) with an input gradient
    (2, (64, 16, 64), 2, 3, "cheb_graph_conv", "glu", 19, 2, 5, True),        # K = Kt * c_in = 4: the wave-per-tile thin kernels with four taps, dZ1 written for dx
]


@pytest.mark.parametrize("c_in,channels,Kt,Ks,gct,act,N,B,T,training", SMALL)
def test_small_cases(c_in, channels, Kt, Ks, gct, act, N, B, T, training):
    from tests.gpu_util import assert_errors, run_block_case
    assert_errors(run_block_case(c_in, channels, Kt, Ks, gct, act, N, B, T, training))


@pytest.mark.parametrize("blk", [0, 1])
@pytest.mark.parametrize("training", [False, True])
def test_c2_full_size(blk, training):
    """BASELINE.json configs[1] at full size: METR-LA 207 nodes (real GSO), bs 32, both ST blocks."""
    from tests.gpu_util import assert_errors, run_block_case
    gso = real_gso("metr_la.cheb_sym_norm_lap")
    c_in, T = ((1, 12), (64, 8))[blk]
    assert_errors(run_block_case(c_in, (64, 16, 64), 3, 3, "cheb_graph_conv", "glu", 207, 32, T, training, gso=gso))


@pytest.mark.parametrize("blk", [0, 1])
@pytest.mark.parametrize("pp", [1, 2, 4])
def test_c2_full_size_every_workgroups_per_slab(blk, pp):
    """tc2_ln_fwd_kernel with 1 / 2 / 4 workgroups per (b, t) slab (round 6, ``stgcn_set_tc2ln_peers``; by default block 1 of C2 runs 2, small
    batches 4) at the full C2 size, training: every stage of the block against the fp64 stage oracle.  At PP = 4 block 0 is 1024 workgroups on
    512 resident slots: parts wait for peers that only start when earlier workgroups end (the ticket order makes that safe)."""
    from stgcn_amd import ops
    from tests.gpu_util import assert_errors, bind_hip, run_block_case
    bind_hip()
    gso = real_gso("metr_la.cheb_sym_norm_lap")
    c_in, T = ((1, 12), (64, 8))[blk]
    prev = ops.set_tc2ln_peers(pp)
    try:
        assert_errors(run_block_case(c_in, (64, 16, 64), 3, 3, "cheb_graph_conv", "glu", 207, 32, T, True, gso=gso))
    finally:
        ops.set_tc2ln_peers(prev)


@pytest.mark.parametrize("pp", [2, 4])
def test_c3_shapes_fp32_workgroups_per_slab(pp):
    """The peer instances of 257 .. 384 nodes (3 / 2 node tiles per wave) at the C3 shapes: 325 nodes = 21 node tiles, bs 64, fp32."""
    from stgcn_amd import ops
    from tests.gpu_util import assert_errors, bind_hip, run_block_case
    bind_hip()
    gso = real_gso("pems_bay.cheb_sym_norm_lap")
    prev = ops.set_tc2ln_peers(pp)
    try:
        assert_errors(run_block_case(64, (64, 16, 64), 3, 3, "cheb_graph_conv", "glu", 325, 64, 8, True, gso=gso))
    finally:
        ops.set_tc2ln_peers(prev)


def test_tc2_ln_peer_wait_give_up_on_the_device():
    """The bounded in-launch wait of a slab's parts on the hardware: NaN for the starved part only, the sticky word names the slab, the next
    launch is clean and bitwise equal to the first (tests/test_emu_forward.py::tc2_ln_peer_give_up_case)."""
    from tests.gpu_util import bind_hip
    from tests.test_emu_forward import tc2_ln_peer_give_up_case
    bind_hip()
    tc2_ln_peer_give_up_case("cuda:0")


def test_thin_first_layer_row_tile_kernels_on_the_device(monkeypatch):
    """STGCN_THIN=0: the thin first layer on the row-tile kernels of rounds 1 - 4 (still in the library as the A/B form of the round-5
    wave-per-tile kernels, which every other test of a 1-channel block runs): full C2 block 0 and a small K = 4 case."""
    from tests.gpu_util import assert_errors, run_block_case
    monkeypatch.setenv("STGCN_THIN", "0")
    gso = real_gso("metr_la.cheb_sym_norm_lap")
    assert_errors(run_block_case(1, (64, 16, 64), 3, 3, "cheb_graph_conv", "glu", 207, 32, 12, True, gso=gso))
    assert_errors(run_block_case(2, (64, 16, 64), 2, 3, "cheb_graph_conv", "glu", 19, 2, 5, True))


def test_ks5_slab_path_more_slabs_than_one_round():
    """207 nodes, Ks = 5 / 4, 320 / 384 slabs -- no split of the graph-conv backward fits one round, so the one-part geometry runs with
    8 tile waves and its parameter-gradient jobs on the 4 waves the launch bounds leave (it used to ask for 6 and fail to launch)."""
    from tests.gpu_util import assert_errors, run_block_case
    gso = real_gso("metr_la.cheb_sym_norm_lap")      # (symmetric, spectrum in [-1, 1]: the higher polynomials stay bounded)
    assert_errors(run_block_case(1, (64, 16, 64), 3, 5, "cheb_graph_conv", "glu", 207, 32, 12, True, gso=gso))
    assert_errors(run_block_case(64, (64, 16, 64), 3, 4, "cheb_graph_conv", "glu", 207, 64, 8, True, gso=gso))


def test_c1_shapes_kipf():
    """BASELINE.json configs[0] shapes (PeMSD7(M) 228 nodes, graph_conv, bs 8) on the GPU path."""
    from tests.gpu_util import assert_errors, run_block_case
    gso = real_gso("pemsd7_m.sym_renorm_adj")
    assert_errors(run_block_case(64, (64, 16, 64), 3, 3, "graph_conv", "glu", 228, 8, 8, True, gso=gso))


def test_c3_shapes_fp32():
    """BASELINE.json configs[2] shapes (PEMS-BAY 325 nodes, bs 64) -- fp32 path (bf16 storage is a later round)."""
    from tests.gpu_util import assert_errors, run_block_case
    gso = real_gso("pems_bay.cheb_sym_norm_lap")
    assert_errors(run_block_case(64, (64, 16, 64), 3, 3, "cheb_graph_conv", "glu", 325, 64, 8, True, gso=gso))


def test_errors_are_loud():
    from stgcn_amd import ops
    from tests.gpu_util import bind_hip
    bind_hip()
    cfg = ops.BlockConfig(Kt=3, Ks=3, n_vertex=20, c_in=64, channels=(64, 16, 64), act_func="glu", graph_conv_type="cheb_graph_conv",
                          droprate=0.5)
    x_cpu = torch.zeros(1, 64, 8, 20)
    with pytest.raises(RuntimeError):          # CPU tensor into the HIP library: no fallback
        ops.st_conv_block(x_cpu, None, None, cfg, [None] * 14, False, 0, 0, ops.WorkspaceCache())
    bad = ops.BlockConfig(Kt=3, Ks=0, n_vertex=20, c_in=64, channels=(64, 16, 64), act_func="glu", graph_conv_type="cheb_graph_conv",
                          droprate=0.5)
    with pytest.raises(ValueError):            # layers.py:147-148
        ops.query_plan(ops.make_desc(bad, 1, 8, False, True))
    with pytest.raises(NotImplementedError):   # layers.py:117-118
        ops.make_desc(ops.BlockConfig(Kt=3, Ks=3, n_vertex=20, c_in=64, channels=(64, 16, 64), act_func="relu6",
                                      graph_conv_type="cheb_graph_conv", droprate=0.5), 1, 8, False, True)
