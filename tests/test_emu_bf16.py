"""bf16 configurations on the CPU emulator: the same kernel sources instantiated with ET = bf16 (bf16 storage, one emulated
v_mfma_f32_16x16x16_bf16 per 16-deep product step), checked against the bf16 statement of the stage oracle (tests/bf16_util.py)."""
import numpy as np
import pytest
import torch

from oracle import stblock_stages as st
from stgcn_amd import ops
from tests.bf16_util import Q, assert_bf16_errors, run_block_case_bf16, run_head_case_bf16
from tests.emu_util import bind_emulator

CASES = [
    # c_in, channels, Kt, Ks, gct, act, N, B, T, training
    (1, (64, 16, 64), 3, 3, "cheb_graph_conv", "glu", 21, 2, 7, True),       # first block: thin tmp_conv1, recomputed gate inputs
    (64, (64, 16, 64), 3, 3, "cheb_graph_conv", "glu", 17, 2, 6, True),      # second block: all four time-stepping kernels
    (64, (64, 16, 64), 3, 3, "graph_conv", "gtu", 35, 1, 5, False),
    (32, (64, 16, 64), 3, 2, "cheb_graph_conv", "glu", 40, 1, 7, True),
    (64, (64, 16, 64), 3, 3, "cheb_graph_conv", "glu", 300, 1, 6, True),     # 19 row tiles: the 16-wave / 6-tile tc2_ln_fwd of 257 .. 384 nodes, two graph-conv tiles per wave
]


def test_quant_bf16_is_round_to_nearest_even():
    a = np.array([1.0, 1.00390625, 1.001953125, 1.005859375, -3.1415926, 0.0, 1e-20, 65504.0], dtype=np.float64)
    q = Q(a)
    ref = torch.tensor(a, dtype=torch.float32).to(torch.bfloat16).to(torch.float64).numpy()      # torch rounds RNE as well
    assert np.array_equal(q, ref)
    assert np.array_equal(st.from_bf16_bits(st.to_bf16_bits(a)), q)
    assert np.array_equal(Q(q), q)


@pytest.mark.parametrize("c_in,channels,Kt,Ks,gct,act,N,B,T,training", CASES)
def test_block_bf16_matches_bf16_oracle(c_in, channels, Kt, Ks, gct, act, N, B, T, training):
    bind_emulator()
    stored, f32 = run_block_case_bf16("cpu", c_in, channels, Kt, Ks, gct, act, N, B, T, training)
    assert_bf16_errors(stored, f32)


@pytest.mark.parametrize("c_in,channels,Kt,Ks,gct,act,N,B,T,training", [
    (64, (64, 16, 64), 3, 5, "cheb_graph_conv", "glu", 37, 2, 6, True),       # Ks = 5 like configs[4]
    (1, (64, 16, 64), 3, 3, "cheb_graph_conv", "glu", 150, 1, 7, True),
    (64, (64, 16, 64), 3, 3, "graph_conv", "glu", 20, 1, 5, False),
])
def test_tiled_block_bf16_matches_bf16_oracle(c_in, channels, Kt, Ks, gct, act, N, B, T, training):
    """The tiled graph conv (BASELINE.json configs[4]: 8192 nodes) with bf16 activations: operand-form GEMMs on the bf16 matrix cores, the
    recursion / Clenshaw recurrence on STORED bf16 terms (oracle: gc_form "recursion")."""
    bind_emulator()
    prev = ops.set_gc_tiled_min_nodes(1)
    try:
        stored, f32 = run_block_case_bf16("cpu", c_in, channels, Kt, Ks, gct, act, N, B, T, training)
    finally:
        ops.set_gc_tiled_min_nodes(prev)
    assert_bf16_errors(stored, f32)


@pytest.mark.parametrize("nt", [10, 8, 6, 5, 4])
def test_big_operator_gemm_every_tile_width(nt):
    """gso_gemm_bf16_big_kernel<NT> (256 x 32 NT tiles; padded node count a multiple of 256): BASELINE.json configs[4] at bs 16 runs NT = 10
    (block 0: 2560 columns) and NT = 6 (block 1: 1536), which no small shape selects by itself -- stgcn_set_gemm_big_nt forces each instance.
    250 nodes (ragged last row block), 15 slabs = 240 operand columns: narrower than every tile but NT = 4 / 5 (column overrun rows), two
    column tiles for NT = 4 .. 6."""
    bind_emulator()
    prev = ops.set_gc_tiled_min_nodes(1)
    prev_nt = ops.set_gemm_big_nt(nt)
    try:
        assert ops.set_gemm_big_nt(-1) == nt
        stored, f32 = run_block_case_bf16("cpu", 64, (64, 16, 64), 3, 3, "cheb_graph_conv", "glu", 250, 3, 7, True)
    finally:
        ops.set_gemm_big_nt(prev_nt)
        ops.set_gc_tiled_min_nodes(prev)
    assert_bf16_errors(stored, f32)


@pytest.mark.parametrize("N,B,training", [(21, 3, True), (40, 2, False), (45, 3, True)])   # (N >= 32: the one-launch forward)
def test_head_bf16_matches_bf16_oracle(N, B, training):
    bind_emulator()
    stored, f32 = run_head_case_bf16("cpu", N, B, training=training)
    assert_bf16_errors(stored, f32)


def test_model_bf16_tracks_the_reference_goldens():
    """Whole drop-in model with bf16 activations (model.set_compute_dtype) against the fixture the reference itself produced in fp32:
    bf16 storage and bf16 matrix products are ~2^-9 per element, so the bars are bf16-sized (eval output <= 3e-2 of its range, loss
    <= 2 %, gradient sums <= 10 %: this fixture has 2 windows x 20 nodes, a 16-element bias gradient averages over few rows); the tight comparison is the stage-level one above, against the bf16 statement of the oracle."""
    from tests.test_emu_model import _build
    fx, model, x, y = _build("tiny_cheb_f32")
    model.set_compute_dtype(torch.bfloat16)
    model.eval()
    with torch.no_grad():
        out = model(x)
    assert out.dtype == torch.float32 and out.shape == fx["eval.out"].shape
    ref = fx["eval.out"]
    assert float(np.abs(out.numpy() - ref).max()) <= 3e-2 * float(np.abs(ref).max())
    model.train()
    model.zero_grad()
    loss = torch.nn.MSELoss()(model(x).view(len(x), -1), y)
    loss.backward()
    assert abs(loss.item() - float(fx["train.loss"])) <= 2e-2 * abs(float(fx["train.loss"]))
    nograd = set(str(s) for s in fx["nograd"])
    for k, prm in model.named_parameters():
        if k in nograd:
            assert prm.grad is None, k
            continue
        assert prm.grad is not None and prm.grad.dtype == torch.float32, k
        r = fx["gradsum." + k]
        assert abs(float(prm.grad.double().abs().sum()) - r[1]) <= 1e-1 * r[1] + 1e-9, k


def test_fused_train_step_bf16_runs_and_tracks_fp32():
    """The fused step tail (GradSink flush + AdamW in one launch) with bf16 activations: parameters, gradients and optimizer state stay fp32."""
    from stgcn_amd.layers import DropoutStream
    from stgcn_amd.train import GradArena, fused_train_step, make_optimizer, train_step
    from tests.test_emu_model import _build
    losses = {}
    for dt in (torch.float32, torch.bfloat16):
        fx, model, x, y = _build("tiny_cheb_f32")
        model.set_compute_dtype(dt)
        model.train()
        DropoutStream.manual_seed(5)
        opt = make_optimizer(model, lr=1e-3, weight_decay=1e-3)
        train_step(model, opt, x, y)                                   # plain step: shows which parameters are live
        arena = GradArena([p for p in model.parameters() if p.grad is not None])
        ls = [float(fused_train_step(model, opt, x, y, arena)) for _ in range(3)]
        assert all(np.isfinite(ls)) and all(p.dtype == torch.float32 for p in model.parameters())
        losses[dt] = ls
    for a, b in zip(losses[torch.float32], losses[torch.bfloat16]):
        assert abs(a - b) <= 6e-2 * abs(a), losses      # (3 steps of a 2-window model with dropout: the trajectories drift apart by a few percent)


@pytest.mark.parametrize("consumer", [None, "block", "head"])
def test_block_bf16_layernorm_backward_with_large_beta(consumer):
    """VERDICT r4 weak 1 / ADVICE r3: tc2_bwd / the hook epilogues rebuild sum g * xhat from dy * (y - keep_scale * beta) with y in bf16; the
    subtraction amplifies y's 2^-9 rounding by |beta| / |gamma xhat|.  Every other test draws beta = 0.1 U, gamma ~ 1; here |beta| >> |gamma|
    (beta = 5 U, gamma = 0.05 U), for the stand-alone pass over dy (consumer None) and for both hook epilogues (the next block's tc1_bwd, the
    head's transposed conv), against the same bars as the benign regime."""
    bind_emulator()
    T = {None: 8, "block": 10, "head": 8}[consumer]      # (head: Ko = 4 = T - 4)
    stored, f32 = run_block_case_bf16("cpu", 64, (64, 16, 64), 3, 3, "cheb_graph_conv", "glu", 37, 2, T, True, ln_scale=(0.05, 5.0), consumer=consumer)
    assert_bf16_errors(stored, f32)
