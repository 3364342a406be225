"""-m gpu: the bf16 configurations (BASELINE.json configs[2] PEMS-BAY 325 nodes bs 64 bf16, configs[4] 8192 nodes Ks 5 bf16) on a real
MI355X, through the C ABI: bf16 storage of every activation / saved tensor / activation gradient, v_mfma_f32_16x16x16_bf16 temporal and
graph convolutions (v_mfma_f32_16x16x32_bf16 operator GEMMs on the tiled path), fp32 accumulation / LayerNorm statistics / parameters.
Checked against the bf16 statement of the stage oracle (oracle/stblock_stages.py QuantBf16; tolerances: tests/bf16_util.py)."""
import numpy as np
import pytest
import torch

from tests.helpers import real_gso

pytestmark = pytest.mark.gpu

SMALL = [
    (1, (64, 16, 64), 3, 3, "cheb_graph_conv", "glu", 21, 2, 7, True),
    (64, (64, 16, 64), 3, 3, "cheb_graph_conv", "glu", 17, 2, 6, True),
    (64, (64, 16, 64), 3, 3, "graph_conv", "gtu", 35, 1, 5, False),
    (32, (64, 16, 64), 3, 2, "cheb_graph_conv", "glu", 40, 1, 7, True),
    (1, (64, 16, 64), 3, 3, "cheb_graph_conv", "glu", 300, 1, 5, False),
]


def _bind():
    from tests.gpu_util import bind_hip
    return bind_hip()


@pytest.mark.parametrize("c_in,channels,Kt,Ks,gct,act,N,B,T,training", SMALL)
def test_small_cases_bf16(c_in, channels, Kt, Ks, gct, act, N, B, T, training):
    from tests.bf16_util import assert_bf16_errors, run_block_case_bf16
    _bind()
    assert_bf16_errors(*run_block_case_bf16("cuda:0", c_in, channels, Kt, Ks, gct, act, N, B, T, training))


@pytest.mark.parametrize("blk", [0, 1])
def test_c3_full_size_bf16(blk):
    """BASELINE.json configs[2] at full size: PEMS-BAY 325 nodes (the real operator), bs 64, bf16, both ST blocks, dropout on."""
    from tests.bf16_util import assert_bf16_errors, run_block_case_bf16
    _bind()
    gso = real_gso("pems_bay.cheb_sym_norm_lap")
    c_in, T = ((1, 12), (64, 8))[blk]
    assert_bf16_errors(*run_block_case_bf16("cuda:0", c_in, (64, 16, 64), 3, 3, "cheb_graph_conv", "glu", 325, 64, T, True, gso=gso))


@pytest.mark.parametrize("pp", [2, 4])
def test_c3_full_size_bf16_workgroups_per_slab(pp):
    """tc2_ln_fwd_kernel with 2 / 4 workgroups per slab (``stgcn_set_tc2ln_peers``; C3's own launches run 1) on the bf16 instances at the
    full C3 size, block 1."""
    from stgcn_amd import ops
    from tests.bf16_util import assert_bf16_errors, run_block_case_bf16
    _bind()
    gso = real_gso("pems_bay.cheb_sym_norm_lap")
    prev = ops.set_tc2ln_peers(pp)
    try:
        assert_bf16_errors(*run_block_case_bf16("cuda:0", 64, (64, 16, 64), 3, 3, "cheb_graph_conv", "glu", 325, 64, 8, True, gso=gso))
    finally:
        ops.set_tc2ln_peers(prev)


def test_c3_head_bf16():
    """The output head at the C3 size (B 64, N 325) with bf16 activations."""
    from tests.bf16_util import assert_bf16_errors, run_head_case_bf16
    _bind()
    assert_bf16_errors(*run_head_case_bf16("cuda:0", 325, 64))


@pytest.mark.parametrize("mode", ["2", "3", "4", "5", "0"])
def test_head_forward_forms_bf16(mode, monkeypatch):
    """Every form of the head's forward at the C3 size: one launch with 32-row tiles by blockIdx ("2": 650 tiles do NOT fit one resident round
    at C3 -- the launcher falls back to two launches there, and takes the form at the small size), 32-row tiles by start-order ticket ("3":
    650 tiles on ~512 resident slots: tiles wait for peers that start only after earlier tiles end), 64-row tiles ("4": what bench.py's C3
    runs), two launches ("0") -- all against the bf16 stage oracle."""
    from tests.bf16_util import assert_bf16_errors, run_head_case_bf16
    _bind()
    monkeypatch.setenv("STGCN_HEAD_FUSE", mode)
    assert_bf16_errors(*run_head_case_bf16("cuda:0", 325, 64))
    assert_bf16_errors(*run_head_case_bf16("cuda:0", 70, 2, training=False))


@pytest.mark.parametrize("blk", [0, 1])
def test_c5_graph_8192_nodes_bf16(blk):
    """BASELINE.json configs[4] graph size (8192 nodes, dense operator, ChebConv Ks = 5) at batch 1 with bf16 activations: operand-form
    GEMMs on the bf16 matrix cores, recursion / Clenshaw recurrence on stored bf16 terms, LayerNorm as a separate pass."""
    from tests.bf16_util import assert_bf16_errors, run_block_case_bf16
    from tests.emu_util import big_gso
    _bind()
    c_in, T = ((1, 6), (64, 5))[blk]
    assert_bf16_errors(*run_block_case_bf16("cuda:0", c_in, (64, 16, 64), 3, 5, "cheb_graph_conv", "glu", 8192, 1, T, True, gso=big_gso(8192, 3)))


@pytest.mark.parametrize("nt", [10, 8, 6, 5, 4])
@pytest.mark.parametrize("N,B,T", [(512, 3, 7), (1000, 5, 8)])
def test_big_operator_gemm_every_tile_width(nt, N, B, T):
    """gso_gemm_bf16_big_kernel<NT> for every NT the launcher can pick (configs[4] at bs 16 runs NT = 10 and NT = 6; batch-1
    tests only ever selected NT = 4).  512 nodes = two full row tiles; 1000 nodes = four row tiles, the last one ragged; 15 slabs = 240 and
    30 slabs = 480 operand columns: ragged against every tile width (column-overrun rows of gc_operand_alloc, LDS-transposed epilogue)."""
    from stgcn_amd import ops
    from tests.bf16_util import assert_bf16_errors, run_block_case_bf16
    _bind()
    prev = ops.set_gc_tiled_min_nodes(1)
    prev_nt = ops.set_gemm_big_nt(nt)
    try:
        from tests.emu_util import big_gso
        res = run_block_case_bf16("cuda:0", 64, (64, 16, 64), 3, 5 if N == 512 else 3, "cheb_graph_conv", "glu", N, B, T, True, gso=big_gso(N, 7))
    finally:
        ops.set_gemm_big_nt(prev_nt)
        ops.set_gc_tiled_min_nodes(prev)
    assert_bf16_errors(*res)


@pytest.mark.parametrize("blk", [0, 1])
def test_c5_full_size_bs16_bf16(blk):
    """BASELINE.json configs[4] AT ITS STATED BATCH: 8192 nodes, dense operator, ChebConv Ks = 5, bs 16, bf16, both ST blocks (block 0:
    12 -> 8 steps, 2560 operand columns = 256 x 320 tiles; block 1: 8 -> 4 steps, 1536 columns = 256 x 192 tiles), every stored tensor
    and gradient against the bf16 statement of the stage oracle (its operator products are BLAS GEMMs: about a minute of host time)."""
    from stgcn_amd import ops
    from tests.bf16_util import assert_bf16_errors, run_block_case_bf16
    from tests.emu_util import big_gso
    _bind()
    assert ops.set_gemm_big_nt(-1) == 0          # the heuristic decides, as in bench.py --config c5
    c_in, T = ((1, 12), (64, 8))[blk]
    assert_bf16_errors(*run_block_case_bf16("cuda:0", c_in, (64, 16, 64), 3, 5, "cheb_graph_conv", "glu", 8192, 16, T, True, gso=big_gso(8192, 3)))


def test_tiled_small_graph_bf16():
    from stgcn_amd import ops
    from tests.bf16_util import assert_bf16_errors, run_block_case_bf16
    _bind()
    prev = ops.set_gc_tiled_min_nodes(1)
    try:
        res = run_block_case_bf16("cuda:0", 64, (64, 16, 64), 3, 4, "cheb_graph_conv", "glu", 150, 3, 7, True)
    finally:
        ops.set_gc_tiled_min_nodes(prev)
    assert_bf16_errors(*res)


def test_c3_model_bf16_vs_reference_golden():
    """Whole model at the C3 size with bf16 activations against the fixture the reference produced in fp32 at bs 64: bf16-sized bars
    (the tight comparison is stage-level, above)."""
    import types
    from stgcn_amd import models
    from tests.helpers import cfg_from_fixture, fixture_gso, fixture_params, load_fixture
    _bind()
    name = "pemsbay_c3_b64_f32"
    fx = load_fixture(name)
    cfg = cfg_from_fixture(fx)
    dev = "cuda:0"
    gso = torch.from_numpy(fixture_gso(name, fx)).to(dev)
    args = types.SimpleNamespace(Kt=cfg.Kt, Ks=cfg.Ks, act_func=cfg.act_func, graph_conv_type=cfg.graph_conv_type, gso=gso,
                                 enable_bias=True, droprate=cfg.droprate, n_his=cfg.n_his)
    model = models.STGCNChebGraphConv(args, cfg.blocks, int(fx["n_vertex"]))
    model.load_state_dict(fixture_params(fx, cfg, torch.float32), strict=True)
    model = model.to(dev).set_compute_dtype(torch.bfloat16)
    rs = np.random.RandomState(int(fx["seed"]) + 1)
    B, N = int(fx["B"]), int(fx["n_vertex"])
    x = torch.from_numpy(rs.standard_normal((B, 1, cfg.n_his, N))).float().to(dev)
    y = torch.from_numpy(rs.standard_normal((B, N))).float().to(dev)
    model.eval()
    with torch.no_grad():
        out = model(x)
    ref = fx["eval.out"]
    err = float(np.abs(out.cpu().numpy() - ref).max()) / float(np.abs(ref).max())
    assert out.dtype == torch.float32 and err <= 3e-2, err
    model.train()
    model.zero_grad()
    loss = torch.nn.MSELoss()(model(x).view(len(x), -1), y)
    loss.backward()
    assert abs(loss.item() - float(fx["train.loss"])) <= 1e-2 * abs(float(fx["train.loss"]))
    nograd = set(str(s) for s in fx["nograd"])
    for k, prm in model.named_parameters():
        if k in nograd:
            assert prm.grad is None, k
            continue
        r = fx["gradsum." + k]
        if prm.numel() == 1:      # fc2.bias: d loss / d b = (2 / n) sum(pred - y), a difference of large sums -- absolute bar
            assert abs(float(prm.grad.double().sum()) - r[0]) <= 5e-3, k
            continue
        assert abs(float(prm.grad.double().abs().sum()) - r[1]) <= 5e-2 * r[1] + 1e-9, k


def test_graph_replay_bf16_matches_eager():
    """The captured training step (hipGraph, device-side windows of a resident series, fused step tail) with bf16 activations: finite,
    decreasing-ish losses, and the parameters stay fp32."""
    import types
    from stgcn_amd import DropoutStream, models
    from stgcn_amd.train import GraphedTrainStep, make_optimizer
    _bind()
    dev = torch.device("cuda", 0)
    N, B = 207, 32
    gso = torch.from_numpy(real_gso("metr_la.cheb_sym_norm_lap")).to(dev)
    args = types.SimpleNamespace(Kt=3, Ks=3, act_func="glu", graph_conv_type="cheb_graph_conv", gso=gso, enable_bias=True, droprate=0.5, n_his=12)
    torch.manual_seed(0)
    model = models.STGCNChebGraphConv(args, [[1], [64, 16, 64], [64, 16, 64], [128, 128], [1]], N).to(dev).set_compute_dtype(torch.bfloat16)
    model.train()
    DropoutStream.manual_seed(3)
    opt = make_optimizer(model, capturable=True)
    g = torch.Generator().manual_seed(1)
    series = torch.randn(8 * B + 24, N, generator=g).to(dev)
    x0 = torch.zeros(B, 1, 12, N, device=dev)
    y0 = torch.zeros(B, N, device=dev)
    with GraphedTrainStep(model, opt, x0, y0, series=series, n_his=12, n_pred=12) as step:
        losses = [float(step()) for _ in range(12)]
    assert all(np.isfinite(losses)), losses
    assert min(losses[6:]) < losses[0], losses
    assert all(p.dtype == torch.float32 for p in model.parameters())


@pytest.mark.parametrize("blk", [0, 1])
def test_c2_backward_bf16x3_products_inside_the_gradient_bar(blk):
    """fp32 blocks with the opt-in "bf16x3" backward products (ops.set_bwd_precision) at the full C2 size: the forward is untouched, every
    gradient stays inside the 1e-3 bar against the float64 stage oracle."""
    from stgcn_amd import ops
    from tests.test_emu_bwdx3 import run
    _bind()
    gso = real_gso("metr_la.cheb_sym_norm_lap")
    c_in, T = ((1, 12), (64, 8))[blk]
    prev = ops.set_bwd_precision("bf16x3")
    try:
        import tests.test_emu_bwdx3 as m
        orig = m.nonsym_gso
        m.nonsym_gso = lambda n, seed: gso
        try:
            err = run(c_in, (64, 16, 64), 3, 3, "cheb_graph_conv", "glu", 207, 32, T, True, dev="cuda:0")
        finally:
            m.nonsym_gso = orig
    finally:
        ops.set_bwd_precision(prev)
    assert err.pop("fwd.y") <= 1e-4
    assert max(err.values()) <= 1e-3, err
    print("bf16x3 backward errors:", {k: f"{v:.2e}" for k, v in err.items()})


@pytest.mark.parametrize("consumer", [None, "block", "head"])
def test_gpu_bf16_layernorm_backward_with_large_beta(consumer):
    """|beta| >> |gamma| (beta = 5 U, gamma = 0.05 U) -- the regime in which rebuilding sum g * xhat from
    dy * (y - keep_scale * beta) with a bf16 y loses the most -- for the stand-alone row pass and both hook epilogues (the next block's
    tc1_bwd, the head's transposed conv), at the C2 / C3 node counts with the real operators, same bars as everywhere else (tests/bf16_util.py)."""
    from tests.bf16_util import assert_bf16_errors, run_block_case_bf16
    _bind()
    N, B, name = (325, 6, "pems_bay.cheb_sym_norm_lap") if consumer == "block" else (207, 8, "metr_la.cheb_sym_norm_lap")
    T = {None: 8, "block": 10, "head": 8}[consumer]
    assert_bf16_errors(*run_block_case_bf16("cuda:0", 64, (64, 16, 64), 3, 3, "cheb_graph_conv", "glu", N, B, T, True, gso=real_gso(name),
                                            ln_scale=(0.05, 5.0), consumer=consumer))
