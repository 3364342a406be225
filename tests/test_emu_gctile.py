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
    (64, (64, 16, 64), 3, 4, "cheb_graph_conv", "glu", 140, 2, 6, True),      # 2 operator row tiles (ragged), 8 slabs
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
    assert gp.shape == (2, 128, 128) and gt.shape == (2, 128, 128)      # fp32 matrix, then its bf16 hi / lo planes
    assert np.array_equal(gp[0, :37, :37].numpy(), gso) and np.array_equal(gt[0, :37, :37].numpy(), gso.T)
    assert gp[0, 37:].abs().sum() == 0 and gp[0, :, 37:].abs().sum() == 0 and gt[0, 37:].abs().sum() == 0 and gt[0, :, 37:].abs().sum() == 0
    for planes, ref in ((gp[1], gso), (gt[1], gso.T)):
        u = planes.numpy().view(np.uint16).reshape(2, 128, 128)
        hi = (u[0].astype(np.uint32) << 16).view(np.float32)
        lo = (u[1].astype(np.uint32) << 16).view(np.float32)
        assert np.abs(hi[:37, :37] - ref).max() <= 2.0 ** -8 * np.abs(ref).max()              # bf16 rounding
        assert np.abs(hi[:37, :37] + lo[:37, :37] - ref).max() <= 2.0 ** -16 * np.abs(ref).max()   # split: 16 bits of mantissa
        assert np.abs(hi[37:]).sum() == 0 and np.abs(hi[:, 37:]).sum() == 0 and np.abs(lo[37:]).sum() == 0 and np.abs(lo[:, 37:]).sum() == 0


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
    """N > 512 selects the tiled path without any knob (the slab-resident kernels stop at 512 nodes); the run itself at that
    size is test_layernorm_backward_on_big_slabs (516 nodes, default threshold, every stage against the oracle)."""
    bind_emulator()
    assert ops.set_gc_tiled_min_nodes(0) == 513
    bcfg = ops.BlockConfig(Kt=3, Ks=3, n_vertex=530, c_in=1, channels=(64, 16, 64), act_func="glu", graph_conv_type="cheb_graph_conv",
                           droprate=0.5)
    assert ops.query_plan(ops.make_desc(bcfg, 1, 5, True, False)).tiled_gc == 1
    bcfg = ops.BlockConfig(Kt=3, Ks=3, n_vertex=512, c_in=1, channels=(64, 16, 64), act_func="glu", graph_conv_type="cheb_graph_conv",
                           droprate=0.5)
    assert ops.query_plan(ops.make_desc(bcfg, 1, 5, True, False)).tiled_gc == 0
    bcfg = ops.BlockConfig(Kt=3, Ks=5, n_vertex=512, c_in=1, channels=(64, 16, 64), act_func="glu", graph_conv_type="cheb_graph_conv",
                           droprate=0.5)
    assert ops.query_plan(ops.make_desc(bcfg, 1, 5, True, False)).tiled_gc == 1      # 5 terms x 512 nodes exceed the slab kernel's LDS


@pytest.mark.parametrize("name", ["tiny_cheb_f32", "tiny_ks5_f32"])
def test_tiled_model_matches_reference_golden(tiled_everywhere, name):
    """Whole drop-in model through the tiled graph conv against the golden fixtures the reference itself produced."""
    from tests.test_emu_model import test_model_matches_reference_golden as run_model_case
    run_model_case(name)


@pytest.fixture
def precision():
    """ops.set_gc_precision(...) for one test, restored afterwards."""
    prev = {}

    def use(mode):
        prev.setdefault("mode", ops.set_gc_precision(mode))
    try:
        yield use
    finally:
        if "mode" in prev:
            ops.set_gc_precision(prev["mode"])


def _block_inputs(gct, Ks, N, B, T):
    c_in, channels, Kt, act = 64, (64, 16, 64), 3, "glu"
    _, p = block_case(c_in, channels, Kt, Ks, gct, act, N, B, T)
    gso = nonsym_gso(N, 9)
    rs = np.random.RandomState(4)
    x_np = rs.standard_normal((B, c_in, T, N)).astype(np.float32)
    dy_np = rs.standard_normal((B, channels[2], T - 2 * (Kt - 1), N)).astype(np.float32)
    return (c_in, channels, Kt, Ks, gct, act, N, B, T, x_np, dy_np, p, gso)


_rms = lambda a: float(np.sqrt((np.asarray(a, np.float64) ** 2).mean()))
_rel = lambda a, b: float(np.abs(a - b).max() / max(1e-30, np.abs(a).max()))

SHAPES = [("cheb_graph_conv", 3, 45, 2, 6), ("cheb_graph_conv", 5, 130, 2, 5), ("graph_conv", 1, 45, 2, 6)]      # 130 nodes: 2 ragged row tiles


@pytest.mark.parametrize("gct,Ks,N,B,T", SHAPES[1:])
def test_bf16x3_operator_products_track_fp32(tiled_everywhere, precision, gct, Ks, N, B, T):
    """Split-bf16 operands (three bf16 MFMAs per product, fp32 accumulation): ~2^-17 relative per product.  Against the
    exact-fp32 path on an operator with row sums up to 6 (worse than any rescaled Laplacian): block output within 1e-3
    abs after five recursion steps (1.5e-4 for Ks <= 3), every gradient within the 1e-3 relative bar of the fp32 configs."""
    args = _block_inputs(gct, Ks, N, B, T)
    ya, dxa, ga = _run(*args)
    precision("bf16x3")
    yb, dxb, gb = _run(*args)
    assert 0 < np.abs(ya - yb).max() < (1e-3 if Ks > 3 else 1.5e-4)          # > 0: the bf16 kernels really ran
    assert _rel(dxa, dxb) < 1e-3
    for a, b in zip(ga, gb):
        assert (a is None) == (b is None)
        if a is not None:
            assert _rel(a, b) < 1e-3


@pytest.mark.parametrize("gct,Ks,N,B,T", SHAPES[:1])
def test_bf16_operator_products_track_fp32(tiled_everywhere, precision, gct, Ks, N, B, T):
    """Plain bf16 operands (~3 significant digits per product): block output within 1 % rms of the fp32 path; gradients
    upstream of the ReLU see its mask flip on ~0.3 % of the elements, i.e. ~6 % rms (inherent to a perturbed forward)."""
    args = _block_inputs(gct, Ks, N, B, T)
    ya, dxa, ga = _run(*args)
    precision("bf16")
    yb, dxb, gb = _run(*args)
    assert 0 < _rms(ya - yb) < 1e-2 * _rms(ya)
    assert _rms(dxa - dxb) < 0.15 * _rms(dxa)
    for a, b in zip(ga, gb):
        assert (a is None) == (b is None)
        if a is not None:
            assert _rms(a - b) <= 0.15 * _rms(a)


@pytest.fixture
def ld_pad():
    prev = {}

    def use(pad):
        prev.setdefault("pad", ops.set_gc_ld_pad(pad))
    try:
        yield use
    finally:
        if "pad" in prev:
            ops.set_gc_ld_pad(prev["pad"])


def test_padded_planes_give_identical_results(tiled_everywhere, precision, ld_pad):
    """stgcn_set_gc_ld_pad only moves rows apart: bf16x3 block output and gradients are bitwise those of the unpadded planes."""
    args = _block_inputs("cheb_graph_conv", 4, 45, 2, 6)
    precision("bf16x3")
    ya, dxa, ga = _run(*args)
    ld_pad(2176)
    yb, dxb, gb = _run(*args)
    assert np.array_equal(ya, yb) and np.array_equal(dxa, dxb)
    for a, b in zip(ga, gb):
        assert (a is None) == (b is None) and (a is None or np.array_equal(a, b))


@pytest.mark.parametrize("pad", [0, 72])
def test_bf16_gemm_equals_rounded_operand_product(tiled_everywhere, precision, ld_pad, pad):
    """The bf16 kernel itself, exactly: X_1 = bf16(L) bf16(X_0) and X_2 = 2 bf16(L) bf16(X_1) - X_0 with fp32 accumulation,
    checked against numpy on the rounded operands (only the summation order differs)."""
    import ctypes as C
    from stgcn_amd import _lib

    def bf16(a):      # round to nearest even, as bf16_rne in the kernels
        u = np.ascontiguousarray(a, np.float32).view(np.uint32).astype(np.uint64)
        return (((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16).astype(np.uint32).view(np.float32)

    c_in, channels, Kt, Ks, gct, act, N, B, T, x_np, _, p, gso = _block_inputs("cheb_graph_conv", 3, 130, 3, 5)
    L = _lib.lib()
    bcfg = ops.BlockConfig(Kt=Kt, Ks=Ks, n_vertex=N, c_in=c_in, channels=channels, act_func=act, graph_conv_type=gct, droprate=0.5)
    desc = ops.make_desc(bcfg, B, T, training=False, need_dx=True)
    plan = ops.query_plan(desc)
    precision("bf16")
    ld_pad(pad)
    plan = ops.query_plan(desc)
    gp, _ = ops.gso_prepare(torch.from_numpy(gso), 3)
    pst = ops._param_struct(_lib.StblockParams, params_in_field_order(p, "st_blocks.0.", gct))
    x_cl = torch.from_numpy(np.ascontiguousarray(x_np.transpose(0, 2, 3, 1)))
    y = torch.zeros(B, plan.T2, N, channels[2])
    saved = torch.zeros(plan.saved_floats)
    ws = torch.zeros(plan.ws_floats)
    L.check(L.dll.stgcn_stblock_forward(C.byref(desc), C.byref(pst), x_cl.data_ptr(), gp.data_ptr(), y.data_ptr(), saved.data_ptr(),
                                        ws.data_ptr(), 1, 1, None, None), "fwd")
    n = B * plan.T1 * N * 16
    shape = (B * plan.T1, N, 16)
    A = saved[plan.sv_A:plan.sv_A + n].numpy().reshape(shape)
    X1 = saved[plan.sv_Xk:plan.sv_Xk + n].numpy().reshape(shape)
    X2 = saved[plan.sv_Xk + n:plan.sv_Xk + 2 * n].numpy().reshape(shape)
    Lr = bf16(gso).astype(np.float64)
    X1_ref = np.einsum("hi,sic->shc", Lr, bf16(A).astype(np.float64))
    X2_ref = 2.0 * np.einsum("hi,sic->shc", Lr, bf16(X1).astype(np.float64)) - A
    assert np.abs(X1 - X1_ref).max() < 2e-5 and np.abs(X2 - X2_ref).max() < 4e-5
    assert np.abs(X1 - np.einsum("hi,sic->shc", gso.astype(np.float64), A.astype(np.float64))).max() > 1e-4      # and it IS bf16


def test_layernorm_backward_on_big_slabs():
    """N * C / 4 >= 64 * 256 float4 columns per slab (here 520 nodes x 128 channels): the slab constants of the LayerNorm
    backward come from ln_slab_consts_kernel instead of every workgroup's own rebuild -- ST block and output head."""
    bind_emulator()
    assert ops.set_gc_tiled_min_nodes(0) == 513
    run_backward_case(32, (64, 16, 128), 3, 2, "cheb_graph_conv", "gtu", 516, 2, 5, True)      # (seeded; no ReLU input within 1e-6 of zero)
    from tests.test_emu_head import test_head_fwd_bwd as run_head_case
    run_head_case(16, (128, 128), 2, 520, 2, 2, "glu", True)


def test_module_reprepares_operator_when_the_layout_knobs_change():
    """STConvBlock caches its prepared operator; changing the tiled threshold / plane padding must invalidate that cache
    (otherwise the kernels would read a fragment-ordered buffer as a dense matrix)."""
    from tests.test_emu_model import _build
    fx, model, x, _ = _build("tiny_cheb_f32")
    model.eval()
    with torch.no_grad():
        a = model(x)
        prev = ops.set_gc_tiled_min_nodes(1)
        try:
            b = model(x)
            pad = ops.set_gc_ld_pad(64)
            try:
                prec = ops.set_gc_precision("bf16x3")
                try:
                    c = model(x)
                finally:
                    ops.set_gc_precision(prec)
            finally:
                ops.set_gc_ld_pad(pad)
        finally:
            ops.set_gc_tiled_min_nodes(prev)
        d = model(x)
    assert torch.equal(a, d)
    assert (a - b).abs().max() < 2e-5 and (a - c).abs().max() < 2e-4
