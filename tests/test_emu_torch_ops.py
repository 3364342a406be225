"""The dispatcher form of the ST-Conv block (stgcn_amd/torch_ops.py: torch.ops.stgcn.stblock_fwd / stblock_bwd) on the
CPU emulator: registered with the schemas the module documents, bit-identical to ops.st_conv_block (the same two C entry points), and
differentiable through the wrapper."""
import numpy as np
import pytest
import torch

from stgcn_amd import _lib, ops, torch_ops
from tests.emu_util import bind_emulator, block_case, nonsym_gso, params_in_field_order


def _case(c_in, Kt, Ks, gct, act, N, B, T):
    bind_emulator()
    channels = (64, 16, 64)
    _, p = block_case(c_in, channels, Kt, Ks, gct, act, N, B, T)
    bcfg = ops.BlockConfig(Kt=Kt, Ks=Ks, n_vertex=N, c_in=c_in, channels=channels, act_func=act, graph_conv_type=gct, droprate=0.5)
    gp, gt = ops.gso_prepare(torch.from_numpy(nonsym_gso(N, 5)), ops.graph_terms(bcfg))
    rs = np.random.RandomState(7)
    x = torch.from_numpy(rs.standard_normal((B, c_in, T, N)).astype(np.float32))
    dy = torch.from_numpy(rs.standard_normal((B, channels[2], T - 2 * (Kt - 1), N)).astype(np.float32))
    plist = params_in_field_order(p, "st_blocks.0.", gct)
    return bcfg, gp, gt, x, dy, plist


def test_schemas_are_registered():
    s = str(torch.ops.stgcn.stblock_fwd.default._schema)
    assert s.startswith("stgcn::stblock_fwd(Tensor x_cl, Tensor gso_pad, Tensor[] params, int[] cfg, str act, str gc_type") and "-> (Tensor, Tensor, Tensor)" in s
    assert "-> Tensor[]" in str(torch.ops.stgcn.stblock_bwd.default._schema)


@pytest.mark.parametrize("c_in,Kt,Ks,gct,act,N,B,T,training", [(64, 3, 3, "cheb_graph_conv", "glu", 17, 2, 6, True),
                                                                (1, 3, 2, "graph_conv", "gtu", 21, 1, 7, False)])
def test_operators_equal_the_module_path(c_in, Kt, Ks, gct, act, N, B, T, training):
    bcfg, gp, gt, x, dy, plist = _case(c_in, Kt, Ks, gct, act, N, B, T)
    seed, offset = 99, 3
    # module path
    pm = [None if t is None else t.clone().requires_grad_(True) for t in plist]
    xm = x.clone().requires_grad_(c_in > 1)
    ym = ops.st_conv_block(xm, gp, gt, bcfg, pm, training, seed, offset, ops.WorkspaceCache())
    ym.backward(dy)
    ops.clear_ln_hooks()
    # operator path (channels-last in and out)
    empty = torch.empty(0)
    po = [empty if t is None else t.clone().requires_grad_(True) for t in plist]
    xo = x.permute(0, 2, 3, 1).contiguous().requires_grad_(c_in > 1)
    cfg = [c_in, 64, 16, 64, Kt, Ks, N]
    yo = torch_ops.stblock(xo, gp, gt, po, cfg, act, gct, 0.5, training, seed, offset)
    yo.backward(dy.permute(0, 2, 3, 1).contiguous())
    assert torch.equal(yo, ym.permute(0, 2, 3, 1))
    if c_in > 1:
        assert torch.equal(xo.grad, xm.grad.permute(0, 2, 3, 1))
    for name, a, b in zip(_lib.PARAM_FIELDS, pm, po):
        if a is None:
            continue
        if a.grad is None:
            assert b.grad is None, f"{name}: the module path leaves .grad None"
        else:
            assert b.grad is not None and torch.equal(a.grad, b.grad), name


def test_plain_operator_call_and_cpu_key_is_not_a_fallback():
    bcfg, gp, gt, x, dy, plist = _case(64, 3, 3, "cheb_graph_conv", "glu", 17, 1, 6)
    empty = torch.empty(0)
    params = [empty if t is None else t for t in plist]
    x_cl = x.permute(0, 2, 3, 1).contiguous()
    y, saved, ws = torch.ops.stgcn.stblock_fwd(x_cl, gp, params, [64, 64, 16, 64, 3, 3, 17], "glu", "cheb_graph_conv", 0.5, False, 0, 0)
    assert y.shape == (1, 2, 17, 64) and saved.dtype == torch.float32 and ws.numel() > 0
    out = torch.ops.stgcn.stblock_bwd(torch.ones_like(y), x_cl, gt, y, saved, ws, params, [64, 64, 16, 64, 3, 3, 17], "glu", "cheb_graph_conv",
                                      0.5, False, 0, 0, True)
    assert len(out) == 1 + len(_lib.PARAM_FIELDS) and out[0].shape == x_cl.shape
    # with the product library bound, host tensors are refused (no CPU fallback)
    import os
    if os.path.exists(_lib.DEFAULT_LIB):
        _lib.use_library(_lib.DEFAULT_LIB)
        try:
            with pytest.raises(RuntimeError, match="no CPU implementation"):
                torch.ops.stgcn.stblock_fwd(x_cl, gp, params, [64, 64, 16, 64, 3, 3, 17], "glu", "cheb_graph_conv", 0.5, False, 0, 0)
        finally:
            bind_emulator()


def test_head_operators_equal_the_module_path():
    from oracle import stgcn_oracle as orc
    bind_emulator()
    c_in, channels, Ko, N, B, T, act = 64, (128, 128), 4, 21, 2, 4, "glu"
    cfg = orc.OracleConfig(Kt=3, Ks=3, n_his=Ko, act_func=act, droprate=0.5, blocks=[[c_in], list(channels), [1]])
    p = {k: v for k, v in orc.random_params(cfg, N, seed=5, dtype=torch.float32).items() if k.startswith("output.")}
    names = ["tmp_conv1.causal_conv.weight", "tmp_conv1.causal_conv.bias", "tmp_conv1.align.align_conv.weight",
             "tmp_conv1.align.align_conv.bias", "tc1_ln.weight", "tc1_ln.bias", "fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias"]
    rs = np.random.RandomState(2)
    x = torch.from_numpy(rs.standard_normal((B, c_in, T, N)).astype(np.float32))
    dout = torch.from_numpy(rs.standard_normal((B, 1, T - Ko + 1, N)).astype(np.float32))
    hcfg = ops.HeadConfig(Ko=Ko, n_vertex=N, c_in=c_in, channels=channels, end_channel=1, act_func=act, droprate=0.5)
    pm = [p["output." + n].clone().requires_grad_(True) for n in names]
    xm = x.clone().requires_grad_(True)
    om = ops.output_block(xm, hcfg, pm, True, 77, 5, ops.WorkspaceCache())
    om.backward(dout)
    po = [p["output." + n].clone().requires_grad_(True) for n in names]
    xo = x.permute(0, 2, 3, 1).contiguous().requires_grad_(True)
    oo = torch_ops.outblock(xo, po, [c_in, channels[0], channels[1], 1, Ko, N], act, 0.5, True, 77, 5)
    oo.backward(dout[:, 0])
    assert torch.equal(oo, om[:, 0])
    assert torch.equal(xo.grad, xm.grad.permute(0, 2, 3, 1))
    for n, a, b in zip(names, pm, po):
        if a.grad is None:
            assert b.grad is None, n
        else:
            assert torch.equal(a.grad, b.grad), n
    assert "-> (Tensor, Tensor, Tensor)" in str(torch.ops.stgcn.outblock_fwd.default._schema)


def test_operator_arguments_are_validated_and_fake_impls_give_shapes():
    """the dispatcher operators hand raw addresses to the C ABI, so wrong-sized / wrong-typed / non-contiguous buffers must be
    refused in Python; absent gradient slots come back as DISTINCT empty tensors; the fake (meta) implementations produce the shapes."""
    bcfg, gp, gt, x, dy, plist = _case(64, 3, 3, "cheb_graph_conv", "glu", 17, 1, 6)
    params = [torch.empty(0) if t is None else t for t in plist]
    x_cl = x.permute(0, 2, 3, 1).contiguous()
    cfg = [64, 64, 16, 64, 3, 3, 17]
    y, saved, ws = torch.ops.stgcn.stblock_fwd(x_cl, gp, params, cfg, "glu", "cheb_graph_conv", 0.5, False, 0, 0)
    args = lambda **kw: [kw.get("dy", torch.ones_like(y)), x_cl, kw.get("gt", gt), kw.get("y", y), kw.get("saved", saved), kw.get("ws", ws), params,
                         cfg, "glu", "cheb_graph_conv", 0.5, False, 0, 0, True]
    for bad in (dict(saved=saved[:-1]), dict(ws=ws[: ws.numel() // 2]), dict(y=y[:, :1]), dict(saved=saved.double()), dict(gt=gt[..., :-1]),
                dict(dy=torch.ones(1, 2, 17, 32)), dict(ws=ws[::2])):
        with pytest.raises(ValueError):
            torch.ops.stgcn.stblock_bwd(*args(**bad))
    with pytest.raises(ValueError):
        torch.ops.stgcn.stblock_fwd(x_cl, gp[..., :-1], params, cfg, "glu", "cheb_graph_conv", 0.5, False, 0, 0)
    out = torch.ops.stgcn.stblock_bwd(*args())
    empties = [t for t in out if t.numel() == 0]
    assert len(empties) >= 2 and len({id(t) for t in empties}) == len(empties)
    # fake tensors: shapes without running anything
    from torch._subclasses.fake_tensor import FakeTensorMode
    with FakeTensorMode() as mode:
        fx = mode.from_tensor(x_cl)
        fy, fs, fw = torch.ops.stgcn.stblock_fwd(fx, mode.from_tensor(gp), [mode.from_tensor(p) for p in params], cfg, "glu", "cheb_graph_conv", 0.5,
                                                 False, 0, 0)
        assert tuple(fy.shape) == tuple(y.shape) and fs.numel() == saved.numel() and fw.numel() == ws.numel()
