"""torch.ops.stgcn.stblock_fwd / stblock_bwd on the MI355X (CUDA dispatch key = the HIP library): equal to the module path, fp32 and bf16."""
import numpy as np
import pytest
import torch

from stgcn_amd import _lib, ops, torch_ops
from tests.emu_util import block_case, nonsym_gso, params_in_field_order
from tests.gpu_util import bind_hip

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_operators_equal_the_module_path_on_gpu(dtype):
    bind_hip()
    dev = "cuda:0"
    c_in, Kt, Ks, gct, act, N, B, T = 64, 3, 3, "cheb_graph_conv", "glu", 207, 4, 12
    channels = (64, 16, 64)
    _, p = block_case(c_in, channels, Kt, Ks, gct, act, N, B, T)
    bcfg = ops.BlockConfig(Kt=Kt, Ks=Ks, n_vertex=N, c_in=c_in, channels=channels, act_func=act, graph_conv_type=gct, droprate=0.5)
    gp, gt = ops.gso_prepare(torch.from_numpy(nonsym_gso(N, 5)).to(dev), ops.graph_terms(bcfg))
    rs = np.random.RandomState(7)
    x = torch.from_numpy(rs.standard_normal((B, c_in, T, N)).astype(np.float32)).to(dev).to(dtype)
    dy = torch.from_numpy(rs.standard_normal((B, channels[2], T - 2 * (Kt - 1), N)).astype(np.float32)).to(dev).to(dtype)
    plist = params_in_field_order(p, "st_blocks.0.", gct)
    pm = [None if t is None else t.clone().to(dev).requires_grad_(True) for t in plist]
    xm = x.clone().requires_grad_(True)
    ym = ops.st_conv_block(xm, gp, gt, bcfg, pm, True, 99, 3, ops.WorkspaceCache())
    ym.backward(dy)
    ops.clear_ln_hooks()
    empty = torch.empty(0, device=dev)
    po = [empty if t is None else t.clone().to(dev).requires_grad_(True) for t in plist]
    xo = x.permute(0, 2, 3, 1).contiguous().requires_grad_(True)
    yo = torch_ops.stblock(xo, gp, gt, po, [c_in, 64, 16, 64, Kt, Ks, N], act, gct, 0.5, True, 99, 3)
    yo.backward(dy.permute(0, 2, 3, 1).contiguous())
    torch.cuda.synchronize()
    assert torch.equal(yo, ym.permute(0, 2, 3, 1))
    assert torch.equal(xo.grad, xm.grad.permute(0, 2, 3, 1))
    for name, a, b in zip(_lib.PARAM_FIELDS, pm, po):
        if a is not None and a.grad is not None:
            assert torch.equal(a.grad, b.grad), name
