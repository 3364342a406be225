"""-m gpu: opt-in bf16x3 operator products of the slab-resident graph conv (forward kernel, stgcn_kernels_gcslab16.hip.h) on a real
MI355X at the full C2 block sizes: against the fp64 stage oracle with the north-star tolerances, and against the exact-fp32 kernels."""
import numpy as np
import pytest
import torch

from tests.helpers import real_gso

pytestmark = pytest.mark.gpu


@pytest.fixture
def bf16x3():
    from stgcn_amd import ops
    from tests.gpu_util import bind_hip
    bind_hip()
    prev = ops.set_slab_gc_precision("bf16x3")
    try:
        yield
    finally:
        ops.set_slab_gc_precision(prev)


@pytest.mark.parametrize("blk", [0, 1])
def test_c2_blocks_against_the_stage_oracle(bf16x3, blk):
    """Both C2 blocks (real METR-LA operator, bs 32, unit-variance inputs) against the fp64 stage oracle.  The mode does NOT keep the
    1e-4 parity bar of the fp32 configs, which is why it is opt-in: activations within 3e-4 abs (X_2 = T_2(L) X0 carries ~1e-5
    per element into the second conv and the LayerNorm), parameter gradients within 1e-2 relative; the element-wise gradients
    downstream of the ReLU are not compared (its mask flips where the perturbed forward crosses zero)."""
    from tests.gpu_util import run_block_case
    gso = real_gso("metr_la.cheb_sym_norm_lap")
    c_in, T = ((1, 12), (64, 8))[blk]
    errs = run_block_case(c_in, (64, 16, 64), 3, 3, "cheb_graph_conv", "glu", 207, 32, T, True, gso=gso)
    for k, v in errs.items():
        if k.startswith("fwd.") and not k.endswith("bitwise"):
            assert v <= 3e-4, (k, v, errs)
        elif k.startswith("grad."):
            assert v <= 1e-2, (k, v, errs)
        elif k.startswith("grad_none_ok") or k.endswith("bitwise"):
            assert v == 0.0, (k, v, errs)


def test_forward_tracks_the_fp32_kernels():
    from stgcn_amd import ops
    from tests.emu_util import block_case, params_in_field_order
    from tests.gpu_util import bind_hip
    bind_hip()
    dev = "cuda:0"
    N, B, T = 207, 32, 8
    _, p = block_case(64, (64, 16, 64), 3, 3, "cheb_graph_conv", "glu", N, B, T)
    bcfg = ops.BlockConfig(Kt=3, Ks=3, n_vertex=N, c_in=64, channels=(64, 16, 64), act_func="glu", graph_conv_type="cheb_graph_conv", droprate=0.5)
    gp, gt = ops.gso_prepare(torch.from_numpy(real_gso("metr_la.cheb_sym_norm_lap")).to(dev), 3)
    x = torch.from_numpy(np.random.RandomState(3).standard_normal((B, 64, T, N)).astype(np.float32)).to(dev)
    params = [None if t is None else t.clone().to(dev) for t in params_in_field_order(p, "st_blocks.0.", "cheb_graph_conv")]
    outs = {}
    for mode in ("fp32", "bf16x3"):
        prev = ops.set_slab_gc_precision(mode)
        try:
            outs[mode] = ops.st_conv_block(x, gp, gt, bcfg, params, True, 7, 1, ops.WorkspaceCache()).cpu().numpy()
        finally:
            ops.set_slab_gc_precision(prev)
    d = np.abs(outs["fp32"] - outs["bf16x3"]).max()
    assert 0 < d < 3e-4, d
