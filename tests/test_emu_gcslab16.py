"""CPU (emulator): slab-resident graph conv with the operator products on the bf16 matrix cores ("bf16x3",
stgcn_kernels_gcslab16.hip.h) against the exact-fp32 slab kernels -- same block, same inputs, same dropout stream."""
import numpy as np
import pytest
import torch

from stgcn_amd import ops
from tests.emu_util import bind_emulator, block_case, nonsym_gso, params_in_field_order
from tests.test_emu_gctile import _block_inputs, _rel, _run


@pytest.fixture
def slab_precision():
    bind_emulator()
    prev = {}

    def use(mode):
        prev.setdefault("mode", ops.set_slab_gc_precision(mode))
    try:
        yield use
    finally:
        if "mode" in prev:
            ops.set_slab_gc_precision(prev["mode"])


def test_bf16_fragment_planes_of_the_operator(slab_precision):
    """stgcn_gso_prepare: behind the fp32 fragments, hi / lo bf16 planes in the fragment order of the 32-deep MFMA; hi + lo carries 16
    bits of mantissa of every entry of T_k (and of T_k^T in the second buffer)."""
    N, terms = 45, 3
    gso = nonsym_gso(N, 9)
    gp, gt = ops.gso_prepare(torch.from_numpy(gso), terms)
    NP, NP32 = 48, 64
    HT, KC32 = NP // 16, NP32 // 32
    per = HT * KC32 * 512                                         # bf16 elements per plane
    assert gp.shape[0] == (terms - 1) + -(-(terms - 1) * (2 * per // 2) // (NP * NP))
    T = [np.eye(N), gso.astype(np.float64)]
    T.append(2 * T[1] @ T[1] - T[0])
    for buf, tr in ((gp, False), (gt, True)):
        flat = buf.numpy().reshape(-1)[(terms - 1) * NP * NP:].view(np.uint16)
        for k in (1, 2):
            ref = np.zeros((NP, NP32))
            ref[:N, :N] = T[k].T if tr else T[k]
            hi = (flat[(k - 1) * 2 * per:(k - 1) * 2 * per + per].astype(np.uint32) << 16).view(np.float32)
            lo = (flat[(k - 1) * 2 * per + per:(k - 1) * 2 * per + 2 * per].astype(np.uint32) << 16).view(np.float32)
            e = np.arange(per)
            j, lane, rest = e & 7, (e >> 3) & 63, e >> 9
            kc, ht = rest % KC32, rest // KC32
            want = ref[ht * 16 + (lane & 15), kc * 32 + 8 * (lane >> 4) + j]
            assert np.abs(hi - want).max() <= 2.0 ** -8 * np.abs(want).max()
            assert np.abs(hi.astype(np.float64) + lo - want).max() <= 2.0 ** -16 * np.abs(want).max()


# 45 nodes: three node tiles, the last ragged, two 32-node chunks; 130 nodes: 9 tiles over 3 parts; Kipf conv: one operator term
@pytest.mark.parametrize("gct,Ks,N,B,T", [("cheb_graph_conv", 3, 45, 2, 6), ("cheb_graph_conv", 4, 130, 1, 5), ("graph_conv", 1, 45, 2, 6)])
def test_bf16x3_slab_products_track_fp32(slab_precision, gct, Ks, N, B, T):
    """~2^-17 relative per product.  On an operator with row sums up to 6 (worse than any rescaled Laplacian: T_2 has entries up to 70)
    the block output stays within 1.5e-4 abs of the exact-fp32 kernels (1e-3 after three recursion steps), input and parameter
    gradients within the 1e-3 relative bar of the fp32 configs; the real METR-LA operator is the next test."""
    args = _block_inputs(gct, Ks, N, B, T)
    slab_precision("fp32")
    ya, dxa, ga = _run(*args)
    slab_precision("bf16x3")
    yb, dxb, gb = _run(*args)
    assert 0 < np.abs(ya - yb).max() < (1e-3 if Ks > 3 else 1.5e-4)          # > 0: the bf16 kernels really ran
    assert _rel(dxa, dxb) < 1e-3
    for a, b in zip(ga, gb):
        assert (a is None) == (b is None)
        if a is not None:
            assert _rel(a, b) < 1e-3


def test_bf16x3_on_the_metr_la_operator(slab_precision):
    """The C2 operator (207 nodes, rescaled symmetric Laplacian, Ks = 3), unit-variance inputs: block output within 1.2e-4 abs of the
    exact-fp32 kernels (the maximum depends on which elements the dropout mask keeps) -- at the 1e-4 parity bar, which is why the mode is opt-in -- gradients within 1e-4 relative."""
    from tests.helpers import real_gso
    c_in, channels, Kt, Ks, gct, act, N, B, T = 64, (64, 16, 64), 3, 3, "cheb_graph_conv", "glu", 207, 1, 5
    _, p = block_case(c_in, channels, Kt, Ks, gct, act, N, B, T)
    gso = real_gso("metr_la.cheb_sym_norm_lap")
    rs = np.random.RandomState(4)
    x_np = rs.standard_normal((B, c_in, T, N)).astype(np.float32)
    dy_np = rs.standard_normal((B, channels[2], T - 2 * (Kt - 1), N)).astype(np.float32)
    args = (c_in, channels, Kt, Ks, gct, act, N, B, T, x_np, dy_np, p, gso)
    slab_precision("fp32")
    ya, dxa, ga = _run(*args)
    slab_precision("bf16x3")
    yb, dxb, gb = _run(*args)
    assert 0 < np.abs(ya - yb).max() < 1.2e-4
    assert _rel(dxa, dxb) < 1e-3
    for a, b in zip(ga, gb):
        if a is not None:
            assert _rel(a, b) < 1e-4
