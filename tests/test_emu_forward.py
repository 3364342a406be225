"""Forward kernels of the HIP library, executed on the CPU emulator (tests/emu), checked stage by stage
against the numpy stage oracle.  Not a GPU test: it validates index math / fragment layouts / barriers of
the *same sources* that hipcc compiles for gfx950."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import stblock_stages as st
from stgcn_amd import _lib, ops
from tests.emu_util import bind_emulator, block_case, nonsym_gso, params_in_field_order

CASES = [
    # c_in, channels, Kt, Ks, gct, act, N, B, T
    (1, (64, 16, 64), 3, 3, "cheb_graph_conv", "glu", 21, 2, 7),
    (64, (64, 16, 64), 3, 3, "cheb_graph_conv", "glu", 17, 1, 6),
    (64, (64, 16, 64), 3, 3, "graph_conv", "gtu", 35, 1, 5),
    (16, (128, 16, 64), 2, 5, "cheb_graph_conv", "glu", 9, 2, 5),
    (1, (64, 16, 64), 3, 1, "cheb_graph_conv", "glu", 16, 1, 5),
    (1, (64, 16, 64), 3, 2, "cheb_graph_conv", "glu", 300, 6, 12),   # >= 256 row tiles: exercises the 64-row tile path
    (2, (64, 16, 64), 2, 3, "cheb_graph_conv", "gtu", 19, 2, 5),     # thin first layer with K = Kt * c_in = 4 taps (two channels x two steps)
]


@pytest.mark.parametrize("c_in,channels,Kt,Ks,gct,act,N,B,T", CASES)
def test_block_forward_stages(c_in, channels, Kt, Ks, gct, act, N, B, T):
    L = bind_emulator()
    ops.set_debug_stages(True)      # the forward stores U2 / S2 only for the stage tests (tc2_bwd_kernel recomputes them)
    cfg, p = block_case(c_in, channels, Kt, Ks, gct, act, N, B, T)
    gso = nonsym_gso(N, 5)
    rs = np.random.RandomState(11)
    x = rs.standard_normal((B, T, N, c_in)).astype(np.float32)      # channels-last

    bcfg = ops.BlockConfig(Kt=Kt, Ks=Ks, n_vertex=N, c_in=c_in, channels=tuple(channels), act_func=act,
                           graph_conv_type=gct, droprate=0.5)
    desc = ops.make_desc(bcfg, B, T, training=True, need_dx=c_in > 1)
    plan = ops.query_plan(desc)
    gp, gt = ops.gso_prepare(torch.from_numpy(gso), ops.graph_terms(bcfg))
    NP = plan.NP
    KCH = NP // 16

    def unpack(f):      # fragment order -> dense (NP, NP): f[((ht*KCH + kc)*64 + lane)*4 + s] = M[ht*16 + lane%16][kc*16 + 4*(lane//16) + s]
        f = f.numpy().reshape(NP // 16, KCH, 4, 16, 4)          # ht, kc, lane//16, lane%16, s
        return f.transpose(0, 3, 1, 2, 4).reshape(NP, NP)
    # T_1 = gso exactly; T_k = 2 gso T_{k-1} - T_{k-2} (fp64 accumulation in the prepare kernel), transposes alongside
    terms = ops.graph_terms(bcfg)
    assert gp.shape[0] >= max(terms - 1, 1)      # the fp32 fragments of every term, then their bf16 planes (tests/test_emu_gcslab16.py)
    tm2, tm1 = np.eye(N), gso.astype(np.float64)
    for k in range(1, terms):
        if k >= 2:
            tm2, tm1 = tm1, 2.0 * gso.astype(np.float64) @ tm1 - tm2
        dp, dt = unpack(gp[k - 1]), unpack(gt[k - 1])
        if k == 1:
            assert np.array_equal(dp[:N, :N], gso) and np.array_equal(dt[:N, :N], gso.T)
        else:
            scale = max(1.0, np.abs(tm1).max())
            assert np.abs(dp[:N, :N] - tm1).max() <= 2e-6 * scale and np.abs(dt[:N, :N] - tm1.T).max() <= 2e-6 * scale
        assert dp[N:].sum() == 0 and dp[:, N:].sum() == 0 and dt[N:].sum() == 0 and dt[:, N:].sum() == 0

    params = params_in_field_order(p, "st_blocks.0.", gct)
    pst = ops._param_struct(_lib.StblockParams, params)
    xt = torch.from_numpy(x)
    y = torch.full((B, plan.T2, N, channels[2]), float("nan"))
    saved = torch.full((plan.saved_floats,), float("nan"))
    ws = torch.full((plan.ws_floats,), float("nan"))
    seed, offset = 1234, 7
    L.check(L.dll.stgcn_stblock_forward(C.byref(desc), C.byref(pst), xt.data_ptr(), gp.data_ptr(), y.data_ptr(),
                                        saved.data_ptr(), ws.data_ptr(), seed, offset, None, None), "fwd")

    # oracle with the library's own dropout mask
    keep_scale = ops.dropout_mask(y.numel(), 0.5, seed, offset, "cpu").numpy().reshape(y.shape)
    assert set(np.unique(keep_scale)) <= {0.0, 2.0}
    assert 0.35 < (keep_scale > 0).mean() < 0.65
    bp = st.block_params_np(p, "st_blocks.0.", gct, np.float64)
    y_ref, sv = st.stblock_fwd(x.astype(np.float64), gso.astype(np.float64), bp, Kt, c_in, channels, gct, act,
                               keep=(keep_scale > 0).astype(np.float64), p_drop=0.5)

    def seg(off, shape):
        n = int(np.prod(shape))
        return saved.numpy()[off:off + n].reshape(shape)

    T1, T2 = plan.T1, plan.T2
    c0, c1, c2 = channels
    tol = 2e-5
    if not plan.recompute_tc1:      # first-block gate inputs are recomputed in backward instead of stored
        assert np.abs(seg(plan.sv_U1, (B, T1, N, c0)) - sv["U1"]).max() < tol
        assert np.abs(seg(plan.sv_S1, (B, T1, N, c0)) - sv["S1"]).max() < tol
    assert np.abs(seg(plan.sv_A, (B, T1, N, c1)) - sv["A"]).max() < tol
    terms = 2 if gct == "graph_conv" else Ks
    for k in range(1, terms):
        got = seg(plan.sv_Xk + (k - 1) * B * T1 * N * c1, (B, T1, N, c1))
        assert np.abs(got - sv["Xs"][k]).max() < tol, f"X{k}"
    assert np.abs(seg(plan.sv_G, (B, T1, N, c1)) - sv["G"]).max() < tol
    assert np.abs(seg(plan.sv_U2, (B, T2, N, c2)) - sv["U2"]).max() < tol
    assert np.abs(seg(plan.sv_S2, (B, T2, N, c2)) - sv["S2"]).max() < tol
    assert np.abs(seg(plan.sv_mean, (B, T2)) - sv["mean"]).max() < tol
    assert np.abs(seg(plan.sv_rstd, (B, T2)) / sv["rstd"] - 1).max() < 1e-4
    assert np.isfinite(y.numpy()).all()
    assert np.abs(y.numpy() - y_ref).max() < 5e-5


@pytest.mark.parametrize("wgs", [1, 3, 5])
def test_tc1_fwd_ranges_cut_inside_items(wgs):
    """tc1_fwd_kernel hands every workgroup an equal range of the (window, node tile, output step) sequence; with fewer workgroups than
    items the cuts fall inside items (the Kt - 1 input tiles in front of a cut are read again, nothing is recomputed): same results."""
    from stgcn_amd import ops
    bind_emulator()
    prev = ops.set_tc1_bwd_wgs(wgs)
    try:
        test_block_forward_stages(64, (64, 16, 64), 3, 3, "cheb_graph_conv", "glu", 17, 2, 7)
        test_block_forward_stages(32, (64, 16, 64), 3, 1, "cheb_graph_conv", "gtu", 16, 1, 5)
    finally:
        ops.set_tc1_bwd_wgs(prev)


PEER_CASES = [
    (64, (64, 16, 64), 3, 3, "cheb_graph_conv", "glu", 130, 1, 6),     # 9 node tiles: 5 + 4 (PP = 2), 2 + 2 + 2 + 3 (PP = 4)
    (16, (64, 16, 64), 2, 2, "cheb_graph_conv", "gtu", 150, 2, 5),     # 10 tiles, ragged last tile, two taps, GTU
    (64, (64, 16, 64), 3, 2, "cheb_graph_conv", "glu", 300, 1, 5),     # 19 tiles: the instances of 257 .. 384 nodes (3 / 2 tiles per wave)
]


@pytest.mark.parametrize("pp", [1, 2, 4])
@pytest.mark.parametrize("case", PEER_CASES)
def test_tc2_ln_fwd_workgroups_per_slab(case, pp):
    """tc2_ln_fwd_kernel with PP workgroups per (b, t) slab (``stgcn_set_tc2ln_peers``): slab and part from a start-order ticket, the parts'
    LayerNorm statistics exchanged inside the launch and merged in part order -- every stage against the oracle, for every PP, and the
    same bits launch after launch (the merge order is fixed)."""
    bind_emulator()
    prev = ops.set_tc2ln_peers(pp)
    try:
        test_block_forward_stages(*case)
    finally:
        ops.set_tc2ln_peers(prev)


def tc2_ln_peer_give_up_case(dev):
    """The bounded wait of the slab's parts (the head's ``head_wait_give_up_case`` for the ST block): with the test setting of the spin bound
    the launch's first workgroup withholds its statistics word, the other parts of ITS slab give up, write NaN and set the sticky word
    (1 + slab index); every other slab is untouched; the next forward is clean.  Shared by the emulator test and tests/test_gpu_block.py."""
    N, B, T = 130, 2, 6
    cfg, p = block_case(64, (64, 16, 64), 3, 3, "cheb_graph_conv", "glu", N, B, T)
    bcfg = ops.BlockConfig(Kt=3, Ks=3, n_vertex=N, c_in=64, channels=(64, 16, 64), act_func="glu", graph_conv_type="cheb_graph_conv", droprate=0.5)
    gp, gt = ops.gso_prepare(torch.from_numpy(nonsym_gso(N, 5)).to(dev), ops.graph_terms(bcfg))
    params = [None if t is None else t.clone().to(dev) for t in params_in_field_order(p, "st_blocks.0.", "cheb_graph_conv")]
    x = torch.from_numpy(np.random.RandomState(3).standard_normal((B, 64, T, N)).astype(np.float32)).to(dev)
    wsc = ops.WorkspaceCache()
    prev_pp = ops.set_tc2ln_peers(2)
    try:
        good = ops.st_conv_block(x, gp, gt, bcfg, params, False, 1, 0, wsc)
        assert ops.block_chain_status(bcfg, B, T, wsc) == 0 and bool(torch.isfinite(good).all())
        prev = ops.set_chain_spin_ticks(-2000)      # test setting: waits bounded by 20 us, the first workgroup withholds its word
        try:
            bad = ops.st_conv_block(x, gp, gt, bcfg, params, False, 1, 0, wsc)
            word = ops.block_chain_status(bcfg, B, T, wsc)
        finally:
            ops.set_chain_spin_ticks(prev)
        assert word == 1, word                                  # 1 + slab 0
        # slab 0 = (b 0, t 0): part 0 (node tiles 0 .. 3, rows 0 .. 63) was normalised with its peer's real statistics, part 1 (rows 64 ..) gave up
        nan = torch.isnan(bad)                                  # logical (B, C, T2, N)
        assert bool(nan[0, :, 0, 64:].all()) and not bool(nan[0, :, 0, :64].any())
        nan[0, :, 0, 64:] = False
        assert not bool(nan.any())
        keep = torch.ones_like(bad, dtype=torch.bool)
        keep[0, :, 0, 64:] = False
        assert torch.equal(bad[keep], good[keep])
        again = ops.st_conv_block(x, gp, gt, bcfg, params, False, 1, 0, wsc)
        assert ops.block_chain_status(bcfg, B, T, wsc) == 0 and torch.equal(again, good)
    finally:
        ops.set_tc2ln_peers(prev_pp)


def test_tc2_ln_peer_wait_give_up_is_loud_and_not_sticky_across_launches():
    bind_emulator()
    tc2_ln_peer_give_up_case("cpu")

