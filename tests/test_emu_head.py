"""Fused output head (OutputBlock) on the CPU emulator vs the autograd oracle (float64), with the library's
own dropout mask."""
import numpy as np
import pytest
import torch

from oracle import stgcn_oracle as orc
from stgcn_amd import _lib, ops
from tests.emu_util import bind_emulator

CASES = [
    # c_in, (c0, c1), Ko, N, B, T, act, training
    (64, (128, 128), 4, 21, 2, 4, "glu", True),
    (64, (128, 128), 3, 9, 1, 5, "gtu", False),      # T1 = 3 > 1
    (16, (64, 128), 2, 33, 2, 2, "glu", True),
    # T1 = 1, 128 channels, N >= 32: the ONE-launch forward (head_fwd_kernel): 4 row tiles of 32 over 3 windows of 40 nodes (tiles that hold
    # one window / straddle two, a ragged last tile); the emulator re-runs the tiles that had to wait for their windows' later tiles
    (64, (128, 128), 4, 40, 3, 4, "glu", True),
    (64, (128, 128), 4, 70, 2, 4, "gtu", False),
]


@pytest.mark.parametrize("c_in,channels,Ko,N,B,T,act,training", CASES)
def test_head_fwd_bwd(c_in, channels, Ko, N, B, T, act, training, dev="cpu"):
    if dev == "cpu":
        bind_emulator()
    T1 = T - Ko + 1
    n_his = Ko          # makes cfg.Ko == Ko with zero ST blocks
    cfg = orc.OracleConfig(Kt=3, Ks=3, n_his=n_his, act_func=act, droprate=0.5, blocks=[[c_in], list(channels), [1]])
    assert cfg.n_st_blocks == 0 and cfg.Ko == Ko
    p = {k: v for k, v in orc.random_params(cfg, N, seed=5, dtype=torch.float32).items() if k.startswith("output.")}
    rs = np.random.RandomState(2)
    x_np = rs.standard_normal((B, c_in, T, N)).astype(np.float32)
    dout_np = rs.standard_normal((B, 1, T1, N)).astype(np.float32)
    hcfg = ops.HeadConfig(Ko=Ko, n_vertex=N, c_in=c_in, channels=tuple(channels), end_channel=1, act_func=act, droprate=0.5)
    names = ["tmp_conv1.causal_conv.weight", "tmp_conv1.causal_conv.bias", "tmp_conv1.align.align_conv.weight",
             "tmp_conv1.align.align_conv.bias", "tc1_ln.weight", "tc1_ln.bias", "fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias"]
    params = [p["output." + n].clone().to(dev).requires_grad_(True) for n in names]
    x = torch.from_numpy(x_np).to(dev).requires_grad_(True)
    seed, offset = 77, 5
    out = ops.output_block(x, hcfg, params, training, seed, offset, ops.WorkspaceCache())
    assert out.shape == (B, 1, T1, N)
    out.backward(torch.from_numpy(dout_np).to(dev))

    keep = None
    if training:
        ks = ops.dropout_mask(B * T1 * N * channels[1], 0.5, seed, offset, dev).cpu().reshape(B, T1, N, channels[1])
        keep = (ks > 0).double()
    leaves = {k: v.double().clone().requires_grad_(True) for k, v in p.items()}
    xr = torch.from_numpy(x_np).double().requires_grad_(True)
    ref = orc.output_block(xr, leaves, "output.", cfg, Ko, c_in, channels, 1, keep)
    gr = torch.autograd.grad(ref, [xr] + [leaves["output." + n] for n in names], torch.from_numpy(dout_np).double(), allow_unused=True)
    assert (out.detach().cpu().double() - ref.detach()).abs().max() < 5e-5
    rel = lambda a, b: float((a.detach().cpu().double() - b).abs().max() / max(1e-30, float(b.abs().max())))
    assert rel(x.grad, gr[0]) < 2e-4, "dx"
    for n, prm, g in zip(names, params, gr[1:]):
        if g is None:
            assert prm.grad is None, n
        else:
            assert prm.grad is not None and rel(prm.grad, g) < 2e-4, (n, rel(prm.grad, g))


def test_head_forward_fused_tile_heights(monkeypatch):
    """The one-launch forward with 64-row tiles (a head whose 32-row tiles do not fit one resident round), with 32-row tiles handed out by
    start-order ticket (heads beyond that: the emulator re-runs a waiting tile under the ticket it drew), and switched off."""
    for mode in ("4", "2", "3", "5", "0"):      # ("3" / "5": 32- / 64-row tiles by start-order ticket -- the default picks one of them since round 5; "2" / "4": tiles by blockIdx, the whole grid resident)
        monkeypatch.setenv("STGCN_HEAD_FUSE", mode)
        test_head_fwd_bwd(64, (128, 128), 4, 70, 2, 4, "glu", True)
        test_head_fwd_bwd(64, (128, 128), 4, 40, 3, 4, "glu", True)


def test_head_tile_height_variants(monkeypatch):
    """The two-launch head with the other tile heights its kernels are instantiated for: 32-row tiles of the fc kernels (what heads of more
    than 32 768 rows take in the forward: fc_fwd_kernel<2>) and 16-row tiles of the conv kernels."""
    monkeypatch.setenv("STGCN_HEAD_FUSE", "0")
    monkeypatch.setenv("STGCN_HEAD_FC_TILE", "32")
    test_head_fwd_bwd(64, (128, 128), 4, 40, 3, 4, "glu", True)
    test_head_fwd_bwd(64, (128, 128), 4, 21, 2, 4, "glu", True)
    from tests.bf16_util import assert_bf16_errors, run_head_case_bf16
    assert_bf16_errors(*run_head_case_bf16("cpu", 45, 3, training=True))


def head_wait_give_up_case(dev):
    """The bounded wait of the one-launch forward: with the bound set to "give up at once" a tile that finds its window's
    counter short writes NaN predictions and sets the sticky word; the NEXT forward (bound restored) is clean.  Shared by the emulator test
    and tests/test_gpu_model.py."""
    N, B, Ko, c_in, channels = 70, 4, 4, 64, (128, 128)
    cfg = orc.OracleConfig(Kt=3, Ks=3, n_his=Ko, act_func="glu", droprate=0.5, blocks=[[c_in], list(channels), [1]])
    p = {k: v for k, v in orc.random_params(cfg, N, seed=5, dtype=torch.float32).items() if k.startswith("output.")}
    names = ["tmp_conv1.causal_conv.weight", "tmp_conv1.causal_conv.bias", "tmp_conv1.align.align_conv.weight",
             "tmp_conv1.align.align_conv.bias", "tc1_ln.weight", "tc1_ln.bias", "fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias"]
    params = [p["output." + n].clone().to(dev) for n in names]
    x = torch.from_numpy(np.random.RandomState(2).standard_normal((B, c_in, Ko, N)).astype(np.float32)).to(dev)
    hcfg = ops.HeadConfig(Ko=Ko, n_vertex=N, c_in=c_in, channels=channels, end_channel=1, act_func="glu", droprate=0.5)
    wsc = ops.WorkspaceCache()
    good = ops.output_block(x, hcfg, params, False, 1, 0, wsc)
    assert ops.head_chain_status(hcfg, B, Ko, wsc) == 0 and bool(torch.isfinite(good).all())
    prev = ops.set_chain_spin_ticks(-2000)      # test setting: waits bounded by 20 us, the first tile withholds its arrival
    try:
        assert ops.set_chain_spin_ticks(0) == -2000
        bad = ops.output_block(x, hcfg, params, False, 1, 0, wsc)
        word = ops.head_chain_status(hcfg, B, Ko, wsc)
    finally:
        ops.set_chain_spin_ticks(prev)
    assert word == 1, word                                     # 1 + the counter index (window 0) the starved tiles gave up on
    # window 0's tiles are rows 0 .. 95 (32-row tiles 0, 1 and 2; tile 2 straddles windows 0 / 1): all of THEIR predictions are NaN,
    # every other tile found its windows complete and is untouched
    nan = torch.isnan(bad).flatten()
    assert bool(nan[:96].all()) and not bool(nan[96:].any()), nan.nonzero().flatten().tolist()
    assert torch.equal(bad.flatten()[96:], good.flatten()[96:])
    again = ops.output_block(x, hcfg, params, False, 1, 0, wsc)
    assert ops.head_chain_status(hcfg, B, Ko, wsc) == 0 and torch.equal(again, good)


def test_head_forward_wait_give_up_is_loud_and_not_sticky_across_launches():
    bind_emulator()
    head_wait_give_up_case("cpu")
