"""Pins oracle/stgcn_oracle.py against outputs of the reference itself (tests/golden/*.npz,
produced by tests/golden/make_golden.py from /root/reference).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import stgcn_oracle as orc
from tests.helpers import cfg_from_fixture, fixture_gso, fixture_params, load_fixture, maxabs

CASES = ["tiny_cheb_f32", "tiny_cheb_f64", "tiny_gc_f32", "tiny_odd_f32", "tiny_ks1_f32", "tiny_ks5_f32",
         "metrla_c2_f32", "pemsd7m_c1_f32", "big600_ks4_f32", "metrla_c2_b32_f32", "pemsbay_c3_b64_f32"]


def _setup(name):
    fx = load_fixture(name)
    cfg = cfg_from_fixture(fx)
    dt = torch.float64 if name.endswith("f64") else torch.float32
    gso = torch.from_numpy(fixture_gso(name, fx)).to(dt)
    p = fixture_params(fx, cfg, dt)
    x = torch.from_numpy(fx["x"] if "x" in fx else None) if "x" in fx else None
    return fx, cfg, dt, gso, p


def _xy(fx, cfg, dt):
    rs = np.random.RandomState(int(fx["seed"]) + 1)
    B, N = int(fx["B"]), int(fx["n_vertex"])
    x = rs.standard_normal((B, 1, cfg.n_his, N))
    y = rs.standard_normal((B, N))
    return torch.from_numpy(x).to(dt), torch.from_numpy(y).to(dt)


@pytest.mark.parametrize("name", CASES)
def test_eval_forward_matches_reference(name):
    fx, cfg, dt, gso, p = _setup(name)
    x, _ = _xy(fx, cfg, dt)
    out, blocks = orc.stgcn_forward(x, gso, p, cfg, None, return_block_outputs=True)
    tol = 1e-12 if dt == torch.float64 else 1e-5
    assert out.shape == fx["eval.out"].shape
    assert maxabs(out.numpy(), fx["eval.out"]) <= tol
    for l, b in enumerate(blocks):
        if f"act.st_blocks.{l}" in fx:      # (not stored for the 600-node fixture)
            assert maxabs(b.numpy(), fx[f"act.st_blocks.{l}"]) <= tol * 3


def test_sublayer_activations_match_reference():
    fx, cfg, dt, gso, p = _setup("tiny_cheb_f32")
    x, _ = _xy(fx, cfg, dt)
    h = x
    for l in range(cfg.n_st_blocks):
        pre = f"st_blocks.{l}."
        ch = cfg.blocks[l + 1]
        t1 = orc.temporal_conv(h, p, pre + "tmp_conv1.", cfg.Kt, cfg.blocks[l][-1], ch[0], cfg.act_func)
        assert maxabs(t1.numpy(), fx[f"act.st_blocks.{l}.tmp_conv1"]) <= 5e-6
        g = orc.graph_conv_layer(t1, gso, p, pre + "graph_conv.", cfg.graph_conv_type, ch[0], ch[1])
        assert maxabs(g.numpy(), fx[f"act.st_blocks.{l}.graph_conv"]) <= 5e-6
        t2 = orc.temporal_conv(torch.relu(g), p, pre + "tmp_conv2.", cfg.Kt, ch[1], ch[2], cfg.act_func)
        assert maxabs(t2.numpy(), fx[f"act.st_blocks.{l}.tmp_conv2"]) <= 5e-6
        h = orc.st_conv_block(h, gso, p, pre, cfg, cfg.blocks[l][-1], ch, None)
        assert maxabs(h.numpy(), fx[f"act.st_blocks.{l}"]) <= 1e-5


@pytest.mark.parametrize("name", CASES)
def test_loss_and_grads_match_reference(name):
    fx, cfg, dt, gso, p = _setup(name)
    x, y = _xy(fx, cfg, dt)
    loss, grads = orc.loss_and_grads(x, y, gso, p, cfg, None)
    rtol = 1e-10 if dt == torch.float64 else 2e-5
    assert abs(float(loss) - float(fx["train.loss"])) <= rtol * abs(float(fx["train.loss"]))
    nograd = set(str(s) for s in fx["nograd"])
    for k, g in grads.items():
        if k in nograd:
            assert g is None, f"{k}: reference leaves .grad None"
            continue
        assert g is not None, k
        ref_sum = fx["gradsum." + k]
        gs = float(g.double().sum()); ga = float(g.double().abs().sum())
        assert abs(ga - ref_sum[1]) <= 1e-4 * ref_sum[1] + 1e-12, k
        assert abs(gs - ref_sum[0]) <= 1e-4 * ref_sum[1] + 1e-12, k
        if ("grad." + k) in fx:
            ref = fx["grad." + k]
            scale = max(1e-30, float(np.abs(ref).max()))
            assert maxabs(g.numpy(), ref) <= ((1e-10 * scale) if dt == torch.float64 else (1e-4 * scale + 5e-7)), k


def test_nograd_set_is_the_unused_align_convs():
    fx, cfg, dt, gso, p = _setup("tiny_cheb_f32")
    nograd = sorted(str(s) for s in fx["nograd"])
    # 10 tensors (align convs with c_in <= c_out) never receive a gradient
    assert len(nograd) == 10
    assert all("align.align_conv" in k for k in nograd)
    assert not any("graph_conv.align" in k for k in nograd)


def test_adamw_trajectory_matches_reference():
    fx, cfg, dt, gso, p = _setup("tiny_cheb_f32")
    x, y = _xy(fx, cfg, dt)
    state = {}
    losses = []
    for _ in range(len(fx["steps.losses"])):
        loss, _ = orc.train_step(x, y, gso, p, cfg, state)
        losses.append(float(loss))
    assert np.allclose(losses, fx["steps.losses"], rtol=2e-5)
    for k, v in p.items():
        ref = fx["steps.paramsum." + k]
        assert abs(float(v.double().abs().sum()) - ref[1]) <= 2e-5 * ref[1] + 1e-9, k
        if ("steps.param." + k) in fx:
            assert maxabs(v.numpy(), fx["steps.param." + k]) <= 2e-5, k


def test_param_count_c2():
    cfg = orc.OracleConfig()
    shapes = orc.param_shapes(cfg, 207)
    n = sum(int(np.prod(s)) for s in shapes.values())
    assert len(shapes) == 38 and n == 244609
    assert sum(int(np.prod(s)) for s in orc.param_shapes(cfg, 325).values()) == 305025
    cfg_gc = orc.OracleConfig(graph_conv_type="graph_conv")
    assert sum(int(np.prod(s)) for s in orc.param_shapes(cfg_gc, 228).values()) == 254337


def test_error_conventions():
    cfg = orc.OracleConfig(act_func="foo")
    with pytest.raises((NotImplementedError, KeyError)):
        p = orc.random_params(orc.OracleConfig(), 5)
        orc.temporal_conv(torch.zeros(1, 1, 12, 5), p, "st_blocks.0.tmp_conv1.", 3, 1, 64, "foo")
