"""stgcn_amd.optim.AdamW (one HIP kernel) on the emulator vs torch.optim.AdamW and vs the reference trajectory."""
import numpy as np
import torch

from stgcn_amd.optim import AdamW
from tests.emu_util import bind_emulator


def test_adamw_matches_torch_and_skips_gradless():
    bind_emulator()
    g = torch.Generator().manual_seed(0)
    shapes = [(64, 1, 3, 1), (128,), (3, 16, 16), (207, 64), (5000,), (1,)]
    ps1 = [torch.nn.Parameter(torch.randn(s, generator=g)) for s in shapes]
    ps2 = [torch.nn.Parameter(p.detach().clone()) for p in ps1]
    frozen1, frozen2 = torch.nn.Parameter(torch.randn(7, generator=g)), None
    frozen2 = torch.nn.Parameter(frozen1.detach().clone())
    o1 = AdamW(ps1 + [frozen1], lr=1e-3, weight_decay=1e-3)
    o2 = torch.optim.AdamW(ps2 + [frozen2], lr=1e-3, weight_decay=1e-3)
    sched1 = torch.optim.lr_scheduler.StepLR(o1, step_size=2, gamma=0.95)      # main.py:156
    sched2 = torch.optim.lr_scheduler.StepLR(o2, step_size=2, gamma=0.95)
    for it in range(5):
        for a, b in zip(ps1, ps2):
            gr = torch.randn(a.shape, generator=g) * (10.0 ** (it - 2))
            a.grad, b.grad = gr.clone(), gr.clone()
        o1.step(); o2.step(); sched1.step(); sched2.step()
    for a, b in zip(ps1, ps2):
        assert (a - b).abs().max() <= 2e-6 * max(1.0, b.abs().max().item())
    assert torch.equal(frozen1, frozen2)            # grad None -> untouched (no decay either)
    assert o1.param_groups[0]["lr"] == o2.param_groups[0]["lr"]


def test_training_trajectory_with_fused_adamw_matches_reference_golden():
    import types
    from stgcn_amd import models
    from stgcn_amd.train import make_optimizer, train_step
    from tests.helpers import cfg_from_fixture, fixture_gso, fixture_params, load_fixture, maxabs
    bind_emulator()
    fx = load_fixture("tiny_cheb_f32")
    cfg = cfg_from_fixture(fx)
    args = types.SimpleNamespace(Kt=cfg.Kt, Ks=cfg.Ks, act_func=cfg.act_func, graph_conv_type=cfg.graph_conv_type,
                                 gso=torch.from_numpy(fixture_gso("tiny_cheb_f32", fx)), enable_bias=True, droprate=0.0, n_his=cfg.n_his)
    model = models.STGCNChebGraphConv(args, cfg.blocks, int(fx["n_vertex"]))
    model.load_state_dict(fixture_params(fx, cfg, torch.float32), strict=True)
    rs = np.random.RandomState(int(fx["seed"]) + 1)
    B, N = int(fx["B"]), int(fx["n_vertex"])
    x = torch.from_numpy(rs.standard_normal((B, 1, cfg.n_his, N))).float()
    y = torch.from_numpy(rs.standard_normal((B, N))).float()
    opt = make_optimizer(model)
    model.train()
    losses = [float(train_step(model, opt, x, y)) for _ in range(len(fx["steps.losses"]))]
    assert np.allclose(losses, fx["steps.losses"], rtol=1e-4)
    for k, v in model.state_dict().items():
        if ("steps.param." + k) in fx:
            assert maxabs(v.numpy(), fx["steps.param." + k]) <= 1e-4, k


def test_mse_loss_and_grad_matches_torch():
    """stgcn_mse_loss_grad = nn.MSELoss() (main.py:136) + the gradient l.backward() feeds into the model output."""
    from stgcn_amd import ops
    from tests.emu_util import bind_emulator
    bind_emulator()
    g = torch.Generator().manual_seed(3)
    for shape in ((8, 207), (3, 5), (32, 325)):
        pred = torch.randn(*shape, generator=g)
        y = torch.randn(*shape, generator=g)
        loss, dpred = ops.mse_loss_and_grad(pred, y)
        p = pred.clone().requires_grad_(True)
        ref = torch.nn.MSELoss()(p, y)
        ref.backward()
        assert abs(float(loss[0]) - float(ref)) <= 1e-6 * abs(float(ref))
        assert float((dpred - p.grad).abs().max()) <= 1e-7 * float(p.grad.abs().max()) + 1e-12


def test_fused_step_tail_equals_plain_step():
    """train.fused_train_step (deferred reductions flushed by ONE stgcn_grad_flush launch with AdamW applied in place, gradients
    in a flat GradArena) is bitwise the plain step (per-module reduce launches + stgcn_adamw_step) -- same partials, same
    summation order, same optimizer arithmetic."""
    import types
    from stgcn_amd import models
    from stgcn_amd.train import GradArena, fused_train_step, make_optimizer, train_step
    from tests.emu_util import bind_emulator
    from tests.helpers import cfg_from_fixture, fixture_gso, fixture_params, load_fixture
    bind_emulator()
    fx = load_fixture("tiny_cheb_f32")
    cfg = cfg_from_fixture(fx)
    N, B = int(fx["n_vertex"]), int(fx["B"])

    def build():
        args = types.SimpleNamespace(Kt=cfg.Kt, Ks=cfg.Ks, act_func=cfg.act_func, graph_conv_type=cfg.graph_conv_type,
                                     gso=torch.from_numpy(fixture_gso("tiny_cheb_f32", fx)), enable_bias=True, droprate=cfg.droprate, n_his=cfg.n_his)
        m = models.STGCNChebGraphConv(args, cfg.blocks, N)
        m.load_state_dict(fixture_params(fx, cfg, torch.float32), strict=True)
        m.train()
        return m, make_optimizer(m, lr=1e-2, weight_decay=1e-2)

    g = torch.Generator().manual_seed(5)
    xs = torch.randn(4, B, 1, cfg.n_his, N, generator=g)
    ys = torch.randn(4, B, N, generator=g)
    m1, o1 = build()
    m2, o2 = build()
    l1 = [float(train_step(m1, o1, xs[i], ys[i])) for i in range(4)]
    l2 = [float(train_step(m2, o2, xs[0], ys[0]))]
    arena = GradArena([p for p in m2.parameters() if p.grad is not None])
    assert len(arena.params) == len([p for p in m1.parameters() if p.grad is not None])
    l2 += [float(fused_train_step(m2, o2, xs[i], ys[i], arena)) for i in range(1, 4)]
    assert l1 == l2, (l1, l2)
    for (k, a), b in zip(m1.state_dict().items(), m2.state_dict().values()):
        assert torch.equal(a, b), k
    for p1, p2 in zip(m1.parameters(), m2.parameters()):
        assert (p1.grad is None) == (p2.grad is None)
        if p1.grad is not None:
            assert torch.equal(p1.grad, p2.grad)
            assert p2.grad.data_ptr() == arena.grads[p2].data_ptr()     # gradients live in the flat arena
    # reduce-only flush (data-parallel path): gradients identical, parameters untouched until optimizer.step()
    before = [p.detach().clone() for p in m2.parameters()]
    from stgcn_amd import ops
    arena.install()
    with ops.grad_sink_scope(arena.sink):
        pred = m2(xs[0]).reshape(B, -1)
        _, dpred = ops.mse_loss_and_grad(pred, ys[0])
        pred.backward(dpred)
    arena.sink.flush()
    for b, p in zip(before, m2.parameters()):
        assert torch.equal(b, p.detach())
    m1.zero_grad(set_to_none=True)
    from stgcn_amd.train import fwd_loss_bwd
    fwd_loss_bwd(m1, xs[0], ys[0])
    for p1, p2 in zip(m1.parameters(), m2.parameters()):
        if p1.grad is not None:
            assert torch.equal(p1.grad, p2.grad)


def test_fused_tail_only_for_models_owned_by_fused_operators():
    """a live parameter outside the fused operators (the nn.Linear pair of the Ko == 0 head, models.py:46-51, or any user
    module around the model) gets its gradient from autograd, which would accumulate into the never-zeroed arena views: such models
    must take the plain step."""
    import types
    import pytest
    from stgcn_amd import models
    from stgcn_amd.train import GradArena, fused_tail_supported, fused_train_step, make_optimizer, train_step
    from tests.emu_util import bind_emulator, nonsym_gso
    bind_emulator()
    N = 9
    args = types.SimpleNamespace(Kt=3, Ks=2, act_func="glu", graph_conv_type="cheb_graph_conv", gso=torch.from_numpy(nonsym_gso(N, 1)),
                                 enable_bias=True, droprate=0.0, n_his=12)

    class Scaled(torch.nn.Module):           # a user module around the drop-in model: its parameter is owned by autograd
        def __init__(self, base):
            super().__init__()
            self.base, self.scale = base, torch.nn.Parameter(torch.ones(1))

        def forward(self, x):
            return self.base(x) * self.scale

    torch.manual_seed(0)
    base = models.STGCNChebGraphConv(args, [[1], [64, 16, 64], [64, 16, 64], [128, 128], [1]], N)
    m = Scaled(base)
    x, y = torch.randn(2, 1, 12, N), torch.randn(2, N)
    opt = make_optimizer(m)
    train_step(m, opt, x, y)
    live = [p for p in m.parameters() if p.grad is not None]
    assert not fused_tail_supported(m, live)
    with pytest.raises(RuntimeError, match="fused_tail_supported"):
        fused_train_step(m, opt, x, y, GradArena(live))
    train_step(base, make_optimizer(base), x, y)
    assert fused_tail_supported(base, [p for p in base.parameters() if p.grad is not None])


def test_four_block_model_packs_in_several_launches():
    """Kt = 2, n_his = 12 allows 4-5 ST blocks; their pack jobs exceed one launch's table and spill into further launches."""
    import types
    from stgcn_amd import models
    from tests.emu_util import bind_emulator, nonsym_gso
    bind_emulator()
    N = 7
    args = types.SimpleNamespace(Kt=2, Ks=2, act_func="glu", graph_conv_type="cheb_graph_conv", gso=torch.from_numpy(nonsym_gso(N, 1)),
                                 enable_bias=True, droprate=0.0, n_his=12)
    blocks = [[1]] + [[64, 16, 64]] * 5 + [[128, 128], [1]]
    torch.manual_seed(0)
    m = models.STGCNChebGraphConv(args, blocks, N)
    assert len(m.st_blocks) == 5 and m.Ko == 2
    out = m(torch.randn(1, 1, 12, N))
    assert out.shape == (1, 1, 1, N) and torch.isfinite(out).all()
    out.sum().backward()
    assert all(torch.isfinite(p.grad).all() for p in m.parameters() if p.grad is not None)


def test_module_reuse_before_flush_is_refused():
    """a module used twice before GradSink.flush() would overwrite its own deferred partials."""
    import types
    import pytest
    from stgcn_amd import models, ops
    from stgcn_amd.train import GradArena, make_optimizer, train_step
    from tests.emu_util import bind_emulator, nonsym_gso
    bind_emulator()
    N = 8
    args = types.SimpleNamespace(Kt=3, Ks=2, act_func="glu", graph_conv_type="cheb_graph_conv", gso=torch.from_numpy(nonsym_gso(N, 1)),
                                 enable_bias=True, droprate=0.0, n_his=12)
    torch.manual_seed(0)
    m = models.STGCNChebGraphConv(args, [[1], [64, 16, 64], [64, 16, 64], [128, 128], [1]], N)
    x, y = torch.randn(2, 1, 12, N), torch.randn(2, N)
    train_step(m, make_optimizer(m), x, y)
    arena = GradArena([p for p in m.parameters() if p.grad is not None])
    arena.install()
    with ops.grad_sink_scope(arena.sink):
        m(x).sum().backward()
        with pytest.raises(RuntimeError, match="flush"):
            m(x)
    arena.sink.flush()
    m(x)        # fine again after the flush


def test_mse_loss_fused_into_the_head_backward(monkeypatch):
    """ops.mse_backward: for a prediction that comes straight out of the fused head the seed gradient is formed inside the fc backward
    kernel and the loss value comes out of the gradient reduction (stgcn_outblock_backward_loss) -- same loss and the same gradients as
    the one-launch loss kernel followed by pred.backward(dpred); any other prediction takes that two-call path."""
    import types
    from stgcn_amd import models, ops
    from tests.emu_util import bind_emulator
    from tests.helpers import cfg_from_fixture, fixture_gso, fixture_params, load_fixture
    bind_emulator()
    fx = load_fixture("tiny_cheb_f32")
    cfg = cfg_from_fixture(fx)
    N, B = int(fx["n_vertex"]), int(fx["B"])
    args = types.SimpleNamespace(Kt=cfg.Kt, Ks=cfg.Ks, act_func=cfg.act_func, graph_conv_type=cfg.graph_conv_type,
                                 gso=torch.from_numpy(fixture_gso("tiny_cheb_f32", fx)), enable_bias=True, droprate=0.0, n_his=cfg.n_his)
    m = models.STGCNChebGraphConv(args, cfg.blocks, N)
    m.load_state_dict(fixture_params(fx, cfg, torch.float32), strict=True)
    m.train()
    g = torch.Generator().manual_seed(11)
    x, y = torch.randn(B, 1, cfg.n_his, N, generator=g), torch.randn(B, N, generator=g)

    pred = m(x).reshape(B, -1)
    assert ops._head_node_of(pred) is not None                       # shape-only views between the head and the loss
    assert ops._head_node_of(pred * 1.0) is None
    loss_a, dpred = ops.mse_loss_and_grad(pred, y, grad_scale=0.5)
    pred.backward(dpred)
    ga = {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}

    calls = []
    real = ops._lib.lib().dll.stgcn_outblock_backward_loss
    monkeypatch.setattr(ops._lib.lib().dll, "stgcn_outblock_backward_loss", lambda *a: (calls.append(1), real(*a))[1])
    m.zero_grad(set_to_none=True)
    loss_b = ops.mse_backward(m(x).reshape(B, -1), y, grad_scale=0.5)
    assert calls == [1]
    assert abs(float(loss_a) - float(loss_b)) <= 1e-6 * abs(float(loss_a))
    for k, p in m.named_parameters():
        if p.grad is not None:
            assert float((p.grad - ga[k]).abs().max()) <= 1e-6 * float(ga[k].abs().max()) + 1e-12, k

    m.zero_grad(set_to_none=True)                                    # a prediction with arithmetic behind the head: the two-call path
    loss_c = ops.mse_backward(m(x).reshape(B, -1) * 1.0, y, grad_scale=0.5)
    assert calls == [1] and abs(float(loss_a) - float(loss_c)) <= 1e-6 * abs(float(loss_a))
