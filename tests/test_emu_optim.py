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
