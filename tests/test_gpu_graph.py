"""-m gpu: the hipGraph-captured training step replays to exactly what eager launches of the same kernels give,
draws a fresh dropout mask on every replay (device-side counter) and actually trains."""
import types

import numpy as np
import pytest
import torch

from tests.helpers import real_gso

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BLOCKS = [[1], [64, 16, 64], [64, 16, 64], [128, 128], [1]]


def _make(droprate):
    from stgcn_amd import models
    from tests.gpu_util import bind_hip
    bind_hip()
    gso = torch.from_numpy(real_gso("metr_la.cheb_sym_norm_lap")).to(DEV)
    args = types.SimpleNamespace(Kt=3, Ks=3, act_func="glu", graph_conv_type="cheb_graph_conv", gso=gso, enable_bias=True,
                                 droprate=droprate, n_his=12)
    torch.manual_seed(1)
    return models.STGCNChebGraphConv(args, BLOCKS, 207).to(DEV)


def test_graph_replay_equals_eager_and_masks_change():
    from stgcn_amd import DropoutStream
    from stgcn_amd.train import GraphedTrainStep, make_optimizer, train_step
    g = torch.Generator().manual_seed(0)
    xs = torch.randn(6, 8, 1, 12, 207, generator=g).to(DEV)
    ys = torch.randn(6, 8, 207, generator=g).to(DEV)

    # graph path (output-block nn.Dropout uses torch's own graph-safe Philox; compare ST-block behaviour with p_out = 0:
    # a model-wide droprate applies to both, so use eval-free comparison on droprate 0 for exactness, 0.5 for mask checks)
    DropoutStream.use_device_counter(torch.device(DEV))
    DropoutStream.manual_seed(3)
    m1 = _make(0.0)
    init = {k: v.clone() for k, v in m1.state_dict().items()}
    o1 = make_optimizer(m1, capturable=True)
    gs = GraphedTrainStep(m1, o1, xs[0], ys[0], warmup=2)
    # GraphedTrainStep's constructor already trained on xs[0]: 2 warm-up steps + 1 verification replay;
    # replicate that eagerly on the twin
    DropoutStream.disable_device_counter()
    m2 = _make(0.0)
    o2 = make_optimizer(m2, capturable=True)
    for _ in range(3):
        train_step(m2, o2, xs[0], ys[0])
    l1 = [float(gs(xs[i], ys[i]).item()) for i in range(1, 6)]
    l2 = [float(train_step(m2, o2, xs[i], ys[i]).item()) for i in range(1, 6)]
    assert np.allclose(l1, l2, rtol=1e-6, atol=0), (l1, l2)
    sd1, sd2 = m1.state_dict(), m2.state_dict()
    unused = [k for k, p in m1.named_parameters() if p.grad is None]
    assert len(unused) == 10
    report = {k: (float((sd1[k] - sd2[k]).abs().max()), float((sd1[k] - init[k]).abs().max()), float((sd2[k] - init[k]).abs().max()))
              for k in sd1}
    for k in unused:      # never touched by either path (the reference skips grad-None parameters)
        assert report[k][1] == 0.0 and report[k][2] == 0.0, (k, report[k])
    bad = {k: v for k, v in report.items() if v[0] > 1e-6}
    assert not bad, bad
    exact = sum(1 for v in report.values() if v[0] == 0.0)
    print(f"graph vs eager: {exact}/{len(report)} tensors bitwise equal; max diff {max(v[0] for v in report.values()):.3e}")

    # dropout masks change from replay to replay
    DropoutStream.use_device_counter(torch.device(DEV))
    DropoutStream.manual_seed(4)
    m3 = _make(0.5)
    o3 = make_optimizer(m3, lr=0.0, weight_decay=0.0, capturable=True)     # frozen weights: only the mask varies
    outs = []
    h = m3.st_blocks[0].register_forward_hook(lambda mod, i, o: outs.append(o))
    gs3 = GraphedTrainStep(m3, o3, xs[0], ys[0], warmup=1)
    h.remove()
    static_out = outs[-1]                 # the block output tensor captured in the graph
    pats = []
    for _ in range(3):
        gs3(xs[0], ys[0])
        torch.cuda.synchronize()
        pats.append((static_out != 0).clone())
    assert abs(pats[0].float().mean().item() - 0.5) < 0.01
    assert (pats[0] != pats[1]).float().mean().item() > 0.4 and (pats[1] != pats[2]).float().mean().item() > 0.4
    DropoutStream.disable_device_counter()


def test_graphed_training_reduces_loss():
    from stgcn_amd import DropoutStream
    from stgcn_amd.train import GraphedTrainStep, make_optimizer
    g = torch.Generator().manual_seed(5)
    x = torch.randn(32, 1, 12, 207, generator=g).to(DEV)
    y = (x[:, 0, -1, :] * 0.5).contiguous()        # learnable target: half of the last observation
    m = _make(0.5)
    o = make_optimizer(m, capturable=True)
    gs = GraphedTrainStep(m, o, x, y, warmup=1)
    first = float(gs(x, y).item())
    for _ in range(150):
        last = gs(x, y)
    DropoutStream.disable_device_counter()
    assert float(last.item()) < 0.6 * first


def test_chained_graph_step_equals_single_chain():
    """The minibatch as 2 / 4 concurrent micro-batch chains (train.chained_fwd_bwd, one stream each, captured into the same
    hipGraph) trains to the same parameters as the single chain (droprate 0: no mask dependence)."""
    from stgcn_amd import DropoutStream
    from stgcn_amd.train import GraphedTrainStep, make_optimizer
    g = torch.Generator().manual_seed(2)
    xs = torch.randn(5, 8, 1, 12, 207, generator=g).to(DEV)
    ys = torch.randn(5, 8, 207, generator=g).to(DEV)
    finals = []
    for chains in (1, 2, 4):
        DropoutStream.use_device_counter(torch.device(DEV))
        DropoutStream.manual_seed(3)
        m = _make(0.0)
        o = make_optimizer(m, capturable=True)
        gs = GraphedTrainStep(m, o, xs[0], ys[0], warmup=2, chains=chains)
        losses = [float(gs(xs[i], ys[i]).item()) for i in range(1, 5)]
        torch.cuda.synchronize()
        finals.append((losses, {k: v.clone() for k, v in m.state_dict().items()}))
        DropoutStream.disable_device_counter()
    for losses, sd in finals[1:]:
        assert np.allclose(losses, finals[0][0], rtol=1e-5, atol=0), (losses, finals[0][0])
        for k, v in sd.items():
            d = float((v - finals[0][1][k]).abs().max())
            assert d <= 2e-5, (k, d)       # AdamW normalises gradients: tiny summation-order differences stay tiny


def test_resident_series_step_equals_explicit_batches():
    """GraphedTrainStep(series=...): device-side windows of a resident series, batch position advanced on the device by the pack
    launch, trains exactly like the same windows materialised as (B, 1, n_his, N) tensors and fed one batch per step."""
    from stgcn_amd import DropoutStream
    from stgcn_amd.train import GraphedTrainStep, make_optimizer, train_step
    n_his, n_pred, B, N = 12, 3, 8, 207
    g = torch.Generator().manual_seed(6)
    series = torch.randn(5 * B + n_his + n_pred, N, generator=g).to(DEV)    # 5 minibatches of windows (num = rows - n_his - n_pred,
                                                                            # script/dataloader.py:36), then it wraps

    def windows(s):
        x = torch.stack([series[s + b:s + b + n_his] for b in range(B)]).unsqueeze(1).contiguous()
        y = torch.stack([series[s + b + n_his + n_pred - 1] for b in range(B)]).contiguous()
        return x, y

    DropoutStream.use_device_counter(torch.device(DEV))
    DropoutStream.manual_seed(3)
    m = _make(0.0)
    o = make_optimizer(m, capturable=True)
    gs = GraphedTrainStep(m, o, *windows(0), warmup=2, series=series, n_his=n_his, n_pred=n_pred)
    assert gs.fold, "the pack launch should carry the batch position"
    losses = [float(gs().item()) for _ in range(4)]        # positions 3B, 4B, 0 (wrap), B after the constructor's 0, B, 2B
    torch.cuda.synchronize()
    assert int(gs.index.item()) == B
    DropoutStream.disable_device_counter()
    m2 = _make(0.0)
    o2 = make_optimizer(m2)
    ref_losses = [float(train_step(m2, o2, *windows((k * B) % (5 * B))).item()) for k in range(3 + 4)]
    torch.cuda.synchronize()
    assert np.allclose(losses, ref_losses[-4:], rtol=1e-5, atol=0), (losses, ref_losses)
    for k, v in m2.state_dict().items():
        assert float((v - m.state_dict()[k]).abs().max()) <= 2e-5, k


def test_tail_step_between_replays_keeps_the_window_index():
    """ADVICE r2 / VERDICT r3 weak 3, on the hardware: a captured step with device-side windows and FOLDED counters (the pack launch advances
    the window index), then the eager partial batch of an epoch's end (train.tail_step), then replays again.  The tail step must not move
    the window index, must consume one dropout position and one optimizer step, and the whole sequence must equal the same sequence of
    eager steps on a twin model."""
    from stgcn_amd import DropoutStream
    from stgcn_amd.train import GraphedTrainStep, make_optimizer, tail_step, train_step
    B, N, n_his, n_pred = 8, 207, 12, 3
    g = torch.Generator().manual_seed(5)
    series = torch.randn(6 * B + n_his + n_pred, N, generator=g).to(DEV)
    xt = torch.randn(5, 1, n_his, N, generator=g).to(DEV)          # the partial batch: 5 of 8 windows
    yt = torch.randn(5, N, generator=g).to(DEV)

    def windows(s):
        x = torch.stack([series[s + b:s + b + n_his] for b in range(B)]).unsqueeze(1)
        return x, series[s + n_his + n_pred - 1:s + n_his + n_pred - 1 + B]

    DropoutStream.use_device_counter(torch.device(DEV))
    DropoutStream.manual_seed(3)
    m1 = _make(0.0)
    o1 = make_optimizer(m1, capturable=True)
    with GraphedTrainStep(m1, o1, *windows(0), warmup=2, series=series, n_his=n_his, n_pred=n_pred) as gs:
        assert gs.fold and gs.index is not None
        usable = (series.shape[0] - n_his - n_pred) // B * B
        l1 = [float(gs())]
        i_before = int(gs.index.item())
        c_before, s_before = int(DropoutStream.counter.item()), int(o1.device_step_counter(torch.device(DEV)).item())
        lt = float(tail_step(m1, o1, xt, yt))
        assert int(gs.index.item()) == i_before                                              # the window position is untouched ...
        assert int(DropoutStream.counter.item()) == c_before + DropoutStream.SITE_STRIDE      # ... one dropout position consumed
        assert int(o1.device_step_counter(torch.device(DEV)).item()) == s_before + 1          # ... one optimizer step counted
        l1 += [float(gs()), float(gs())]
        assert int(gs.index.item()) == (i_before + 2 * B) % usable
    # the same sequence eagerly: the constructor ran warmup + 1 steps on windows 0, B, 2B (folded: the pack launch advances BEFORE the step)
    DropoutStream.disable_device_counter()
    m2 = _make(0.0)
    o2 = make_optimizer(m2)
    pos = 0
    for _ in range(3):
        train_step(m2, o2, *windows(pos % usable))
        pos += B
    l2 = [float(train_step(m2, o2, *windows(pos % usable)))]
    pos += B
    lt2 = float(tail_step(m2, o2, xt, yt))
    for _ in range(2):
        l2.append(float(train_step(m2, o2, *windows(pos % usable))))
        pos += B
    assert np.allclose(l1 + [lt], l2 + [lt2], rtol=2e-5, atol=0), (l1, lt, l2, lt2)
    sd1, sd2 = m1.state_dict(), m2.state_dict()
    assert max(float((sd1[k] - sd2[k]).abs().max()) for k in sd1) <= 2e-5
