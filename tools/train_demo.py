#!/usr/bin/env python3
"""End-to-end run of the reference's main.py flow on the MI355X path with a synthetic METR-LA-shaped dataset
(34 272 rows x 207 sensors, v = clip(55 + 10 sin(2 pi t / 288 + phi_n) + N(0, 3^2), 0, 80); the real
vel.csv is not available offline): 70/15/15 split (main.py:108-114), z-score fitted on the training rows (main.py:116-119),
12 -> n_pred windows (script/dataloader.py:32-47), STGCNChebGraphConv on the real METR-LA operator, MSE + AdamW(1e-3, 1e-3)
with StepLR(10, 0.95) per epoch (main.py:147-156), unshuffled minibatches of 32 (main.py:126-131), validation loss per epoch
(script/utility.py:90-101), test MAE / RMSE / WMAPE at the end (script/utility.py:103-121).

The training loop is `GraphedTrainStep(series=...)`: the z-scored (time, N) training series stays resident on the GPU and the
captured step windows it in place.  One JSON line per epoch and one for the test metrics; a persistence forecast
(y_hat = last observed value) on the same windows is printed beside them for scale.

  python tools/train_demo.py [--epochs 3] [--n-pred 3] [--rows 34272]
"""
import argparse
import json
import os
import sys
import time
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

N_HIS, KT, KS, BS = 12, 3, 3, 32
BLOCKS = [[1], [64, 16, 64], [64, 16, 64], [128, 128], [1]]


def synthetic_speeds(rows, n):
    rng = np.random.default_rng(0)
    phi = rng.uniform(0, 2 * np.pi, n)
    t = np.arange(rows)[:, None]
    return np.clip(55 + 10 * np.sin(2 * np.pi * t / 288 + phi[None, :]) + rng.normal(0, 3, (rows, n)), 0, 80)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=3)
    ap.add_argument("--n-pred", type=int, default=3)
    ap.add_argument("--rows", type=int, default=34272)
    a = ap.parse_args()

    from stgcn_amd import DropoutStream, data, models
    from stgcn_amd.train import GraphedTrainStep, make_optimizer

    assert torch.cuda.is_available(), "needs the MI355X (no CPU fallback)"
    dev = torch.device("cuda", 0)
    gso_np = np.load(os.path.join(ROOT, "tests", "golden", "gso_real.npz"))["metr_la.cheb_sym_norm_lap"]
    n = gso_np.shape[0]
    vel = synthetic_speeds(a.rows, n)
    len_train, len_val, len_test = data.split_lengths(a.rows)
    zs = data.ZScore()
    train = zs.fit_transform(vel[:len_train])
    val, test = zs.transform(vel[len_train:len_train + len_val]), zs.transform(vel[len_train + len_val:])

    args = types.SimpleNamespace(Kt=KT, Ks=KS, act_func="glu", graph_conv_type="cheb_graph_conv", gso=torch.from_numpy(gso_np).to(dev),
                                 enable_bias=True, droprate=0.5, n_his=N_HIS)
    torch.manual_seed(42)
    model = models.STGCNChebGraphConv(args, BLOCKS, n).to(dev)
    DropoutStream.manual_seed(42)
    opt = make_optimizer(model, lr=1e-3, weight_decay=1e-3, capturable=True)
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=10, gamma=0.95)

    series = torch.from_numpy(train.astype(np.float32)).to(dev)
    x0 = torch.zeros(BS, 1, N_HIS, n, device=dev)
    y0 = torch.zeros(BS, n, device=dev)
    model.train()
    step = GraphedTrainStep(model, opt, x0, y0, series=series, n_his=N_HIS, n_pred=a.n_pred)
    windows = series.shape[0] - N_HIS - a.n_pred + 1
    steps_per_epoch = windows // BS
    val_s = data.WindowSampler(val, N_HIS, a.n_pred, dev)
    test_s = data.WindowSampler(test, N_HIS, a.n_pred, dev)
    mse = torch.nn.MSELoss()

    for epoch in range(a.epochs):
        model.train()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        acc = torch.zeros((), device=dev)
        for _ in range(steps_per_epoch):
            acc += step()                       # loss stays on the device: no per-step host sync (main.py:170 does .item())
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        step.check()                            # (once per epoch, where the loss is read: names the operator if an in-launch wait gave up)
        sched.step()
        opt.sync_lr()                           # the captured AdamW reads its learning rate from device memory
        val_loss = data.evaluate_model(model, mse, val_s.batches(BS))
        print(json.dumps({"epoch": epoch + 1, "train_loss": round(float(acc.item()) / steps_per_epoch, 6), "val_loss": round(val_loss, 6),
                          "lr": opt.param_groups[0]["lr"], "steps": steps_per_epoch, "epoch_s": round(el, 3),
                          "train_windows_per_s": round(steps_per_epoch * BS / el, 1)}), flush=True)

    mae, rmse, wmape = data.evaluate_metric(model, test_s.batches(BS), zs)
    # persistence forecast on the same test windows, in the original units
    ys, ps = [], []
    for x, y in test_s.batches(BS):
        ys.append(zs.inverse_transform(y.cpu().numpy()).reshape(-1))
        ps.append(zs.inverse_transform(x[:, 0, -1, :].cpu().numpy()).reshape(-1))
    p_mae, p_rmse, p_wmape = data.metrics_from_arrays(np.concatenate(ys), np.concatenate(ps))
    print(json.dumps({"test": {"MAE": round(mae, 4), "RMSE": round(rmse, 4), "WMAPE": round(wmape, 6)},
                      "persistence_forecast": {"MAE": round(p_mae, 4), "RMSE": round(p_rmse, 4), "WMAPE": round(p_wmape, 6)},
                      "test_windows": len(test_s), "n_pred": a.n_pred, "data": "synthetic METR-LA-shaped speeds, real METR-LA graph"}), flush=True)


if __name__ == "__main__":
    main()
