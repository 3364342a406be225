"""Data formats on either side of the hot path, restated from the reference's ``script/`` helpers so that the
reference's datasets (``adj.npz`` / ``vel.csv``) and metric definitions drop in unchanged.

  calc_gso / calc_chebynet_gso   script/utility.py:6-57, 59-76   (graph shift operator for args.gso)
  data_transform                 script/dataloader.py:32-47      (sliding windows x:(num,1,n_his,N), y:(num,N))
  WindowSampler                  device-side windowing: the (rows, N) series stays on the GPU and minibatches are
                                 gathered on the fly instead of materialising the 12x replicated window tensor
                                 (SURVEY.md section 8f #3); batches are identical to the reference's DataLoader(shuffle=False)
  ZScore                         sklearn.preprocessing.StandardScaler as used at main.py:116-119
  evaluate_model / evaluate_metric   script/utility.py:90-101, 103-121 (MSE; MAE, RMSE, WMAPE after inverse transform)

These are host-side one-off or per-epoch helpers, not part of the timed training step.
"""
from __future__ import annotations

import math
from typing import Iterator, Optional, Tuple

import numpy as np
import scipy.sparse as sp
import torch


# ------------------------------------------------------------------------------------------------ graph shift operator
def calc_gso(dir_adj, gso_type: str):
    """script/utility.py:6-57.  Returns a scipy CSC matrix.

    sym_*: D^-1/2 A D^-1/2 (optionally A+I first, optionally I - .); rw_*: D^-1 A.  (The reference's rw_* branch fails
    under current scipy/numpy because it mixes a dense np.diag with a sparse matrix, utility.py:44-46; the intended
    operator D^-1 A is computed here with sparse diagonals.)"""
    n = dir_adj.shape[0]
    adj = sp.csc_matrix(dir_adj) if not sp.issparse(dir_adj) else dir_adj.tocsc()
    ident = sp.identity(n, format="csc")
    at = adj.T
    gt = at > adj
    adj = adj + at.multiply(gt) - adj.multiply(gt)                     # symmetrise by taking the larger weight (:17)
    if gso_type in ("sym_renorm_adj", "rw_renorm_adj", "sym_renorm_lap", "rw_renorm_lap"):
        adj = adj + ident                                              # renormalisation trick (:20-22)
    deg = np.asarray(adj.sum(axis=1)).reshape(-1)
    if gso_type in ("sym_norm_adj", "sym_renorm_adj", "sym_norm_lap", "sym_renorm_lap"):
        with np.errstate(divide="ignore"):
            dis = np.power(deg, -0.5)
        dis[np.isinf(dis)] = 0.0
        d = sp.diags(dis, format="csc")
        norm_adj = d.dot(adj).dot(d)                                   # :24-31
        return (ident - norm_adj) if gso_type.endswith("lap") else norm_adj
    if gso_type in ("rw_norm_adj", "rw_renorm_adj", "rw_norm_lap", "rw_renorm_lap"):
        with np.errstate(divide="ignore"):
            di = np.power(deg, -1.0)
        di[np.isinf(di)] = 0.0
        norm_adj = sp.diags(di, format="csc").dot(adj)                 # :39-46 (intended math)
        return (ident - norm_adj) if gso_type.endswith("lap") else norm_adj
    raise ValueError(f"{gso_type} is not defined.")                    # :55


def calc_chebynet_gso(gso, lambda_max: str = "scipy_norm2", seed=None):
    """script/utility.py:59-76: 2 L / lambda_max - I (or L - I when lambda_max >= 2).

    lambda_max="scipy_norm2" (default, drop-in): scipy.sparse.linalg.norm(gso, 2) exactly as the reference calls it.  That
    routine is an un-converged randomised solver whose result depends on numpy's global RNG state (SURVEY.md section 8c
    hazard 1: 1.00955 vs the true 1.01200 for METR-LA under np.random.seed(42)); pass ``seed`` to pin it (numpy's global
    seed is set immediately before the call, like the fixtures of tests/golden do, and the caller's RNG state is restored
    afterwards), or leave it None to inherit the caller's RNG state like the reference does.  Data-parallel runs: every rank
    builds its own operator, so either pass the same ``seed`` on every rank or call ``train.sync_operators(model)``.
    lambda_max="exact": the true largest singular value via dense LAPACK -- deterministic, but a ~0.25 % different
    operator from the one the reference trains with."""
    gso = sp.csc_matrix(gso) if not sp.issparse(gso) else gso.tocsc()
    ident = sp.identity(gso.shape[0], format="csc")
    if lambda_max == "scipy_norm2":
        from scipy.sparse.linalg import norm
        if seed is not None:
            state = np.random.get_state()          # the pin must not clobber the caller's global RNG stream
            np.random.seed(int(seed))
            try:
                eig = norm(gso, 2)
            finally:
                np.random.set_state(state)
        else:
            eig = norm(gso, 2)
    elif lambda_max == "exact":
        eig = float(np.linalg.norm(gso.toarray(), 2))
    else:
        raise ValueError(f"unknown lambda_max mode {lambda_max}")
    if eig >= 2:
        return gso - ident
    return 2 * gso / eig - ident


def gso_tensor(gso, device) -> torch.Tensor:
    """main.py:101-103: dense float32 operator on the device (what ``args.gso`` holds)."""
    dense = gso.toarray() if sp.issparse(gso) else np.asarray(gso)
    return torch.from_numpy(dense.astype(np.float32)).to(device)


# ------------------------------------------------------------------------------------------------ windows
def split_lengths(n_rows: int, val_and_test_rate: float = 0.15) -> Tuple[int, int, int]:
    """main.py:108-114 (70/15/15 chronological split)."""
    len_val = int(math.floor(n_rows * val_and_test_rate))
    len_test = int(math.floor(n_rows * val_and_test_rate))
    return n_rows - len_val - len_test, len_val, len_test


def data_transform(data, n_his: int, n_pred: int, device) -> Tuple[torch.Tensor, torch.Tensor]:
    """script/dataloader.py:32-47 without the Python loop: x[i] = data[i : i+n_his], y[i] = data[i + n_his + n_pred - 1]."""
    data = np.asarray(data)
    num = len(data) - n_his - n_pred
    n_vertex = data.shape[1]
    if num <= 0:
        return torch.zeros(0, 1, n_his, n_vertex, device=device), torch.zeros(0, n_vertex, device=device)
    win = np.lib.stride_tricks.sliding_window_view(data, (n_his, n_vertex))[:num, 0]      # (num, n_his, N) view
    x = torch.from_numpy(np.ascontiguousarray(win, dtype=np.float32)).unsqueeze(1)
    y = torch.from_numpy(np.ascontiguousarray(data[n_his + n_pred - 1: n_his + n_pred - 1 + num], dtype=np.float32))
    return x.to(device), y.to(device)


class WindowSampler:
    """Device-side windowing: keeps the (rows, N) float32 series on the device and gathers each minibatch with one
    index op.  ``batches(bs)`` yields exactly the (x, y) pairs of DataLoader(TensorDataset(*data_transform(...)),
    batch_size=bs, shuffle=False) (main.py:126-131); ``rank``/``world`` give the rank-strided shard of every global
    batch used for data parallelism."""

    def __init__(self, series, n_his: int, n_pred: int, device):
        s = torch.as_tensor(np.asarray(series), dtype=torch.float32) if not torch.is_tensor(series) else series.float()
        self.series = s.to(device).contiguous()
        self.n_his, self.n_pred = n_his, n_pred
        self.num = max(0, self.series.shape[0] - n_his - n_pred)
        self._t = torch.arange(n_his, device=device)

    def __len__(self):
        return self.num

    def gather(self, idx: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        rows = idx[:, None] + self._t[None, :]                      # (B, n_his)
        x = self.series[rows].unsqueeze(1)                          # (B, 1, n_his, N)
        y = self.series[idx + (self.n_his + self.n_pred - 1)]       # (B, N)
        return x, y

    def batches(self, batch_size: int, rank: int = 0, world: int = 1) -> Iterator[Tuple[torch.Tensor, torch.Tensor]]:
        gb = batch_size * world
        dev = self.series.device
        for start in range(0, self.num, gb):
            lo = start + rank * batch_size
            hi = min(lo + batch_size, min(start + gb, self.num))
            if hi > lo:
                yield self.gather(torch.arange(lo, hi, device=dev))


class ZScore:
    """sklearn.preprocessing.StandardScaler() as used at main.py:116-119: per-column mean / population std."""

    def fit(self, a):
        a = np.asarray(a, dtype=np.float64)
        self.mean_ = a.mean(axis=0)
        var = a.var(axis=0)
        self.scale_ = np.sqrt(var)
        self.scale_[self.scale_ == 0.0] = 1.0
        return self

    def transform(self, a):
        return (np.asarray(a, dtype=np.float64) - self.mean_) / self.scale_

    def fit_transform(self, a):
        return self.fit(a).transform(a)

    def inverse_transform(self, a):
        return np.asarray(a) * self.scale_ + self.mean_


# ------------------------------------------------------------------------------------------------ evaluation
@torch.no_grad()
def evaluate_model(model, loss, data_iter) -> float:
    """script/utility.py:90-101: sample-weighted mean of the per-batch loss."""
    model.eval()
    l_sum, n = 0.0, 0
    for x, y in data_iter:
        y_pred = model(x).view(len(x), -1)
        l_sum += loss(y_pred, y).item() * y.shape[0]
        n += y.shape[0]
    return l_sum / n


def metrics_from_arrays(y_true: np.ndarray, y_pred: np.ndarray) -> Tuple[float, float, float]:
    """MAE, RMSE, WMAPE of script/utility.py:111-121 on inverse-transformed flat arrays."""
    d = np.abs(np.asarray(y_true, dtype=np.float64) - np.asarray(y_pred, dtype=np.float64))
    return float(d.mean()), float(np.sqrt((d ** 2).mean())), float(d.sum() / np.asarray(y_true, dtype=np.float64).sum())


@torch.no_grad()
def evaluate_metric(model, data_iter, scaler) -> Tuple[float, float, float]:
    """script/utility.py:103-121."""
    model.eval()
    ys, ps = [], []
    for x, y in data_iter:
        ys.append(scaler.inverse_transform(y.cpu().numpy()).reshape(-1))
        ps.append(scaler.inverse_transform(model(x).view(len(x), -1).cpu().numpy()).reshape(-1))
    return metrics_from_arrays(np.concatenate(ys), np.concatenate(ps))
