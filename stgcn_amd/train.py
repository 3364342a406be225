"""Data-parallel training step of the reference loop body (main.py:164-171) on MI355X.

One process per GPU (torchrun / torch.distributed, backend "nccl" = RCCL over xGMI).  The model is
replicated (245 K parameters + a 171 KB operator), the minibatch is sharded: rank r takes windows
[step*B_global + r*B_local, ... + B_local) of the unshuffled index (the reference iterates with
shuffle=False, main.py:127), so W ranks x bs 32 reproduces one device at bs 32*W.  The only exchange is
ONE all-reduce per step over a flat fp32 gradient buffer (~0.98 MB for METR-LA): latency-bound, not
link-bound on xGMI, so no bucketing.

Parameters that never receive a gradient (10 unused align convs) are excluded from the flat buffer and
from the optimizer's work exactly like torch.optim.AdamW skips ``grad is None`` in the reference.
"""
from __future__ import annotations

import os
from typing import List, Optional

import torch
import torch.distributed as dist


def dist_env():
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    return rank, local_rank, world


def init_distributed(backend: Optional[str] = None):
    rank, local_rank, world = dist_env()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


class FlatGradAllReduce:
    """Sum-all-reduce of every live gradient in one collective, then scale by 1/world."""

    def __init__(self, params: List[torch.nn.Parameter], world: int):
        self.params = list(params)
        self.world = world
        self.flat: Optional[torch.Tensor] = None
        self.live: Optional[List[torch.nn.Parameter]] = None

    def __call__(self):
        if self.world <= 1:
            return
        live = [p for p in self.params if p.grad is not None]
        if self.live is None or len(live) != len(self.live):
            self.live = live
            n = sum(p.numel() for p in live)
            self.flat = torch.empty(n, dtype=torch.float32, device=live[0].device)
        grads = [p.grad.reshape(-1) for p in live]
        torch.cat(grads, out=self.flat)
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
        self.flat.mul_(1.0 / self.world)
        torch._foreach_copy_([p.grad.reshape(-1) for p in live], list(self.flat.split([p.numel() for p in live])))


def make_optimizer(model: torch.nn.Module, lr: float = 1e-3, weight_decay: float = 1e-3, name: str = "adamw"):
    """main.py:147-154 (adamw is the default; nadamw is NAdam with decoupled decay)."""
    params = list(model.parameters())
    on_gpu = len(params) > 0 and params[0].is_cuda
    if name == "adamw":
        return torch.optim.AdamW(params, lr=lr, weight_decay=weight_decay, fused=True if on_gpu else None)
    if name == "nadamw":
        return torch.optim.NAdam(params, lr=lr, weight_decay=weight_decay, decoupled_weight_decay=True)
    raise ValueError(f"ERROR: The {name} optimizer is undefined.")   # main.py:154


def train_step(model, optimizer, x, y, allreduce: Optional[FlatGradAllReduce] = None):
    """zero_grad -> forward -> MSELoss -> backward -> [all-reduce] -> optimizer.step (main.py:165-169).
    Returns the loss tensor (no host sync: the reference's per-step .item() at main.py:170 is deferred)."""
    optimizer.zero_grad(set_to_none=True)
    y_pred = model(x).reshape(len(x), -1)
    loss = torch.nn.functional.mse_loss(y_pred, y)
    loss.backward()
    if allreduce is not None:
        allreduce()
    optimizer.step()
    return loss.detach()
