"""AdamW whose whole step is ONE HIP kernel (``stgcn_adamw_step``) instead of the per-tensor loops /
multi-tensor launches of ``torch.optim.AdamW`` (reference: main.py:147-148, step at main.py:169).

Semantics follow torch.optim.AdamW with amsgrad=False, maximize=False: decoupled weight decay, bias-corrected
moments, and -- like the reference's optimizer -- parameters whose ``.grad`` is None are skipped entirely (no
decay, no state).  It subclasses ``torch.optim.Optimizer`` so LR schedulers (StepLR at main.py:156) work.

``capturable=True`` keeps the step count and the learning rate in device memory so that ``step()`` can be
recorded into a hipGraph; call ``sync_lr()`` after a scheduler changed ``param_groups[i]['lr']``.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


class AdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, capturable=False):
        if lr < 0.0 or eps < 0.0 or not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0 or weight_decay < 0.0:
            raise ValueError("invalid AdamW hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.capturable = capturable
        self._dev = {}     # per group: (step tensor, lr tensor) on the device (capturable mode)
        self.trainer_owns_step = False   # a trainer advances the device step count itself (stgcn_prepack counters): step() must not

    def sync_lr(self):
        for gi, group in enumerate(self.param_groups):
            if gi in self._dev:
                self._dev[gi][1].fill_(float(group["lr"]))

    @torch.no_grad()
    def device_step_counter(self, device) -> torch.Tensor:
        """The device-side step count of group 0 (capturable mode), created on first use.  A trainer that advances it
        itself (e.g. on the weight-pack launch at the start of the step, ``stgcn_prepack`` counters) passes
        ``bump_step=False`` to ``flush_with``."""
        group = self.param_groups[0]
        if 0 not in self._dev:
            self._dev[0] = (torch.zeros(1, dtype=torch.int64, device=device), torch.full((1,), float(group["lr"]), device=device))
        return self._dev[0][0]

    def flush_with(self, sink, grads, bump_step: bool = True):
        """``sink.flush()`` + ``step()`` in ONE launch (stgcn_grad_flush with the optimizer table): every gradient element is
        reduced from the backward partials and consumed by AdamW on the spot.  ``grads``: {parameter: gradient buffer} of the
        sink (all live parameters of the single param group); same arithmetic and state as ``step()``."""
        from . import ops
        assert len(self.param_groups) == 1, "flush_with handles one parameter group"
        group = self.param_groups[0]
        live = [p for p in group["params"] if p in grads]
        assert len(live) == len(grads), "gradient sink holds parameters this optimizer does not own"
        dev = live[0].device
        table = (_lib.AdamwTensor * len(live))()
        for i, p in enumerate(live):
            if p.dtype != torch.float32 or not p.is_contiguous():
                raise RuntimeError("stgcn_amd.optim.AdamW handles contiguous float32 parameters only")
            st = self.state[p]
            if not st:
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
            table[i].param, table[i].grad = p.data_ptr(), grads[p].data_ptr()
            table[i].exp_avg, table[i].exp_avg_sq, table[i].numel = st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel()
        hyper = _lib.AdamwHyper()
        b1, b2 = group["betas"]
        hyper.lr, hyper.beta1, hyper.beta2, hyper.eps, hyper.weight_decay = float(group["lr"]), float(b1), float(b2), float(group["eps"]), float(group["weight_decay"])
        if self.capturable and dev.type == "cuda":
            self.device_step_counter(dev)
            step_t, lr_t = self._dev[0]
            if bump_step:
                step_t.add_(1)
            hyper.step, hyper.step_dev, hyper.lr_dev = 0, step_t.data_ptr(), lr_t.data_ptr()
        else:
            group["_step"] = group.get("_step", 0) + 1
            hyper.step, hyper.step_dev, hyper.lr_dev = group["_step"], None, None
        stream = torch.cuda.current_stream(dev).cuda_stream if dev.type == "cuda" else None
        sink.flush(table, hyper, stream)

    def state_dict(self):
        """torch's state_dict plus the step counts, which capturable mode keeps in device tensors outside ``state`` (without them a
        resumed run would restart the bias correction at t = 1: a ~10x too large first update)."""
        sd = super().state_dict()
        for gi, g in enumerate(sd["param_groups"]):
            if gi in self._dev:
                g["_step"] = int(self._dev[gi][0].item())
        return sd

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        for gi, group in enumerate(self.param_groups):
            if self.capturable and "_step" in group:
                live = [p for p in group["params"]]
                if live and live[0].is_cuda:
                    dev = live[0].device
                    if gi not in self._dev:
                        self._dev[gi] = (torch.zeros(1, dtype=torch.int64, device=dev), torch.full((1,), float(group["lr"]), device=dev))
                    self._dev[gi][0].fill_(int(group["_step"]))
                    self._dev[gi][1].fill_(float(group["lr"]))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        L = _lib.lib()
        for gi, group in enumerate(self.param_groups):
            live = [p for p in group["params"] if p.grad is not None]
            if not live:
                continue
            dev = live[0].device
            for p in live:
                if p.dtype != torch.float32 or not p.is_contiguous() or not p.grad.is_contiguous():
                    raise RuntimeError("stgcn_amd.optim.AdamW handles contiguous float32 parameters only")
                st = self.state[p]
                if not st:
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
            step_dev = lr_dev = None
            if self.capturable and dev.type == "cuda":
                if gi not in self._dev:
                    self._dev[gi] = (torch.zeros(1, dtype=torch.int64, device=dev), torch.full((1,), float(group["lr"]), device=dev))
                step_t, lr_t = self._dev[gi]
                if not self.trainer_owns_step:
                    step_t.add_(1)
                step_dev, lr_dev = step_t.data_ptr(), lr_t.data_ptr()
                step = 0
            else:
                group["_step"] = group.get("_step", 0) + 1
                step = group["_step"]
            table = (_lib.AdamwTensor * len(live))()
            for i, p in enumerate(live):
                st = self.state[p]
                table[i].param, table[i].grad = p.data_ptr(), p.grad.data_ptr()
                table[i].exp_avg, table[i].exp_avg_sq, table[i].numel = st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel()
            stream = torch.cuda.current_stream(dev).cuda_stream if dev.type == "cuda" else None
            b1, b2 = group["betas"]
            L.check(L.dll.stgcn_adamw_step(table, len(live), float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                                           float(group["weight_decay"]), step, step_dev, lr_dev, stream), "stgcn_adamw_step")
        return loss
