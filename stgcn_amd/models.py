"""Drop-in model classes: ``STGCNChebGraphConv(args, blocks, n_vertex)`` and
``STGCNGraphConv(args, blocks, n_vertex)`` with the reference's signatures, attribute names and
``state_dict`` keys (hazdzz/STGCN model/models.py:28-53 and :78-103), assembled from the fused
``layers.STConvBlock``.  Input (B, 1, n_his, N) -> output (B, 1, 1, N) as at main.py:166.

``args`` needs exactly the attributes the reference reads (models.py:32-42): Kt, Ks, act_func,
graph_conv_type, gso, enable_bias, droprate, n_his.
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn

from . import layers


class _STGCNBase(nn.Module):
    def __init__(self, args, blocks, n_vertex):
        super().__init__()
        n_st = len(blocks) - 3
        self.st_blocks = nn.Sequential(*[
            layers.STConvBlock(args.Kt, args.Ks, n_vertex, blocks[l][-1], blocks[l + 1], args.act_func,
                               args.graph_conv_type, args.gso, args.enable_bias, args.droprate)
            for l in range(n_st)])
        for l, blk in enumerate(self.st_blocks):
            blk.cfg = blk.cfg.__class__(**{**blk.cfg.__dict__, "tag": l})      # block index, for the kernel timer labels
        self.Ko = args.n_his - n_st * 2 * (args.Kt - 1)
        if self.Ko > 1:
            self.output = layers.OutputBlock(self.Ko, blocks[-3][-1], blocks[-2], blocks[-1][0], n_vertex,
                                             args.act_func, args.enable_bias, args.droprate)
        elif self.Ko == 0:
            self.fc1 = nn.Linear(in_features=blocks[-3][-1], out_features=blocks[-2][0], bias=args.enable_bias)
            self.fc2 = nn.Linear(in_features=blocks[-2][0], out_features=blocks[-1][0], bias=args.enable_bias)
            self.relu = nn.ReLU()
            self._make_dropout(args.droprate)
        # Ko == 1: like the reference, no head at all (models.py:46-51)

    def _make_dropout(self, p):
        self.dropout = nn.Dropout(p=p)

    def _prepack(self, x):
        """All weight packs of the step in one launch (the per-module pack launches are skipped for this forward)."""
        from . import ops, _lib
        from .layers import STConvBlock, OutputBlock
        if x.dim() != 4 or not (x.is_cuda or _lib.lib().is_emulator) or os.environ.get("STGCN_PREPACK", "1") == "0":
            return []            # (the modules raise the proper error themselves / tuning knob: per-module pack launches)
        blocks, T = [], x.shape[2]
        for blk in self.st_blocks:
            if not isinstance(blk, STConvBlock):
                return []
            blocks.append((blk.cfg, T, blk._params(), blk._ws))
            T -= 2 * (blk.cfg.Kt - 1)
        head = None
        if self.Ko > 1 and isinstance(self.output, OutputBlock) and ops.head_supported(self.output.cfg):
            head = (self.output.cfg, T, self.output._params(), self.output._ws)
        # step counters a trainer asked to advance with the first launch of each TRAINING forward (train.GraphedTrainStep)
        counters = getattr(self, "_step_counters", None) if self.training else None
        ops.prepack_modules(blocks, head, x.shape[0], x.device, counters, dtype=x.dtype, park=True)      # (forward() runs the first block next)
        return [b[3] for b in blocks] + ([head[3]] if head is not None else [])

    def set_compute_dtype(self, dtype):
        """Storage / arithmetic type of the activations: ``torch.float32`` (the reference's arithmetic, default) or ``torch.bfloat16``
        (BASELINE.json configs[2], [4]: bf16 activations and saved tensors, bf16 matrix cores with fp32 accumulation; parameters,
        LayerNorm statistics, gradients and the optimizer stay fp32).  A float32 input is cast once at the model's entry; the
        prediction comes back as float32.  No counterpart in the reference (it runs fp32 only)."""
        if dtype not in (torch.float32, torch.bfloat16):
            raise ValueError(f"compute dtype must be torch.float32 or torch.bfloat16, got {dtype}")
        self.compute_dtype = dtype
        return self

    def forward(self, x):
        cd = getattr(self, "compute_dtype", None)
        if cd is not None and x.dtype != cd and x.is_floating_point():
            x = x.to(cd)
        marked = self._prepack(x)
        try:
            x = self.st_blocks(x)
            if self.Ko > 1:
                x = self.output(x)
            elif self.Ko == 0:
                x = self.fc1(x.permute(0, 2, 3, 1).float())
                x = self.relu(x)
                x = self.fc2(x).permute(0, 3, 1, 2)
        finally:
            for wsc in marked:      # a module that did not run (exception) must pack by itself next time
                wsc.prepacked = False
            if marked:              # (a pack still parked -- the first block never reached the library -- goes out now, while its tensors live)
                from . import ops
                ops.prepack_flush()
        return x if x.dtype == torch.float32 else x.float()      # (Ko == 1 hands the last block's output through)


class STGCNChebGraphConv(_STGCNBase):
    """'TGTND TGTND TNFF' with Chebyshev graph convolution (models.py:6-53)."""


class STGCNGraphConv(_STGCNBase):
    """Same structure with the first-order (Kipf) graph convolution (models.py:55-103); the reference
    names the unused Ko == 0 dropout ``do`` in this class (models.py:92)."""

    def _make_dropout(self, p):
        self.do = nn.Dropout(p=p)
