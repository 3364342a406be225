"""Drop-in operator modules: same class names, constructor signatures, parameter names/shapes
(``state_dict`` keys) and initialisation order as the reference's ``model/layers.py``, so that the
reference's ``main.py`` / checkpoints work unchanged -- but ``STConvBlock.forward`` is ONE fused HIP
operator (fwd) + ONE (bwd) on MI355X instead of ~60 ATen launches.

Reference map (hazdzz/STGCN, model/layers.py):
    Align :7-23 | CausalConv2d :40-57 | TemporalConvLayer :59-120 | ChebGraphConv :122-172 |
    GraphConv :174-206 | GraphConvLayer :208-231 | STConvBlock :233-258 | OutputBlock :260-284

The sub-layer modules only OWN parameters (that is what fixes the checkpoint keys and the initialisation
order); they have no arithmetic of their own: ``STConvBlock.forward`` / ``OutputBlock.forward`` hand the
parameter pointers to ``stgcn_stblock_forward`` / ``stgcn_outblock_forward``, and calling a sub-layer directly
raises -- there is no eager / ATen path in this package.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.init as init

from . import ops


def _no_eager(name):
    raise NotImplementedError(f"stgcn_amd.layers.{name} only owns parameters: its arithmetic runs inside the fused HIP operators "
                              f"(STConvBlock / OutputBlock); there is no eager PyTorch path")


class DropoutStream:
    """Seed / running offset of the counter-based dropout RNG (Philox4x32-10 inside the kernels).

    Eager mode: every training forward of every block consumes one host-side offset, so masks never
    repeat.  Graph mode (``use_device_counter``): kernel arguments are frozen inside a captured hipGraph, so
    the running part of the offset lives in a device int64 that the captured step bumps (``advance``);
    each block adds its own static site id.  The trainer gives each data-parallel rank its own seed."""
    seed: int = 0x5EED5EED
    rank_offset: int = 0      # added to every seed set through manual_seed (train.init_distributed: the data-parallel rank)
    _offset: int = 0
    _sites: int = 0
    counter = None            # device int64[1] in graph mode
    SITE_STRIDE = 1 << 20     # room for a million dropout sites per step
    CHAIN_STRIDE = 1 << 12    # ... of which each concurrent micro-batch chain (ops.chain_scope) gets 4096

    @classmethod
    def manual_seed(cls, seed: int):
        cls.seed = (int(seed) + cls.rank_offset) & 0xFFFFFFFFFFFFFFFF      # ranks stay decorrelated when user code re-seeds after init
        cls._offset = 0
        if cls.counter is not None:
            cls.counter.zero_()

    @classmethod
    def next_offset(cls) -> int:
        cls._offset += 1
        return cls._offset

    @classmethod
    def new_site(cls) -> int:
        cls._sites += 1
        return cls._sites

    @classmethod
    def use_device_counter(cls, device):
        cls.counter = torch.zeros(1, dtype=torch.int64, device=device)

    @classmethod
    def disable_device_counter(cls):
        cls.counter = None

    @classmethod
    def advance(cls):
        """Call once per training step (inside the captured region in graph mode)."""
        if cls.counter is not None:
            cls.counter.add_(cls.SITE_STRIDE)


class Align(nn.Module):
    """Channel matcher of the residual branches (layers.py:7-23).  The 1x1 conv is always
    allocated, exactly like the reference, even when the pad / identity branch is taken."""

    def __init__(self, c_in, c_out):
        super().__init__()
        self.c_in = c_in
        self.c_out = c_out
        self.align_conv = nn.Conv2d(in_channels=c_in, out_channels=c_out, kernel_size=(1, 1))

    def forward(self, x):
        _no_eager("Align")


class CausalConv2d(nn.Conv2d):
    """layers.py:40-57.  In this model it is always built with enable_padding=False, i.e. a valid
    (Kt x 1) convolution along time."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, enable_padding=False, dilation=1, groups=1, bias=True):
        kernel_size = nn.modules.utils._pair(kernel_size)
        stride = nn.modules.utils._pair(stride)
        dilation = nn.modules.utils._pair(dilation)
        if enable_padding:
            self._left = [int((kernel_size[i] - 1) * dilation[i]) for i in range(len(kernel_size))]
        else:
            self._left = None
        super().__init__(in_channels, out_channels, kernel_size, stride=stride, padding=0, dilation=dilation, groups=groups, bias=bias)

    def forward(self, input):
        _no_eager("CausalConv2d")


class TemporalConvLayer(nn.Module):
    """Gated causal temporal convolution (layers.py:59-120)."""

    def __init__(self, Kt, c_in, c_out, n_vertex, act_func):
        super().__init__()
        self.Kt = Kt
        self.c_in = c_in
        self.c_out = c_out
        self.n_vertex = n_vertex
        self.align = Align(c_in, c_out)
        gated = act_func in ("glu", "gtu")
        self.causal_conv = CausalConv2d(in_channels=c_in, out_channels=2 * c_out if gated else c_out, kernel_size=(Kt, 1),
                                        enable_padding=False, dilation=1)
        self.act_func = act_func

    def forward(self, x):
        _no_eager("TemporalConvLayer")


def _init_graph_weight(weight, bias):
    """layers.py:136-141 / :187-192."""
    init.kaiming_uniform_(weight, a=math.sqrt(5))
    if bias is not None:
        fan_in, _ = init._calculate_fan_in_and_fan_out(weight)
        bound = 1 / math.sqrt(fan_in) if fan_in > 0 else 0
        init.uniform_(bias, -bound, bound)


class ChebGraphConv(nn.Module):
    """layers.py:122-172.  Parameters are created on the CPU like the reference's FloatTensor."""

    def __init__(self, c_in, c_out, Ks, gso, bias):
        super().__init__()
        self.c_in, self.c_out, self.Ks, self.gso = c_in, c_out, Ks, gso
        self.weight = nn.Parameter(torch.empty(Ks, c_in, c_out, dtype=torch.float32, device="cpu"))
        if bias:
            self.bias = nn.Parameter(torch.empty(c_out, dtype=torch.float32, device="cpu"))
        else:
            self.register_parameter("bias", None)
        _init_graph_weight(self.weight, self.bias)

    def forward(self, x):
        _no_eager("ChebGraphConv")


class GraphConv(nn.Module):
    """layers.py:174-206."""

    def __init__(self, c_in, c_out, gso, bias):
        super().__init__()
        self.c_in, self.c_out, self.gso = c_in, c_out, gso
        self.weight = nn.Parameter(torch.empty(c_in, c_out, dtype=torch.float32, device="cpu"))
        if bias:
            self.bias = nn.Parameter(torch.empty(c_out, dtype=torch.float32, device="cpu"))
        else:
            self.register_parameter("bias", None)
        _init_graph_weight(self.weight, self.bias)

    def forward(self, x):
        _no_eager("GraphConv")


class GraphConvLayer(nn.Module):
    """layers.py:208-231."""

    def __init__(self, graph_conv_type, c_in, c_out, Ks, gso, bias):
        super().__init__()
        self.graph_conv_type = graph_conv_type
        self.c_in, self.c_out = c_in, c_out
        self.align = Align(c_in, c_out)
        self.Ks = Ks
        self.gso = gso
        if graph_conv_type == "cheb_graph_conv":
            self.cheb_graph_conv = ChebGraphConv(c_out, c_out, Ks, gso, bias)
        elif graph_conv_type == "graph_conv":
            self.graph_conv = GraphConv(c_out, c_out, gso, bias)

    def forward(self, x):
        _no_eager("GraphConvLayer")


class STConvBlock(nn.Module):
    """'TGTND' block (layers.py:233-258) as one fused HIP operator.

    Constructor signature and parameter tree are the reference's; ``forward`` accepts the logical
    (B, C, T, N) tensor with any strides and returns logical (B, channels[2], T - 2(Kt-1), N).
    """

    def __init__(self, Kt, Ks, n_vertex, last_block_channel, channels, act_func, graph_conv_type, gso, bias, droprate):
        super().__init__()
        self.tmp_conv1 = TemporalConvLayer(Kt, last_block_channel, channels[0], n_vertex, act_func)
        self.graph_conv = GraphConvLayer(graph_conv_type, channels[0], channels[1], Ks, gso, bias)
        self.tmp_conv2 = TemporalConvLayer(Kt, channels[1], channels[2], n_vertex, act_func)
        self.tc2_ln = nn.LayerNorm([n_vertex, channels[2]], eps=1e-12)
        self.relu = nn.ReLU()
        self.dropout = nn.Dropout(p=droprate)
        self.cfg = ops.BlockConfig(Kt=Kt, Ks=Ks, n_vertex=n_vertex, c_in=last_block_channel, channels=tuple(channels),
                                   act_func=act_func, graph_conv_type=graph_conv_type, droprate=float(droprate),
                                   ln_eps=self.tc2_ln.eps)
        self.gso = gso                      # plain attribute like the reference: not in state_dict, not moved by .to()
        self._gso_cache = None              # (key, padded, padded transposed)
        self._ws = ops.WorkspaceCache()
        self._site = DropoutStream.new_site()

    def _operators(self, device):
        gso = self.gso
        key = (gso.data_ptr(), gso._version, str(device), ops.gc_layout_epoch())
        if self._gso_cache is None or self._gso_cache[0] != key:
            gp, gt = ops.gso_prepare(gso.to(device), ops.graph_terms(self.cfg))
            self._gso_cache = (key, gp, gt)
        return self._gso_cache[1], self._gso_cache[2]

    def _params(self):
        gc = self.graph_conv.cheb_graph_conv if self.cfg.graph_conv_type == "cheb_graph_conv" else self.graph_conv.graph_conv
        t1, t2, al = self.tmp_conv1, self.tmp_conv2, self.graph_conv.align.align_conv
        return [t1.causal_conv.weight, t1.causal_conv.bias, t1.align.align_conv.weight, t1.align.align_conv.bias,
                al.weight, al.bias, gc.weight, gc.bias,
                t2.causal_conv.weight, t2.causal_conv.bias, t2.align.align_conv.weight, t2.align.align_conv.bias,
                self.tc2_ln.weight, self.tc2_ln.bias]

    def forward(self, x):
        gp, gt = self._operators(x.device)
        training = self.training and self.cfg.droprate > 0.0
        counter = DropoutStream.counter if training else None
        if counter is not None:
            offset = self._site + DropoutStream.CHAIN_STRIDE * ops.current_chain()   # static per block and chain; the device counter supplies the step
        else:
            offset = DropoutStream.next_offset() if training else 0
        self._last_call = (int(x.shape[0]), int(x.shape[2]), x.dtype)      # (what chain_status needs to find the control words again)
        return ops.st_conv_block(x, gp, gt, self.cfg, self._params(), training, DropoutStream.seed, offset, self._ws, counter)

    def chain_status(self) -> int:
        """0, or 1 + the (b, t) slab whose LayerNorm statistics a workgroup of the last forward gave up waiting for (its outputs are NaN);
        synchronises the stream (``ops.block_chain_status``)."""
        last = getattr(self, "_last_call", None)
        return 0 if last is None else ops.block_chain_status(self.cfg, last[0], last[1], self._ws, dtype=last[2])


class OutputBlock(nn.Module):
    """'TNFF' head (layers.py:260-284) as one fused HIP operator (fwd) + one (bwd) for the reference's
    channel plan ([128, 128] -> 1, which main.py:84-92 hard-wires).  Other channel plans are outside what the kernels cover:
    ``forward`` raises NotImplementedError (there is no eager fallback)."""

    def __init__(self, Ko, last_block_channel, channels, end_channel, n_vertex, act_func, bias, droprate):
        super().__init__()
        self.tmp_conv1 = TemporalConvLayer(Ko, last_block_channel, channels[0], n_vertex, act_func)
        self.fc1 = nn.Linear(in_features=channels[0], out_features=channels[1], bias=bias)
        self.fc2 = nn.Linear(in_features=channels[1], out_features=end_channel, bias=bias)
        self.tc1_ln = nn.LayerNorm([n_vertex, channels[0]], eps=1e-12)
        self.relu = nn.ReLU()
        self.dropout = nn.Dropout(p=droprate)
        self.cfg = ops.HeadConfig(Ko=Ko, n_vertex=n_vertex, c_in=last_block_channel, channels=(channels[0], channels[1]),
                                  end_channel=end_channel, act_func=act_func, droprate=float(droprate), ln_eps=self.tc1_ln.eps)
        self._ws = ops.WorkspaceCache()
        self._site = DropoutStream.new_site()

    def _params(self):
        t = self.tmp_conv1
        return [t.causal_conv.weight, t.causal_conv.bias, t.align.align_conv.weight, t.align.align_conv.bias,
                self.tc1_ln.weight, self.tc1_ln.bias, self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias]

    def forward(self, x):
        if not ops.head_supported(self.cfg):
            raise NotImplementedError(f"OutputBlock channel plan {self.cfg.channels} -> {self.cfg.end_channel} (c_in {self.cfg.c_in}) is not "
                                      f"covered by the fused head kernels (supported: [64|128, 128] -> 1)")
        training = self.training and self.cfg.droprate > 0.0
        counter = DropoutStream.counter if training else None
        if counter is not None:
            offset = self._site + DropoutStream.CHAIN_STRIDE * ops.current_chain()
        else:
            offset = DropoutStream.next_offset() if training else 0
        self._last_call = (int(x.shape[0]), int(x.shape[2]), x.dtype)
        return ops.output_block(x, self.cfg, self._params(), training, DropoutStream.seed, offset, self._ws, counter)

    def chain_status(self) -> int:
        """0, or 1 + the window whose row statistics a tile of the last forward gave up waiting for (its predictions are NaN);
        synchronises the stream (``ops.head_chain_status``)."""
        last = getattr(self, "_last_call", None)
        return 0 if last is None else ops.head_chain_status(self.cfg, last[0], last[1], self._ws, dtype=last[2])
