"""Host side of the fused ST-Conv-block operator: buffer planning, autograd glue, stream plumbing.

``st_conv_block(x, ...)`` is the custom ``autograd.Function`` the north star asks for: one forward call
(`stgcn_stblock_forward`) and one backward call (`stgcn_stblock_backward`) into the HIP library replace
the ~60 ATen launches the reference issues per block per step.

PyTorch is used for device memory (caching allocator), streams and autograd bookkeeping only.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch

from . import _lib
from ._lib import (HEAD_PARAM_FIELDS, PARAM_FIELDS, AdamwHyper, AdamwTensor, FlushBlock, OutblockDesc, OutblockGrads, OutblockParams, OutblockPlan, StblockDesc,
                   StblockGrads, StblockParams, StblockPlan)


@dataclass(frozen=True)
class BlockConfig:
    """Static configuration of one STConvBlock (constructor arguments at model/layers.py:241)."""
    Kt: int
    Ks: int
    n_vertex: int
    c_in: int
    channels: Tuple[int, int, int]
    act_func: str
    graph_conv_type: str
    droprate: float
    ln_eps: float = 1e-12
    tag: int = 0          # block index: labels the library's kernel timer only


def _stream_of(t: torch.Tensor) -> Optional[int]:
    if t.is_cuda:
        return torch.cuda.current_stream(t.device).cuda_stream
    return None


ACT_DTYPES = {torch.float32: _lib.DTYPE_F32, torch.bfloat16: _lib.DTYPE_BF16}      # storage / arithmetic types of the activations


def _check_device(t: torch.Tensor, what: str, activation: bool = False):
    L = _lib.lib()
    if activation:
        if t.dtype not in ACT_DTYPES:
            raise TypeError(f"{what}: expected float32 or bfloat16 activations, got {t.dtype}")
    elif t.dtype != torch.float32:
        raise TypeError(f"{what}: expected float32, got {t.dtype}")
    if L.is_emulator:
        if t.is_cuda:
            raise RuntimeError(f"{what}: the bound library is the CPU emulator but the tensor is on {t.device}")
    elif not t.is_cuda:
        raise RuntimeError(f"{what}: stgcn_amd runs on MI355X only (tensor is on {t.device}); there is no CPU fallback")


_input_index: Dict[int, Tuple[torch.Tensor, int]] = {}


def bind_input_index(base: torch.Tensor, index: torch.Tensor, stride_floats: int) -> None:
    """Device-side batch position: from now on a fused-operator input (or an MSE target) whose storage starts at
    ``base.data_ptr()`` is read ``index[0] * stride_floats`` floats further on, with ``index`` an int64 DEVICE scalar that a
    captured training step advances by itself (``stgcn_prepack`` counters).  Kernel arguments are frozen inside a hipGraph;
    this is how the replayed step walks through a resident series without per-step input copies."""
    assert index.dtype == torch.int64 and index.numel() == 1 and index.device == base.device
    _input_index[base.data_ptr()] = (index, int(stride_floats))


def unbind_input_index(base: torch.Tensor) -> None:
    _input_index.pop(base.data_ptr(), None)


def _index_of(t: torch.Tensor):
    e = _input_index.get(t.data_ptr())
    return (None, 0) if e is None else (e[0].data_ptr(), e[1])


def window_strided_rows(x_p: torch.Tensor) -> Optional[int]:
    """``x_p``: the (B, T, N, C) permutation of a block input.  Returns the window stride in rows of C floats if every window is a
    dense (T, N, C) slab but consecutive windows are NOT back to back (e.g. overlapping windows of one resident series built with
    ``torch.as_strided``), else None (dense input, or a layout that needs a contiguous copy)."""
    B, T, N, Cc = x_p.shape
    sb, st, sn, sc = x_p.stride()
    inner = (Cc == 1 or sc == 1) and sn == Cc and st == N * Cc
    if B > 1 and inner and sb >= 0 and sb % Cc == 0 and sb != T * N * Cc:
        return sb // Cc
    return None


def make_desc(cfg: BlockConfig, B: int, T: int, training: bool, need_dx: bool, prepacked: bool = False, defer: bool = False,
              x_bstride: int = 0, x_index: Optional[int] = None, x_index_stride: int = 0, dtype: torch.dtype = torch.float32) -> StblockDesc:
    if cfg.act_func not in _lib.ACT:
        raise NotImplementedError(f"ERROR: The activation function {cfg.act_func} is not implemented.")  # layers.py:117-118
    if cfg.graph_conv_type not in _lib.GRAPH_CONV:
        raise ValueError(f"unknown graph_conv_type {cfg.graph_conv_type}")
    d = StblockDesc()
    d.B, d.T, d.N, d.c_in = B, T, cfg.n_vertex, cfg.c_in
    d.c0, d.c1, d.c2 = cfg.channels
    d.Kt, d.Ks = cfg.Kt, cfg.Ks
    d.act = _lib.ACT[cfg.act_func]
    d.graph_conv = _lib.GRAPH_CONV[cfg.graph_conv_type]
    d.training = 1 if training else 0
    d.droprate = float(cfg.droprate)
    d.ln_eps = float(cfg.ln_eps)
    d.need_dx = 1 if need_dx else 0
    d.reserved = int(cfg.tag)
    d.prepacked = 1 if prepacked else 0
    d.defer_reduce = 1 if defer else 0
    d.x_bstride, d.x_index_dev, d.x_index_stride = int(x_bstride), x_index, int(x_index_stride)
    d.dtype = ACT_DTYPES[dtype]
    return d


_plan_cache: Dict[tuple, StblockPlan] = {}


def query_plan(desc: StblockDesc) -> StblockPlan:
    key = tuple(getattr(desc, f) for f, _ in StblockDesc._fields_)
    p = _plan_cache.get(key)
    if p is None:
        L = _lib.lib()
        p = StblockPlan()
        L.check(L.dll.stgcn_stblock_plan_query(C.byref(desc), C.byref(p)), "stgcn_stblock_plan_query")
        _plan_cache[key] = p
    return p


def gso_prepare(gso: torch.Tensor, terms: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Once per model (main.py:101-103 upload): the dense (N, N) graph shift operator in the layout the graph-conv kernels
    read (``stgcn_gso_layout``).  Up to 512 nodes: the Chebyshev polynomials T_1 .. T_{terms-1} and their transposes, zero
    padded and in MFMA fragment order; beyond (tiled graph conv): the dense zero-padded operator and its transpose.
    ``terms`` = operator terms of the graph conv including the identity: Ks for ChebGraphConv, 2 for GraphConv
    (``graph_terms``)."""
    L = _lib.lib()
    gso = gso.detach().to(torch.float32).contiguous()
    _check_device(gso, "gso")
    N = gso.shape[0]
    assert gso.shape == (N, N)
    np_, nm_, ns_, tiled_ = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int32()
    L.check(L.dll.stgcn_gso_layout(N, int(terms), C.byref(np_), C.byref(nm_), C.byref(ns_), C.byref(tiled_)), "stgcn_gso_layout")
    NP, nm = int(np_.value), int(nm_.value)
    gp = torch.zeros(nm, NP, NP, dtype=torch.float32, device=gso.device)
    gt = torch.zeros(nm, NP, NP, dtype=torch.float32, device=gso.device)
    scratch = torch.empty(max(int(ns_.value), 1), NP if ns_.value else 1, NP if ns_.value else 1, dtype=torch.float32, device=gso.device)
    L.check(L.dll.stgcn_gso_prepare(gso.data_ptr(), N, int(terms), gp.data_ptr(), gt.data_ptr(), scratch.data_ptr(), _stream_of(gso)),
            "stgcn_gso_prepare")
    if gso.is_cuda:
        torch.cuda.current_stream(gso.device).synchronize()     # scratch is freed on return
    return gp, gt


_gc_layout_epoch = 0      # bumped whenever a knob changes operator / workspace layouts: modules re-prepare their cached operators


def gc_layout_epoch() -> int:
    return _gc_layout_epoch


def set_slab_gc_precision(mode) -> str:
    """Operator products of the slab-resident graph conv (graphs up to 512 nodes): "fp32" (exact fp32 MFMAs) or "bf16x3" (three bf16
    MFMAs per product, fp32-class results); returns the previous mode."""
    names = ["fp32", "bf16x3"]
    m = names.index(mode) if isinstance(mode, str) else int(mode)
    return names[int(_lib.lib().dll.stgcn_set_slab_gc_precision(m))]


def set_bwd_precision(mode) -> str:
    """Matrix products of the backward kernels of fp32 blocks: "fp32" (exact, default) or "bf16x3" (split operands, three bf16 MFMAs per
    product, ~2^-16 relative: inside the 1e-3 gradient bar, an explicit opt-in); returns the previous mode (``stgcn_set_bwd_precision``)."""
    names = ["fp32", "bf16x3"]
    m = names.index(mode) if isinstance(mode, str) else int(mode)
    return names[int(_lib.lib().dll.stgcn_set_bwd_precision(m))]


def set_gc_tiled_min_nodes(n: int) -> int:
    """Graphs with at least ``n`` nodes use the tiled graph conv (default 513); returns the previous threshold.  Operators
    (``gso_prepare``) and plans made under one setting must be used under the same setting (test / tuning knob)."""
    global _gc_layout_epoch
    prev = int(_lib.lib().dll.stgcn_set_gc_tiled_min_nodes(int(n)))
    _plan_cache.clear()
    _gc_layout_epoch += 1
    return prev


def set_tc1_bwd_wgs(n: int) -> int:
    """Workgroups of the fused tmp_conv1 backward (0 = one per CU); returns the previous value.  Changes workspace plans: set it
    before the forward of the step it should apply to (test / tuning knob, ``stgcn_set_tc1_bwd_wgs``)."""
    prev = int(_lib.lib().dll.stgcn_set_tc1_bwd_wgs(int(n)))
    _plan_cache.clear()
    return prev


def prepack_flush() -> None:
    """Launch a weight pack that ``prepack_modules(..., park=True)`` left parked (no-op otherwise): ``models.STGCN*.forward`` calls it when it
    leaves, so that a forward that raised before its first block ran cannot leave pointers to its tensors behind."""
    L = _lib.lib()
    L.check(L.dll.stgcn_prepack_flush(), "stgcn_prepack_flush")


def set_tc2ln_peers(n: int) -> int:
    """Workgroups per (b, t) slab of the fused tmp_conv2 + LayerNorm + dropout forward: 0 = by the device, 1 / 2 / 4 force a form
    (test / tuning knob, ``stgcn_set_tc2ln_peers``); returns the previous value."""
    return int(_lib.lib().dll.stgcn_set_tc2ln_peers(int(n)))


def set_chain_spin_ticks(ticks: int) -> int:
    """Bound of one in-launch wait of the head's one-launch forward, in ticks of the device's 100 MHz clock (default 2 s); negative (test
    setting): bound |ticks| and the first tile of every launch withholds its arrival, so its peers' waits run out for certain.  0 only
    queries.  Returns the previous value (``stgcn_set_chain_spin_ticks``)."""
    return int(_lib.lib().dll.stgcn_set_chain_spin_ticks(int(ticks)))


def set_gemm_big_nt(nt: int) -> int:
    """Force the column extent (32 * nt, nt in {4, 5, 6, 8, 10}) of the big bf16 operator GEMM's tiles, 0 = the grid-rounds heuristic
    (``stgcn_set_gemm_big_nt``; the parity tests run every instance the bs-16 8192-node configuration selects).  Returns the previous value."""
    return int(_lib.lib().dll.stgcn_set_gemm_big_nt(int(nt)))


def set_gc_ld_pad(pad: int) -> int:
    """Row padding (bf16 elements, multiple of 8) of the 16-bit planes of the tiled graph conv (``stgcn_set_gc_ld_pad``);
    returns the previous value.  Operators and plans made under one setting must be used under the same setting."""
    global _gc_layout_epoch
    prev = int(_lib.lib().dll.stgcn_set_gc_ld_pad(int(pad)))
    _plan_cache.clear()
    _gc_layout_epoch += 1
    return prev


def set_debug_stages(on: bool) -> bool:
    """Stage tests only: make the fused kernels also write the intermediates they keep on chip (dZ2 -> plan.ws_dZ2);
    returns the previous setting (``stgcn_set_debug_stages``)."""
    _plan_cache.clear()      # (the debug mode makes the forward store U2 / S2 again: another `saved` layout)
    return bool(_lib.lib().dll.stgcn_set_debug_stages(1 if on else 0))


GC_PRECISION = {"fp32": 0, "bf16x3": 1, "bf16": 2}


def set_gc_precision(mode) -> str:
    """Arithmetic of the operator products of the tiled graph conv (graphs beyond 512 nodes): "fp32" (default, exact fp32
    MFMA), "bf16x3" (split bf16 operands, fp32-class results) or "bf16" (``stgcn_set_gc_precision``).  Returns the previous
    mode's name."""
    code = GC_PRECISION[mode] if isinstance(mode, str) else int(mode)
    prev = int(_lib.lib().dll.stgcn_set_gc_precision(code))
    return {v: k for k, v in GC_PRECISION.items()}[prev]


def graph_terms(cfg: "BlockConfig") -> int:
    return int(cfg.Ks) if cfg.graph_conv_type == "cheb_graph_conv" else 2


def _optr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def dropout_mask(n: int, droprate: float, seed: int, offset: int, device, offset_dev: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Keep-scale (0 or 1/(1-p)) for the first n elements of a block output, as the forward draws it."""
    L = _lib.lib()
    out = torch.empty(n, dtype=torch.float32, device=device)
    L.check(L.dll.stgcn_dropout_mask(out.data_ptr(), n, float(droprate), seed, offset, _optr(offset_dev), _stream_of(out)),
            "stgcn_dropout_mask")
    return out


def mse_loss_and_grad(pred: torch.Tensor, target: torch.Tensor, grad_scale: float = 1.0) -> Tuple[torch.Tensor, torch.Tensor]:
    """``nn.MSELoss()(pred, target)`` (main.py:136/167) and d loss / d pred in one launch (stgcn_mse_loss_grad), outside
    autograd: the trainer calls ``pred.backward(dpred)`` instead of ``loss.backward()`` (main.py:168), which removes the
    five small ATen launches of the loss head.  Returns (loss[1], dpred like pred)."""
    L = _lib.lib()
    p = pred.detach()
    _check_device(p, "pred")
    if p.shape != target.shape or not p.is_contiguous() or not target.is_contiguous() or target.dtype != torch.float32:
        raise ValueError(f"mse_loss_and_grad: contiguous float32 tensors of one shape expected, got {tuple(p.shape)} / {tuple(target.shape)}")
    loss = torch.empty(1, dtype=torch.float32, device=p.device)
    dpred = torch.empty_like(p)
    ti, tis = _index_of(target)
    L.check(L.dll.stgcn_mse_loss_grad(p.data_ptr(), target.data_ptr(), p.numel(), float(grad_scale), loss.data_ptr(), dpred.data_ptr(),
                                      ti, tis, _stream_of(p)), "stgcn_mse_loss_grad")
    return loss, dpred


class _PendingLoss:
    """MSE loss waiting to be formed inside the head's backward (``stgcn_outblock_backward_loss``): ``mse_backward`` posts it, the
    ``_OutBlockFn.backward`` that receives the placeholder gradient takes it."""
    __slots__ = ("placeholder", "pred", "target", "index", "index_stride", "grad_scale", "loss")


_pending_loss: Optional[_PendingLoss] = None
_VIEW_NODES = ("ViewBackward", "UnsafeViewBackward", "ReshapeAliasBackward", "UnsqueezeBackward", "SqueezeBackward", "AliasBackward")


def _head_node_of(pred: torch.Tensor):
    """The fused head's autograd node if ``pred`` is its output seen through shape-only views, else None."""
    fn = pred.grad_fn
    while fn is not None and type(fn).__name__.startswith(_VIEW_NODES):
        nxt = [f for f, _ in fn.next_functions if f is not None]
        if len(nxt) != 1:
            return None
        fn = nxt[0]
    return fn if fn is not None and type(fn).__name__ == "_OutBlockFnBackward" else None


def mse_backward(pred: torch.Tensor, target: torch.Tensor, grad_scale: float = 1.0) -> torch.Tensor:
    """``l = nn.MSELoss()(pred, target); l.backward()`` (main.py:167-168) for a prediction that comes straight out of the fused output
    head: the seed gradient is formed inside the head's fc backward kernel and the loss value comes out of the step's gradient reduction
    (no loss launch at all).  Any other ``pred`` takes the one-launch loss kernel (``mse_loss_and_grad``) followed by
    ``pred.backward(dpred)``.  Returns the loss (shape [1]); with an active gradient sink it is valid after the sink's flush."""
    global _pending_loss
    p = pred.detach()
    if (_head_node_of(pred) is None or not p.is_contiguous() or p.shape != target.shape or not target.is_contiguous()
            or target.dtype != torch.float32 or os.environ.get("STGCN_FUSED_LOSS", "1") == "0"):
        loss, dpred = mse_loss_and_grad(pred, target, grad_scale)
        pred.backward(dpred)
        return loss
    _check_device(p, "pred")
    pl = _PendingLoss()
    pl.placeholder = torch.empty_like(p)            # never written, never read: its address identifies the request
    pl.pred, pl.target, pl.grad_scale = p, target, float(grad_scale)
    pl.index, pl.index_stride = _index_of(target)
    pl.loss = torch.empty(1, dtype=torch.float32, device=p.device)
    _pending_loss = pl
    try:
        pred.backward(pl.placeholder)
        if _pending_loss is not None:
            raise RuntimeError("mse_backward: the output head did not take the fused loss (was its backward run through another path?)")
    finally:
        _pending_loss = None
    return pl.loss


class GradSink:
    """Whole-model gradient flush (``stgcn_grad_flush``).  While a sink is active (``grad_sink_scope``) the fused operators'
    backward calls leave their per-workgroup gradient partials in the module workspaces, write NO parameter gradients and
    return none to autograd; ``flush()`` then reduces the partials of every module of the step in ONE launch straight into
    the sink's persistent gradient buffers (``grads[param]``, e.g. views of one flat data-parallel all-reduce buffer, which
    the trainer also installs as ``param.grad``), optionally applying AdamW to each element as it is produced.  Three
    reduce launches + the optimizer launch of a step become one."""

    def __init__(self, grads: Dict[torch.nn.Parameter, torch.Tensor]):
        self.by_ptr = {p.data_ptr(): g for p, g in grads.items()}
        self.blocks = []          # (desc, grads struct, ws tensor, keep-alive)
        self.head = None
        self.owners = []          # WorkspaceCache objects whose buffers hold partials of this (unflushed) step

    def holds(self, wsc) -> bool:
        return any(o is wsc for o in self.owners)

    def grad_for(self, p: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
        return None if p is None else self.by_ptr.get(p.data_ptr())

    def flush(self, opt_table=None, hyper: Optional[AdamwHyper] = None, stream: Optional[int] = None) -> None:
        L = _lib.lib()
        arr = (FlushBlock * max(len(self.blocks), 1))()
        for i, (desc, gst, ws, _) in enumerate(self.blocks):
            arr[i].desc, arr[i].grads, arr[i].ws = C.pointer(desc), C.pointer(gst), ws.data_ptr()
        hd = hg = hws = None
        if self.head is not None:
            hd, hg, hws = C.byref(self.head[0]), C.byref(self.head[1]), self.head[2].data_ptr()
        n_opt = 0 if opt_table is None else len(opt_table)
        L.check(L.dll.stgcn_grad_flush(len(self.blocks), arr, hd, hg, hws, opt_table, n_opt, None if hyper is None else C.byref(hyper), stream),
                "stgcn_grad_flush")
        self.blocks, self.head = [], None
        for o in self.owners:
            o.pending_sink = None
        self.owners = []


_sink: Optional[GradSink] = None


class grad_sink_scope:
    def __init__(self, sink: Optional[GradSink]):
        self.sink = sink

    def __enter__(self):
        global _sink
        self.prev, _sink = _sink, self.sink
        return self.sink

    def __exit__(self, *exc):
        global _sink
        _sink = self.prev
        return False


_chain = 0


def current_chain() -> int:
    """Index of the micro-batch chain the calling code runs for (``train.chained_fwd_bwd``); 0 outside of it."""
    return _chain


class chain_scope:
    """Everything launched inside belongs to micro-batch chain ``c``: modules use that chain's workspace (chains run
    concurrently on different HIP streams and must not share backward temporaries) and its dropout sub-stream."""

    def __init__(self, c: int):
        self.c = int(c)

    def __enter__(self):
        global _chain
        self.prev, _chain = _chain, self.c
        return self

    def __exit__(self, *exc):
        global _chain
        _chain = self.prev
        return False


class WorkspaceCache:
    """Per-module scratch (`ws` of the C ABI): packed weights written by forward and re-read by the
    backward of the same step, plus backward temporaries.  Re-allocated only when the plan grows.
    One buffer per micro-batch chain (``chain_scope``); ``buf`` is chain 0's."""

    def __init__(self):
        self.bufs: Dict[int, torch.Tensor] = {}
        self.prepacked = False      # one-shot: set by prepack_modules, consumed by the next forward of the owning module
        self.pending_sink = None    # GradSink holding deferred gradient partials of this module that have not been flushed yet

    @property
    def buf(self) -> Optional[torch.Tensor]:
        return self.bufs.get(0)

    def take_prepacked(self) -> bool:
        p, self.prepacked = self.prepacked, False
        return p

    def get(self, n_floats: int, device) -> torch.Tensor:
        if self.pending_sink is not None and self.pending_sink.holds(self):
            # a second forward of the same module (gradient accumulation, a module applied twice) would overwrite the deferred
            # partials of the first call: silent wrong gradients.  The caller must flush the sink between the two uses.
            raise RuntimeError("this module's gradient partials are still waiting in an active GradSink: call sink.flush() before the module "
                               "runs again (one forward/backward per module per flush)")
        b = self.bufs.get(_chain)
        if b is None or b.numel() < n_floats or b.device != device:
            nb = torch.empty(max(n_floats, 1), dtype=torch.float32, device=device)
            if b is not None and b.device == device:
                nb[:b.numel()].copy_(b)     # weights a prepack launch left at the head of the buffer stay valid (offsets do not depend on the size)
            b = self.bufs[_chain] = nb
        return b


class LnHookState:
    """LayerNorm of one ST-block forward call, offered to whichever fused operator consumes that block's output: the consumer's
    backward forms the block's LayerNorm-backward row partials while its input gradient is still on chip (``stgcn_ln_hook``) and marks
    the state ready; the block's own backward then skips its pass over dy -- if the gradient it receives is the very buffer the
    consumer wrote (``dx_ptr``), i.e. nothing else contributed to it."""
    __slots__ = ("hook", "keep", "ready", "dx_ptr", "dx_version", "y_ptr")

    def __init__(self, hook, keep):
        self.hook, self.keep, self.ready, self.dx_ptr, self.dx_version, self.y_ptr = hook, keep, False, None, -1, None

    def mark_ready(self, dx: torch.Tensor) -> None:
        # the producer trusts the partials only for THIS buffer in THIS state: an in-place edit of the gradient between consumer and
        # producer (a tensor hook doing g.mul_(), in-place clipping) bumps the version counter and sends the producer back to its own pass
        self.ready, self.dx_ptr, self.dx_version = True, dx.data_ptr(), dx._version

    def matches(self, dy: torch.Tensor) -> bool:
        return self.ready and self.dx_ptr == dy.data_ptr() and self.dx_version == dy._version


_ln_hooks: "Dict[int, tuple]" = {}      # data_ptr of a block output -> (state, the output tensor itself)
_offer_hook_next = False                # set by st_conv_block for the apply() that follows (grad mode is invisible inside Function.forward)


def _offer_ln_hook(y: torch.Tensor, state: LnHookState) -> None:
    # The entry holds a strong reference to y: while it exists the allocator cannot hand y's address to another tensor, so a later
    # input with this data_ptr IS (a view of) y.  Entries leave when a consumer takes them; outputs nobody consumed through a fused
    # operator (a block used on its own) are evicted oldest-first, so at most 8 outputs are pinned.
    while len(_ln_hooks) >= 8:
        _ln_hooks.pop(next(iter(_ln_hooks)))
    state.y_ptr = y.data_ptr()
    _ln_hooks[state.y_ptr] = (state, y)


def _retire_ln_hook(state: Optional[LnHookState]) -> None:
    """The producing block's backward ran: an offer nobody took (the block's output went to a non-fused consumer) must not keep
    pinning y / saved / ws."""
    if state is not None and state.y_ptr is not None:
        e = _ln_hooks.get(state.y_ptr)
        if e is not None and e[0] is state:
            del _ln_hooks[state.y_ptr]


def clear_ln_hooks() -> None:
    """Drop every pending LayerNorm-hook offer (forwards whose backward never ran)."""
    _ln_hooks.clear()


def _take_ln_hook(x_cl: torch.Tensor) -> Optional[LnHookState]:
    e = _ln_hooks.pop(x_cl.data_ptr(), None)
    if e is None:
        return None
    state, y = e
    if tuple(y.shape) != tuple(x_cl.shape) or y.stride() != x_cl.stride():      # not the whole output (a slice / other view of it)
        return None
    return state


def _param_struct(cls, tensors):
    s = cls()
    for name, t in zip(PARAM_FIELDS, tensors):
        setattr(s, name, None if t is None else t.data_ptr())
    return s


class _STBlockFn(torch.autograd.Function):
    """x_cl: (B, T, N, c_in) contiguous -> y_cl: (B, T2, N, c2) contiguous."""

    @staticmethod
    def forward(ctx, x_cl, gso_pad, gso_t_pad, cfg: BlockConfig, training: bool, seed: int, offset: int, offset_dev,
                wsc: WorkspaceCache, *params):
        L = _lib.lib()
        B, T, N, c_in = x_cl.shape
        need_dx = bool(x_cl.requires_grad)
        bstride = 0 if x_cl.is_contiguous() else window_strided_rows(x_cl)
        assert bstride is not None and not (bstride and need_dx), "st_conv_block hands over dense or window-strided inputs only"
        xi, xis = _index_of(x_cl)
        desc = make_desc(cfg, B, T, training, need_dx, prepacked=wsc.take_prepacked(), x_bstride=bstride, x_index=xi, x_index_stride=xis,
                         dtype=x_cl.dtype)
        plan = query_plan(desc)
        dev = x_cl.device
        ps = [None if p is None else p.detach() for p in params]
        for p in ps:
            if p is not None:
                assert p.is_contiguous() and p.dtype == torch.float32 and p.device == dev
        y = torch.empty(B, plan.T2, N, cfg.channels[2], dtype=x_cl.dtype, device=dev)      # (activations keep the input's type)
        saved = torch.empty(plan.saved_floats, dtype=torch.float32, device=dev)            # (4-byte units whatever the activation type)
        ws = wsc.get(plan.ws_floats, dev)
        pst = _param_struct(StblockParams, ps)
        L.check(L.dll.stgcn_stblock_forward(C.byref(desc), C.byref(pst), x_cl.data_ptr(), gso_pad.data_ptr(), y.data_ptr(),
                                            saved.data_ptr(), ws.data_ptr(), seed, offset, _optr(offset_dev), _stream_of(x_cl)),
                "stgcn_stblock_forward")
        ctx.in_hook = _take_ln_hook(x_cl) if need_dx else None          # LayerNorm of the module that produced x (if it was a fused block)
        ctx.own_hook = None
        global _offer_hook_next
        offer, _offer_hook_next = _offer_hook_next, False
        if offer:       # a later backward is possible: grad mode on and something upstream requires grad (decided in st_conv_block)
            hk = _lib.LnHook()
            L.check(L.dll.stgcn_stblock_ln_hook(C.byref(desc), C.byref(pst), y.data_ptr(), ws.data_ptr(), seed, offset, _optr(offset_dev),
                                                C.byref(hk)), "stgcn_stblock_ln_hook")
            ctx.own_hook = LnHookState(hk, (ws, ps, offset_dev))      # (y itself is pinned by the _ln_hooks entry)
            _offer_ln_hook(y, ctx.own_hook)
        ctx.save_for_backward(x_cl, saved, gso_t_pad, y, *[p for p in params if p is not None])
        ctx.param_present = [p is not None for p in params]
        ctx.cfg, ctx.training, ctx.seed, ctx.offset, ctx.wsc, ctx.ws = cfg, training, seed, offset, wsc, ws
        ctx.offset_dev = offset_dev
        ctx.need_dx = need_dx
        ctx.x_window = (bstride, xi, xis)
        ctx.param_needs_grad = [p is not None and p.requires_grad for p in params]
        return y

    @staticmethod
    def backward(ctx, dy):
        L = _lib.lib()
        x_cl, saved, gso_t_pad, y, *present = ctx.saved_tensors
        it = iter(present)
        params = [next(it) if pr else None for pr in ctx.param_present]
        cfg = ctx.cfg
        B, T, N, c_in = x_cl.shape
        sink = _sink
        desc = make_desc(cfg, B, T, ctx.training, ctx.need_dx, defer=sink is not None, x_bstride=ctx.x_window[0], x_index=ctx.x_window[1],
                         x_index_stride=ctx.x_window[2], dtype=x_cl.dtype)
        dy = dy.contiguous()
        if dy.dtype != x_cl.dtype:
            dy = dy.to(x_cl.dtype)
        own = ctx.own_hook
        _retire_ln_hook(own)
        if own is not None and own.matches(dy):
            desc.dy_rowstats_ready = 1       # the consumer of y wrote this block's LayerNorm-backward row partials with its dx
        plan = query_plan(desc)
        dev = x_cl.device
        c0, c1, c2 = cfg.channels
        # which parameters the forward actually used (reference leaves .grad None for the others)
        used = {"tc1_aw": c_in > c0, "tc1_ab": c_in > c0, "al_w": c0 > c1, "al_b": c0 > c1, "tc2_aw": c1 > c2, "tc2_ab": c1 > c2}
        grads = []
        for name, p, need in zip(PARAM_FIELDS, params, ctx.param_needs_grad):
            if p is None or not need or not used.get(name, True):
                grads.append(None)
            elif sink is not None:
                grads.append(sink.grad_for(p))      # persistent buffer, filled by sink.flush()
                if grads[-1] is None:
                    raise RuntimeError(f"gradient sink has no buffer for parameter {name}")
            else:
                grads.append(torch.empty_like(p))
        dx = torch.empty_like(x_cl) if ctx.need_dx else None
        ws = ctx.ws
        pst = _param_struct(StblockParams, [None if p is None else p.detach() for p in params])
        gst = _param_struct(StblockGrads, grads)
        ih = ctx.in_hook if dx is not None else None
        L.check(L.dll.stgcn_stblock_backward_hook(C.byref(desc), C.byref(pst), x_cl.data_ptr(), gso_t_pad.data_ptr(), dy.data_ptr(),
                                                  y.data_ptr(), saved.data_ptr(), ws.data_ptr(), C.byref(gst),
                                                  None if dx is None else dx.data_ptr(), ctx.seed, ctx.offset, _optr(ctx.offset_dev),
                                                  None if ih is None else C.byref(ih.hook), _stream_of(x_cl)),
                "stgcn_stblock_backward")
        if ih is not None:
            ih.mark_ready(dx)
        if sink is not None:
            sink.blocks.append((desc, gst, ws, (grads, params)))
            sink.owners.append(ctx.wsc)
            ctx.wsc.pending_sink = sink
            grads = [None] * len(grads)
        return (dx, None, None, None, None, None, None, None, None, *grads)


def st_conv_block(x: torch.Tensor, gso_pad: torch.Tensor, gso_t_pad: torch.Tensor, cfg: BlockConfig, params, training: bool,
                  seed: int, offset: int, wsc: WorkspaceCache, offset_dev: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Fused STConvBlock.forward (model/layers.py:250-258).

    ``x`` is logical (B, c_in, T, N) with ARBITRARY strides (NCHW for the first block, channels-last
    for later ones -- SURVEY.md section 3.3); the result is logical (B, c2, T-2(Kt-1), N) with channels-last
    strides, exactly what the reference's own forward returns.  ``params`` follows PARAM_FIELDS order.
    ``offset_dev``: optional 1-element int64 device tensor added to ``offset`` on the device (hipGraph replay).
    """
    _check_device(x, "x", activation=True)
    if x.dim() != 4 or x.shape[1] != cfg.c_in or x.shape[3] != cfg.n_vertex:
        raise ValueError(f"expected input (B, {cfg.c_in}, T, {cfg.n_vertex}), got {tuple(x.shape)}")
    x_cl = x.permute(0, 2, 3, 1)
    if x.requires_grad or window_strided_rows(x_cl) is None:
        x_cl = x_cl.contiguous()                    # no copy when x is already channels-last (or c_in == 1)
    # (else: overlapping windows of a resident series are read in place -- device-side windowing, no 12x replicated tensor)
    global _offer_hook_next
    # the LayerNorm-hook offer pins y / saved / ws until a consumer's or this block's backward takes it: only when a backward can follow
    # (eval / no_grad forwards would otherwise leave stale entries behind)
    _offer_hook_next = torch.is_grad_enabled() and (x.requires_grad or any(p is not None and p.requires_grad for p in params))
    y_cl = _STBlockFn.apply(x_cl, gso_pad, gso_t_pad, cfg, training, seed, offset, offset_dev, wsc, *params)
    return y_cl.permute(0, 3, 1, 2)


def prepack_modules(blocks, head, B: int, device, counters=None, dtype: torch.dtype = torch.float32, park: bool = False) -> None:
    """One pack launch for a whole model step (stgcn_prepack): ``blocks`` is a list of (cfg, T_in, params, wsc) of the ST
    blocks in order, ``head`` is (cfg, T_in, params, wsc) or None.  Marks every workspace so that the module's next forward
    skips its own pack launch.  Parameters only change in optimizer.step(), so this runs once per forward of the model."""
    L = _lib.lib()
    arr = (_lib.PrepackBlock * max(len(blocks), 1))()
    keep = []
    for i, (cfg, T, params, wsc) in enumerate(blocks):
        desc = make_desc(cfg, B, T, True, True, dtype=dtype)
        plan = query_plan(desc)
        ps = [None if p is None else p.detach() for p in params]
        pst = _param_struct(StblockParams, ps)
        ws = wsc.get(plan.ws_floats, device)
        keep += [desc, pst, ps, ws]
        arr[i].desc = C.pointer(desc)
        arr[i].params = C.pointer(pst)
        arr[i].ws = ws.data_ptr()
    hd = hp = hws = None
    if head is not None:
        cfg, T, params, wsc = head
        hdesc = make_head_desc(cfg, B, T, True, True, dtype=dtype)
        hplan = query_head_plan(hdesc)
        hps = [None if p is None else p.detach() for p in params]
        hpst = _head_struct(OutblockParams, hps)
        hws_t = wsc.get(hplan.ws_floats, device)
        keep += [hdesc, hpst, hps, hws_t]
        hd, hp, hws = C.byref(hdesc), C.byref(hpst), hws_t.data_ptr()
    stream = torch.cuda.current_stream(device).cuda_stream if torch.device(device).type == "cuda" else None
    carr = None
    if counters:
        carr = (_lib.StepCounter * len(counters))()
        for i, (t, inc, mod) in enumerate(counters):
            assert t.dtype == torch.int64 and t.numel() == 1
            carr[i].ptr, carr[i].inc, carr[i].mod = t.data_ptr(), int(inc), int(mod)
    # (stgcn_prepack_park: the caller -- models.STGCN*.forward -- runs the first block right after this call; the library then sends the pack
    #  out together with that block's first layer when it is the thin one, or first thing in the next call otherwise)
    fn = L.dll.stgcn_prepack_park if park else L.dll.stgcn_prepack
    L.check(fn(len(blocks), arr, hd, hp, hws, 0 if carr is None else len(counters), carr, stream), "stgcn_prepack")
    for _, _, _, wsc in blocks:
        wsc.prepacked = True
    if head is not None:
        head[3].prepacked = True


# ------------------------------------------------------------------------------------------------ output head
@dataclass(frozen=True)
class HeadConfig:
    """Static configuration of OutputBlock (constructor arguments at model/layers.py:267)."""
    Ko: int
    n_vertex: int
    c_in: int
    channels: Tuple[int, int]
    end_channel: int
    act_func: str
    droprate: float
    ln_eps: float = 1e-12


def head_supported(cfg: HeadConfig) -> bool:
    return (cfg.channels[0] in (64, 128) and cfg.channels[1] == 128 and cfg.end_channel == 1 and cfg.act_func in _lib.ACT
            and (cfg.c_in % 4 == 0 or cfg.Ko * cfg.c_in <= 16))


def make_head_desc(cfg: HeadConfig, B: int, T: int, training: bool, need_dx: bool, prepacked: bool = False, defer: bool = False,
                   dtype: torch.dtype = torch.float32) -> OutblockDesc:
    if cfg.act_func not in _lib.ACT:
        raise NotImplementedError(f"ERROR: The activation function {cfg.act_func} is not implemented.")
    d = OutblockDesc()
    d.B, d.T, d.N, d.c_in = B, T, cfg.n_vertex, cfg.c_in
    d.c0, d.c1 = cfg.channels
    d.c_end, d.Ko = cfg.end_channel, cfg.Ko
    d.act = _lib.ACT[cfg.act_func]
    d.training = 1 if training else 0
    d.droprate, d.ln_eps = float(cfg.droprate), float(cfg.ln_eps)
    d.need_dx = 1 if need_dx else 0
    d.prepacked = 1 if prepacked else 0
    d.defer_reduce = 1 if defer else 0
    d.dtype = ACT_DTYPES[dtype]
    return d


_head_plan_cache: Dict[tuple, OutblockPlan] = {}


def query_head_plan(desc: OutblockDesc) -> OutblockPlan:
    key = tuple(getattr(desc, f) for f, _ in OutblockDesc._fields_)
    p = _head_plan_cache.get(key)
    if p is None:
        L = _lib.lib()
        p = OutblockPlan()
        L.check(L.dll.stgcn_outblock_plan_query(C.byref(desc), C.byref(p)), "stgcn_outblock_plan_query")
        _head_plan_cache[key] = p
    return p


def _head_struct(cls, tensors):
    s = cls()
    for name, t in zip(HEAD_PARAM_FIELDS, tensors):
        setattr(s, name, None if t is None else t.data_ptr())
    return s


class _OutBlockFn(torch.autograd.Function):
    """x_cl: (B, T, N, c_in) contiguous -> out: (B, T1, N)."""

    @staticmethod
    def forward(ctx, x_cl, cfg: HeadConfig, training: bool, seed: int, offset: int, offset_dev, wsc: WorkspaceCache, *params):
        L = _lib.lib()
        B, T, N, c_in = x_cl.shape
        need_dx = bool(x_cl.requires_grad)
        desc = make_head_desc(cfg, B, T, training, need_dx, prepacked=wsc.take_prepacked(), dtype=x_cl.dtype)
        plan = query_head_plan(desc)
        dev = x_cl.device
        ps = [None if p is None else p.detach() for p in params]
        out = torch.empty(B, plan.T1, N, dtype=torch.float32, device=dev)
        saved = torch.empty(plan.saved_floats, dtype=torch.float32, device=dev)
        ws = wsc.get(plan.ws_floats, dev)
        pst = _head_struct(OutblockParams, ps)
        L.check(L.dll.stgcn_outblock_forward(C.byref(desc), C.byref(pst), x_cl.data_ptr(), out.data_ptr(), saved.data_ptr(), ws.data_ptr(),
                                             seed, offset, _optr(offset_dev), _stream_of(x_cl)), "stgcn_outblock_forward")
        ctx.in_hook = _take_ln_hook(x_cl) if need_dx else None
        ctx.save_for_backward(x_cl, saved, *[p for p in params if p is not None])
        ctx.param_present = [p is not None for p in params]
        ctx.param_needs_grad = [p is not None and p.requires_grad for p in params]
        ctx.cfg, ctx.training, ctx.ws, ctx.need_dx, ctx.wsc = cfg, training, ws, need_dx, wsc
        return out

    @staticmethod
    def backward(ctx, dout):
        L = _lib.lib()
        x_cl, saved, *present = ctx.saved_tensors
        it = iter(present)
        params = [next(it) if pr else None for pr in ctx.param_present]
        cfg = ctx.cfg
        B, T, N, c_in = x_cl.shape
        sink = _sink
        desc = make_head_desc(cfg, B, T, ctx.training, ctx.need_dx, defer=sink is not None, dtype=x_cl.dtype)
        dout = dout.contiguous()
        if dout.dtype != torch.float32:
            dout = dout.float()
        global _pending_loss
        fused_loss = _pending_loss if (_pending_loss is not None and _pending_loss.placeholder.data_ptr() == dout.data_ptr()) else None
        if fused_loss is not None:
            _pending_loss = None
        used = {"tc_aw": c_in > cfg.channels[0], "tc_ab": c_in > cfg.channels[0]}
        grads = []
        for name, p, need in zip(HEAD_PARAM_FIELDS, params, ctx.param_needs_grad):
            if not (p is not None and need and used.get(name, True)):
                grads.append(None)
            elif sink is not None:
                grads.append(sink.grad_for(p))
                if grads[-1] is None:
                    raise RuntimeError(f"gradient sink has no buffer for head parameter {name}")
            else:
                grads.append(torch.empty_like(p))
        dx = torch.empty_like(x_cl) if ctx.need_dx else None
        pst = _head_struct(OutblockParams, [None if p is None else p.detach() for p in params])
        gst = _head_struct(OutblockGrads, grads)
        ih = ctx.in_hook if dx is not None else None
        if fused_loss is not None:
            gst.loss = fused_loss.loss.data_ptr()
            hl = _lib.HeadLoss()
            hl.pred, hl.target = fused_loss.pred.data_ptr(), fused_loss.target.data_ptr()
            hl.target_index_dev, hl.target_index_stride, hl.grad_scale = fused_loss.index, fused_loss.index_stride, fused_loss.grad_scale
            ctx.keep_loss = fused_loss                  # the buffers outlive a deferred reduction
            L.check(L.dll.stgcn_outblock_backward_loss(C.byref(desc), C.byref(pst), x_cl.data_ptr(), C.byref(hl), saved.data_ptr(),
                                                       ctx.ws.data_ptr(), C.byref(gst), None if dx is None else dx.data_ptr(),
                                                       None if ih is None else C.byref(ih.hook), _stream_of(x_cl)),
                    "stgcn_outblock_backward_loss")
        else:
            L.check(L.dll.stgcn_outblock_backward_hook(C.byref(desc), C.byref(pst), x_cl.data_ptr(), dout.data_ptr(), saved.data_ptr(),
                                                       ctx.ws.data_ptr(), C.byref(gst), None if dx is None else dx.data_ptr(),
                                                       None if ih is None else C.byref(ih.hook), _stream_of(x_cl)),
                    "stgcn_outblock_backward")
        if ih is not None:
            ih.mark_ready(dx)
        if sink is not None:
            sink.head = (desc, gst, ctx.ws, (grads, params))
            sink.owners.append(ctx.wsc)
            ctx.wsc.pending_sink = sink
            grads = [None] * len(grads)
        return (dx, None, None, None, None, None, None, *grads)


def block_chain_status(cfg: BlockConfig, B: int, T: int, wsc: WorkspaceCache, dtype=torch.float32) -> int:
    """Sticky word of an ST block's forward (``stgcn_stblock_chain_status``): 0 if every in-launch wait of the last forward on ``wsc``
    completed, else 1 + the (b, t) slab of tmp_conv2 + LayerNorm whose peer statistics a workgroup gave up waiting for (that part's outputs
    are NaN).  Synchronises the current stream."""
    L = _lib.lib()
    ws = wsc.buf
    if ws is None:
        return 0
    desc = make_desc(cfg, B, T, False, False, dtype=dtype)
    w = C.c_uint32(0)
    L.check(L.dll.stgcn_stblock_chain_status(C.byref(desc), ws.data_ptr(), C.byref(w), _stream_of(ws)), "stgcn_stblock_chain_status")
    return int(w.value)


def head_chain_status(cfg: HeadConfig, B: int, T: int, wsc: WorkspaceCache, dtype=torch.float32) -> int:
    """Sticky word of the head's one-launch forward (``stgcn_outblock_chain_status``): 0 if every in-launch wait of the last forward on
    ``wsc`` completed, else 1 + the window whose row statistics a tile gave up waiting for (that tile's predictions are NaN).
    Synchronises the current stream."""
    L = _lib.lib()
    ws = wsc.buf
    if ws is None:
        return 0
    desc = make_head_desc(cfg, B, T, False, False, dtype=dtype)
    w = C.c_uint32(0)
    L.check(L.dll.stgcn_outblock_chain_status(C.byref(desc), ws.data_ptr(), C.byref(w), _stream_of(ws)), "stgcn_outblock_chain_status")
    return int(w.value)


def output_block(x: torch.Tensor, cfg: HeadConfig, params, training: bool, seed: int, offset: int, wsc: WorkspaceCache,
                 offset_dev: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Fused OutputBlock.forward (model/layers.py:276-284): logical (B, c_in, T, N) -> (B, 1, T-Ko+1, N).
    ``params`` follows HEAD_PARAM_FIELDS order."""
    _check_device(x, "x", activation=True)
    if x.dim() != 4 or x.shape[1] != cfg.c_in or x.shape[3] != cfg.n_vertex:
        raise ValueError(f"expected input (B, {cfg.c_in}, T, {cfg.n_vertex}), got {tuple(x.shape)}")
    x_cl = x.permute(0, 2, 3, 1).contiguous()
    out = _OutBlockFn.apply(x_cl, cfg, training, seed, offset, offset_dev, wsc, *params)
    return out.unsqueeze(1)        # (B, 1, T1, N) == fc2(x).permute(0, 3, 1, 2) with end_channel 1
