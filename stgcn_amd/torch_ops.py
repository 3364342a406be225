"""The ST-Conv block and the output head as PyTorch dispatcher operators (last row: ``stgcn::stblock_fwd`` /
``stgcn::stblock_bwd``, ``stgcn::outblock_fwd`` / ``stgcn::outblock_bwd``).

``stgcn_amd.ops.st_conv_block`` (what ``layers.STConvBlock`` calls) drives the C ABI with extra state of a training step -- workspace
caches, LayerNorm hooks between modules, deferred gradient reductions.  The two operators here are the PLAIN form of the same two entry
points (``stgcn_stblock_forward`` / ``stgcn_stblock_backward``, include/stgcn_hip.h) with tensors in, tensors out, registered with the
dispatcher so that code that works on ``torch.ops`` (custom passes, ``torch.library.opcheck``, exporters) sees the block as one node:

    y, saved, ws = torch.ops.stgcn.stblock_fwd(x_cl, gso_pad, params, cfg, act, gc_type, droprate, training, seed, offset)
    dx, *grads   = torch.ops.stgcn.stblock_bwd(dy, x_cl, gso_t_pad, y, saved, ws, params, cfg, act, gc_type, droprate, training,
                                               seed, offset, need_dx)

``x_cl``: (B, T, N, c_in) contiguous, float32 or bfloat16; ``params``: the 14 tensors of ``_lib.PARAM_FIELDS`` in order, an EMPTY tensor
for a parameter the block does not have (``align_conv`` of a temporal layer whose channels already fit; ``model/layers.py:14-23``);
``cfg`` = [c_in, c0, c1, c2, Kt, Ks, n_vertex]; ``gso_pad`` / ``gso_t_pad`` from ``ops.gso_prepare``.  ``stblock_bwd`` returns 15 tensors
(dx, then one gradient per parameter slot; empty where there is nothing to return).  ``stblock(x_cl, ...)`` below is the differentiable
wrapper over the pair.

Dispatch keys: ``CUDA`` is the HIP library.  ``CPU`` is NOT a fallback: it raises unless the bound library is the CPU emulator of the test
suite (tests/emu), which runs the same kernel sources on host tensors -- the product path needs an MI355X.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Sequence

import torch

from . import _lib, ops
from ._lib import PARAM_FIELDS, StblockGrads, StblockParams

_NS = "stgcn"
_libdef = torch.library.Library(_NS, "DEF")
_libdef.define("stblock_fwd(Tensor x_cl, Tensor gso_pad, Tensor[] params, int[] cfg, str act, str gc_type, float droprate, bool training, "
               "int seed, int offset) -> (Tensor, Tensor, Tensor)")
_libdef.define("stblock_bwd(Tensor dy, Tensor x_cl, Tensor gso_t_pad, Tensor y, Tensor saved, Tensor ws, Tensor[] params, int[] cfg, str act, "
               "str gc_type, float droprate, bool training, int seed, int offset, bool need_dx) -> Tensor[]")


def _cfg(cfg: Sequence[int], act: str, gc_type: str, droprate: float) -> ops.BlockConfig:
    if len(cfg) != 7:
        raise ValueError("cfg = [c_in, c0, c1, c2, Kt, Ks, n_vertex]")
    c_in, c0, c1, c2, Kt, Ks, n = (int(v) for v in cfg)
    return ops.BlockConfig(Kt=Kt, Ks=Ks, n_vertex=n, c_in=c_in, channels=(c0, c1, c2), act_func=act, graph_conv_type=gc_type,
                           droprate=float(droprate))


def _params(params: Sequence[torch.Tensor], dev) -> List:
    if len(params) != len(PARAM_FIELDS):
        raise ValueError(f"params: expected {len(PARAM_FIELDS)} tensors in the order {PARAM_FIELDS}")
    out = []
    for name, p in zip(PARAM_FIELDS, params):
        if p.numel() == 0:
            out.append(None)
            continue
        if p.dtype != torch.float32 or not p.is_contiguous() or p.device != dev:
            raise ValueError(f"params.{name}: expected a contiguous float32 tensor on {dev}")
        out.append(p.detach())
    return out


def _buf(t: torch.Tensor, name: str, dev, dtype, min_numel: int = 0, shape=None) -> torch.Tensor:
    """The C ABI takes raw device addresses: everything it would otherwise find out by faulting is checked here."""
    if t.device != dev or t.dtype != dtype or not t.is_contiguous():
        raise ValueError(f"{name}: expected a contiguous {dtype} tensor on {dev}, got {t.dtype} on {t.device}"
                         f"{'' if t.is_contiguous() else ' (not contiguous)'}")
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise ValueError(f"{name}: expected shape {tuple(shape)}, got {tuple(t.shape)}")
    if t.numel() < min_numel:
        raise ValueError(f"{name}: {t.numel()} elements, the plan of this call needs {min_numel}")
    return t


def _gso_floats(bc: ops.BlockConfig) -> int:
    np_, nm_, ns_, tiled_ = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int32()
    L = _lib.lib()
    L.check(L.dll.stgcn_gso_layout(bc.n_vertex, int(ops.graph_terms(bc)), C.byref(np_), C.byref(nm_), C.byref(ns_), C.byref(tiled_)), "stgcn_gso_layout")
    return int(np_.value) * int(np_.value) * int(nm_.value)


def _fwd(x_cl, gso_pad, params, cfg, act, gc_type, droprate, training, seed, offset):
    ops._check_device(x_cl, "x_cl", activation=True)
    if x_cl.dim() != 4 or not x_cl.is_contiguous():
        raise ValueError("x_cl: expected a contiguous (B, T, N, c_in) tensor")
    L = _lib.lib()
    bc = _cfg(cfg, act, gc_type, droprate)
    B, T, N, c_in = x_cl.shape
    if N != bc.n_vertex or c_in != bc.c_in:
        raise ValueError(f"x_cl is {tuple(x_cl.shape)}, cfg says N={bc.n_vertex}, c_in={bc.c_in}")
    desc = ops.make_desc(bc, B, T, bool(training), True, dtype=x_cl.dtype)
    plan = ops.query_plan(desc)
    _buf(gso_pad, "gso_pad", x_cl.device, torch.float32, _gso_floats(bc))
    ps = _params(params, x_cl.device)
    y = torch.empty(B, plan.T2, N, bc.channels[2], dtype=x_cl.dtype, device=x_cl.device)
    saved = torch.empty(plan.saved_floats, dtype=torch.float32, device=x_cl.device)
    ws = torch.empty(plan.ws_floats, dtype=torch.float32, device=x_cl.device)
    pst = ops._param_struct(StblockParams, ps)
    L.check(L.dll.stgcn_stblock_forward(C.byref(desc), C.byref(pst), x_cl.data_ptr(), gso_pad.data_ptr(), y.data_ptr(), saved.data_ptr(),
                                        ws.data_ptr(), int(seed), int(offset), None, ops._stream_of(x_cl)), "stgcn_stblock_forward")
    return y, saved, ws


def _bwd(dy, x_cl, gso_t_pad, y, saved, ws, params, cfg, act, gc_type, droprate, training, seed, offset, need_dx):
    ops._check_device(x_cl, "x_cl", activation=True)
    L = _lib.lib()
    bc = _cfg(cfg, act, gc_type, droprate)
    if x_cl.dim() != 4 or not x_cl.is_contiguous():
        raise ValueError("x_cl: expected a contiguous (B, T, N, c_in) tensor")
    B, T, N, c_in = x_cl.shape
    if N != bc.n_vertex or c_in != bc.c_in:
        raise ValueError(f"x_cl is {tuple(x_cl.shape)}, cfg says N={bc.n_vertex}, c_in={bc.c_in}")
    desc = ops.make_desc(bc, B, T, bool(training), bool(need_dx), dtype=x_cl.dtype)
    plan = ops.query_plan(ops.make_desc(bc, B, T, bool(training), True, dtype=x_cl.dtype))      # (the sizes stblock_fwd allocated with)
    ps = _params(params, x_cl.device)
    c0, c1, c2 = bc.channels
    dev = x_cl.device
    _buf(gso_t_pad, "gso_t_pad", dev, torch.float32, _gso_floats(bc))
    _buf(y, "y", dev, x_cl.dtype, shape=(B, plan.T2, N, c2))
    _buf(saved, "saved", dev, torch.float32, int(plan.saved_floats))
    _buf(ws, "ws", dev, torch.float32, int(plan.ws_floats))
    if dy.device != dev or tuple(dy.shape) != (B, plan.T2, N, c2):
        raise ValueError(f"dy: expected shape {(B, plan.T2, N, c2)} on {dev}, got {tuple(dy.shape)} on {dy.device}")
    used = {"tc1_aw": c_in > c0, "tc1_ab": c_in > c0, "al_w": c0 > c1, "al_b": c0 > c1, "tc2_aw": c1 > c2, "tc2_ab": c1 > c2}
    grads = [torch.empty_like(p) if (p is not None and used.get(n, True)) else None for n, p in zip(PARAM_FIELDS, ps)]
    dy = dy.contiguous()
    if dy.dtype != x_cl.dtype:
        dy = dy.to(x_cl.dtype)
    dx = torch.empty_like(x_cl) if need_dx else None
    pst = ops._param_struct(StblockParams, ps)
    gst = ops._param_struct(StblockGrads, grads)
    L.check(L.dll.stgcn_stblock_backward_hook(C.byref(desc), C.byref(pst), x_cl.data_ptr(), gso_t_pad.data_ptr(), dy.data_ptr(), y.data_ptr(),
                                              saved.data_ptr(), ws.data_ptr(), C.byref(gst), None if dx is None else dx.data_ptr(),
                                              int(seed), int(offset), None, None, ops._stream_of(x_cl)), "stgcn_stblock_backward")
    # (a fresh empty tensor per absent slot: dispatcher outputs must not alias each other)
    return [dx if dx is not None else x_cl.new_empty(0)] + [g if g is not None else x_cl.new_empty(0, dtype=torch.float32) for g in grads]


def _cpu_guard(fn):
    def impl(*args):
        if not _lib.lib().is_emulator:
            raise RuntimeError("stgcn::* operators have no CPU implementation: the HIP library (MI355X) is the only product path")
        return fn(*args)
    return impl


# ---- OutputBlock (model/layers.py:267-284): stgcn::outblock_fwd / outblock_bwd ----------------------------------------------------------
# out, saved, ws = torch.ops.stgcn.outblock_fwd(x_cl, params, [c_in, c0, c1, end_channel, Ko, n_vertex], act, droprate, training, seed, offset)
# dx, *grads     = torch.ops.stgcn.outblock_bwd(dout, x_cl, saved, ws, params, cfg, act, droprate, training, need_dx)
# params: the 10 tensors of _lib.HEAD_PARAM_FIELDS (empty = absent); out: (B, T - Ko + 1, N) float32.
_libdef.define("outblock_fwd(Tensor x_cl, Tensor[] params, int[] cfg, str act, float droprate, bool training, int seed, int offset) "
               "-> (Tensor, Tensor, Tensor)")
_libdef.define("outblock_bwd(Tensor dout, Tensor x_cl, Tensor saved, Tensor ws, Tensor[] params, int[] cfg, str act, float droprate, "
               "bool training, bool need_dx) -> Tensor[]")


def _head_cfg(cfg: Sequence[int], act: str, droprate: float) -> ops.HeadConfig:
    if len(cfg) != 6:
        raise ValueError("cfg = [c_in, c0, c1, end_channel, Ko, n_vertex]")
    c_in, c0, c1, c_end, Ko, n = (int(v) for v in cfg)
    hc = ops.HeadConfig(Ko=Ko, n_vertex=n, c_in=c_in, channels=(c0, c1), end_channel=c_end, act_func=act, droprate=float(droprate))
    if not ops.head_supported(hc):
        raise ValueError(f"OutputBlock plan {cfg} is not covered by the HIP operator (INTEGRATION.md section B2)")
    return hc


def _head_params(params: Sequence[torch.Tensor], dev) -> List:
    from ._lib import HEAD_PARAM_FIELDS
    if len(params) != len(HEAD_PARAM_FIELDS):
        raise ValueError(f"params: expected {len(HEAD_PARAM_FIELDS)} tensors in the order {HEAD_PARAM_FIELDS}")
    out = []
    for name, p in zip(HEAD_PARAM_FIELDS, params):
        if p.numel() == 0:
            out.append(None)
            continue
        if p.dtype != torch.float32 or not p.is_contiguous() or p.device != dev:
            raise ValueError(f"params.{name}: expected a contiguous float32 tensor on {dev}")
        out.append(p.detach())
    return out


def _head_fwd(x_cl, params, cfg, act, droprate, training, seed, offset):
    ops._check_device(x_cl, "x_cl", activation=True)
    if x_cl.dim() != 4 or not x_cl.is_contiguous():
        raise ValueError("x_cl: expected a contiguous (B, T, N, c_in) tensor")
    L = _lib.lib()
    hc = _head_cfg(cfg, act, droprate)
    B, T, N, c_in = x_cl.shape
    desc = ops.make_head_desc(hc, B, T, bool(training), True, dtype=x_cl.dtype)
    plan = ops.query_head_plan(desc)
    ps = _head_params(params, x_cl.device)
    out = torch.empty(B, plan.T1, N, dtype=torch.float32, device=x_cl.device)
    saved = torch.empty(plan.saved_floats, dtype=torch.float32, device=x_cl.device)
    ws = torch.empty(plan.ws_floats, dtype=torch.float32, device=x_cl.device)
    pst = ops._head_struct(_lib.OutblockParams, ps)
    L.check(L.dll.stgcn_outblock_forward(C.byref(desc), C.byref(pst), x_cl.data_ptr(), out.data_ptr(), saved.data_ptr(), ws.data_ptr(),
                                         int(seed), int(offset), None, ops._stream_of(x_cl)), "stgcn_outblock_forward")
    return out, saved, ws


def _head_bwd(dout, x_cl, saved, ws, params, cfg, act, droprate, training, need_dx):
    from ._lib import HEAD_PARAM_FIELDS
    ops._check_device(x_cl, "x_cl", activation=True)
    L = _lib.lib()
    hc = _head_cfg(cfg, act, droprate)
    if x_cl.dim() != 4 or not x_cl.is_contiguous():
        raise ValueError("x_cl: expected a contiguous (B, T, N, c_in) tensor")
    B, T, N, c_in = x_cl.shape
    desc = ops.make_head_desc(hc, B, T, bool(training), bool(need_dx), dtype=x_cl.dtype)
    hplan = ops.query_head_plan(ops.make_head_desc(hc, B, T, bool(training), True, dtype=x_cl.dtype))
    _buf(saved, "saved", x_cl.device, torch.float32, int(hplan.saved_floats))
    _buf(ws, "ws", x_cl.device, torch.float32, int(hplan.ws_floats))
    if dout.device != x_cl.device or dout.numel() != B * hplan.T1 * N:
        raise ValueError(f"dout: expected {B * hplan.T1 * N} elements on {x_cl.device}, got {dout.numel()} on {dout.device}")
    ps = _head_params(params, x_cl.device)
    used = {"tc_aw": c_in > hc.channels[0], "tc_ab": c_in > hc.channels[0]}
    grads = [torch.empty_like(p) if (p is not None and used.get(n, True)) else None for n, p in zip(HEAD_PARAM_FIELDS, ps)]
    dout = dout.contiguous().float()
    dx = torch.empty_like(x_cl) if need_dx else None
    pst = ops._head_struct(_lib.OutblockParams, ps)
    gst = ops._head_struct(_lib.OutblockGrads, grads)
    L.check(L.dll.stgcn_outblock_backward_hook(C.byref(desc), C.byref(pst), x_cl.data_ptr(), dout.data_ptr(), saved.data_ptr(), ws.data_ptr(),
                                               C.byref(gst), None if dx is None else dx.data_ptr(), None, ops._stream_of(x_cl)),
            "stgcn_outblock_backward")
    return [dx if dx is not None else x_cl.new_empty(0)] + [g if g is not None else x_cl.new_empty(0, dtype=torch.float32) for g in grads]


_libimpl_cuda = torch.library.Library(_NS, "IMPL", "CUDA")
_libimpl_cuda.impl("stblock_fwd", _fwd)
_libimpl_cuda.impl("stblock_bwd", _bwd)
_libimpl_cpu = torch.library.Library(_NS, "IMPL", "CPU")
_libimpl_cpu.impl("stblock_fwd", _cpu_guard(_fwd))
_libimpl_cpu.impl("stblock_bwd", _cpu_guard(_bwd))
_libimpl_cuda.impl("outblock_fwd", _head_fwd)
_libimpl_cuda.impl("outblock_bwd", _head_bwd)
_libimpl_cpu.impl("outblock_fwd", _cpu_guard(_head_fwd))
_libimpl_cpu.impl("outblock_bwd", _cpu_guard(_head_bwd))


# ---- shape-only ("fake") implementations: tracing / export see correctly shaped outputs without a GPU (the plan query is host code) ----
def _fake_fwd(x_cl, gso_pad, params, cfg, act, gc_type, droprate, training, seed, offset):
    bc = _cfg(cfg, act, gc_type, droprate)
    B, T, N, _ = x_cl.shape
    plan = ops.query_plan(ops.make_desc(bc, B, T, bool(training), True, dtype=x_cl.dtype))
    return (x_cl.new_empty((B, int(plan.T2), N, bc.channels[2])), x_cl.new_empty((int(plan.saved_floats),), dtype=torch.float32),
            x_cl.new_empty((int(plan.ws_floats),), dtype=torch.float32))


def _fake_bwd(dy, x_cl, gso_t_pad, y, saved, ws, params, cfg, act, gc_type, droprate, training, seed, offset, need_dx):
    c_in, c0, c1, c2 = (int(v) for v in cfg[:4])
    used = {"tc1_aw": c_in > c0, "tc1_ab": c_in > c0, "al_w": c0 > c1, "al_b": c0 > c1, "tc2_aw": c1 > c2, "tc2_ab": c1 > c2}
    return [torch.empty_like(x_cl) if need_dx else x_cl.new_empty(0)] + [
        torch.empty_like(p) if (p.numel() and used.get(n, True)) else x_cl.new_empty(0, dtype=torch.float32) for n, p in zip(PARAM_FIELDS, params)]


def _fake_head_fwd(x_cl, params, cfg, act, droprate, training, seed, offset):
    hc = _head_cfg(cfg, act, droprate)
    B, T, N, _ = x_cl.shape
    plan = ops.query_head_plan(ops.make_head_desc(hc, B, T, bool(training), True, dtype=x_cl.dtype))
    return (x_cl.new_empty((B, int(plan.T1), N), dtype=torch.float32), x_cl.new_empty((int(plan.saved_floats),), dtype=torch.float32),
            x_cl.new_empty((int(plan.ws_floats),), dtype=torch.float32))


def _fake_head_bwd(dout, x_cl, saved, ws, params, cfg, act, droprate, training, need_dx):
    from ._lib import HEAD_PARAM_FIELDS
    c_in, c0 = int(cfg[0]), int(cfg[1])
    used = {"tc_aw": c_in > c0, "tc_ab": c_in > c0}
    return [torch.empty_like(x_cl) if need_dx else x_cl.new_empty(0)] + [
        torch.empty_like(p) if (p.numel() and used.get(n, True)) else x_cl.new_empty(0, dtype=torch.float32) for n, p in zip(HEAD_PARAM_FIELDS, params)]


for _name, _fn in (("stblock_fwd", _fake_fwd), ("stblock_bwd", _fake_bwd), ("outblock_fwd", _fake_head_fwd), ("outblock_bwd", _fake_head_bwd)):
    torch.library.register_fake(f"{_NS}::{_name}", _fn, lib=_libdef)


class _Block(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x_cl, gso_pad, gso_t_pad, cfg, act, gc_type, droprate, training, seed, offset, *params):
        y, saved, ws = torch.ops.stgcn.stblock_fwd(x_cl, gso_pad, [p.detach() for p in params], cfg, act, gc_type, droprate, training, seed, offset)
        ctx.save_for_backward(x_cl, gso_t_pad, y, saved, ws, *params)
        ctx.meta = (list(cfg), act, gc_type, droprate, training, seed, offset)
        return y

    @staticmethod
    def backward(ctx, dy):
        x_cl, gso_t_pad, y, saved, ws, *params = ctx.saved_tensors
        cfg, act, gc_type, droprate, training, seed, offset = ctx.meta
        out = torch.ops.stgcn.stblock_bwd(dy, x_cl, gso_t_pad, y, saved, ws, [p.detach() for p in params], cfg, act, gc_type, droprate, training,
                                          seed, offset, bool(ctx.needs_input_grad[0]))
        dx = out[0] if out[0].numel() else None
        grads = [g if (g.numel() and ctx.needs_input_grad[10 + i]) else None for i, g in enumerate(out[1:])]
        return (dx, None, None, None, None, None, None, None, None, None, *grads)


def stblock(x_cl: torch.Tensor, gso_pad: torch.Tensor, gso_t_pad: torch.Tensor, params: Sequence[torch.Tensor], cfg: Sequence[int], act: str,
            gc_type: str, droprate: float, training: bool, seed: int = 0, offset: int = 0) -> torch.Tensor:
    """Differentiable ST-Conv block over the two dispatcher operators (channels-last in, channels-last out)."""
    return _Block.apply(x_cl, gso_pad, gso_t_pad, list(cfg), act, gc_type, float(droprate), bool(training), int(seed), int(offset), *params)


class _Head(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x_cl, cfg, act, droprate, training, seed, offset, *params):
        out, saved, ws = torch.ops.stgcn.outblock_fwd(x_cl, [p.detach() for p in params], cfg, act, droprate, training, seed, offset)
        ctx.save_for_backward(x_cl, saved, ws, *params)
        ctx.meta = (list(cfg), act, droprate, training)
        return out

    @staticmethod
    def backward(ctx, dout):
        x_cl, saved, ws, *params = ctx.saved_tensors
        cfg, act, droprate, training = ctx.meta
        res = torch.ops.stgcn.outblock_bwd(dout, x_cl, saved, ws, [p.detach() for p in params], cfg, act, droprate, training,
                                           bool(ctx.needs_input_grad[0]))
        dx = res[0] if res[0].numel() else None
        grads = [g if (g.numel() and ctx.needs_input_grad[7 + i]) else None for i, g in enumerate(res[1:])]
        return (dx, None, None, None, None, None, None, *grads)


def outblock(x_cl: torch.Tensor, params: Sequence[torch.Tensor], cfg: Sequence[int], act: str, droprate: float, training: bool,
             seed: int = 0, offset: int = 0) -> torch.Tensor:
    """Differentiable OutputBlock over the two dispatcher operators: (B, T, N, c_in) channels-last in, (B, T - Ko + 1, N) out."""
    return _Head.apply(x_cl, list(cfg), act, float(droprate), bool(training), int(seed), int(offset), *params)
