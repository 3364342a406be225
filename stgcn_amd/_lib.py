"""ctypes binding of libstgcn_hip.so (C ABI declared in include/stgcn_hip.h).

The product path has exactly one backend: the HIP library built for gfx950 by
``__graft_entry__.build()`` / ``stgcn_amd/build.py``.  If it is missing, importing the ops raises --
there is no CPU or eager-PyTorch fallback.  (``use_library`` exists so the test-suite can point the
same host code at the CPU-emulated twin built from the same sources, tests/emu/.)
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(_HERE, "libstgcn_hip.so")

STGCN_OK = 0
ABI_VERSION = 6      # include/stgcn_hip.h: STGCN_ABI_VERSION (tests/test_capi_symbols.py keeps the two equal)
ACT = {"glu": 0, "gtu": 1}
GRAPH_CONV = {"cheb_graph_conv": 0, "graph_conv": 1}
DTYPE_F32, DTYPE_BF16 = 0, 1

_fp = C.POINTER(C.c_float)


class StblockDesc(C.Structure):
    _fields_ = [("B", C.c_int32), ("T", C.c_int32), ("N", C.c_int32), ("c_in", C.c_int32),
                ("c0", C.c_int32), ("c1", C.c_int32), ("c2", C.c_int32), ("Kt", C.c_int32), ("Ks", C.c_int32),
                ("act", C.c_int32), ("graph_conv", C.c_int32), ("training", C.c_int32),
                ("droprate", C.c_float), ("ln_eps", C.c_float), ("need_dx", C.c_int32), ("reserved", C.c_int32),
                ("prepacked", C.c_int32), ("defer_reduce", C.c_int32),
                ("x_bstride", C.c_int64), ("x_index_dev", C.c_void_p), ("x_index_stride", C.c_int64),
                ("dy_rowstats_ready", C.c_int32), ("dtype", C.c_int32)]


class LnHook(C.Structure):            # stgcn_ln_hook
    _fields_ = [("rowstat", C.c_void_p), ("y", C.c_void_p), ("gamma", C.c_void_p), ("beta", C.c_void_p),
                ("N", C.c_int32), ("C", C.c_int32), ("reserved", C.c_int32), ("training", C.c_int32), ("droprate", C.c_float), ("dtype", C.c_int32),
                ("seed", C.c_uint64), ("offset", C.c_uint64), ("offset_dev", C.c_void_p)]


PARAM_FIELDS = ["tc1_w", "tc1_b", "tc1_aw", "tc1_ab", "al_w", "al_b", "gc_w", "gc_b",
                "tc2_w", "tc2_b", "tc2_aw", "tc2_ab", "ln_w", "ln_b"]


class StblockParams(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in PARAM_FIELDS]


class StblockGrads(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in PARAM_FIELDS]


PLAN_FIELDS = ["T1", "T2", "rows1", "rows2", "NP", "y_floats", "saved_floats", "ws_floats",
               "sv_U1", "sv_S1", "sv_A", "sv_Xk", "sv_G", "sv_U2", "sv_S2", "sv_mean", "sv_rstd", "sv_rowstat",
               "ws_W1p", "ws_W1d", "ws_b1", "ws_Wap", "ws_WaT", "ws_ba", "ws_W2p", "ws_W2d", "ws_b2", "ws_W1dense", "recompute_tc1", "ws_WaDense", "thin_tc1",
               "ws_rowstat_b", "ws_dZ2", "ws_dYg", "ws_dA", "ws_dZ1", "ws_part", "part_floats",
               "tiled_gc", "ws_Gk", "ws_XT", "fused_tc2_bwd", "ws_W2dense", "fused_tc1_bwd", "stored_US2", "ws_chain", "chain_words"]


class StblockPlan(C.Structure):
    _fields_ = [(n, C.c_int64) for n in PLAN_FIELDS]


class OutblockDesc(C.Structure):
    _fields_ = [("B", C.c_int32), ("T", C.c_int32), ("N", C.c_int32), ("c_in", C.c_int32), ("c0", C.c_int32), ("c1", C.c_int32),
                ("c_end", C.c_int32), ("Ko", C.c_int32), ("act", C.c_int32), ("training", C.c_int32), ("droprate", C.c_float),
                ("ln_eps", C.c_float), ("need_dx", C.c_int32), ("dtype", C.c_int32), ("prepacked", C.c_int32),
                ("defer_reduce", C.c_int32)]


HEAD_PARAM_FIELDS = ["tc_w", "tc_b", "tc_aw", "tc_ab", "ln_w", "ln_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b"]


class OutblockParams(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in HEAD_PARAM_FIELDS]


class PrepackBlock(C.Structure):      # stgcn_prepack_block
    _fields_ = [("desc", C.POINTER(StblockDesc)), ("params", C.POINTER(StblockParams)), ("ws", C.c_void_p)]


class OutblockGrads(C.Structure):      # stgcn_outblock_grads: the parameter gradients, then the fused loss value (NULL unless used)
    _fields_ = [(n, C.c_void_p) for n in HEAD_PARAM_FIELDS] + [("loss", C.c_void_p)]


class HeadLoss(C.Structure):           # stgcn_head_loss
    _fields_ = [("pred", C.c_void_p), ("target", C.c_void_p), ("target_index_dev", C.c_void_p), ("target_index_stride", C.c_int64),
                ("grad_scale", C.c_float), ("reserved", C.c_int32)]


HEAD_PLAN_FIELDS = ["T1", "rows", "rows_in", "out_floats", "saved_floats", "ws_floats", "sv_U", "sv_S", "sv_mean", "sv_rstd", "sv_yln",
                    "sv_hd", "sv_rowstat", "ws_Wp", "ws_Wd", "ws_b", "ws_W1p", "ws_W1d", "ws_rowstat_b", "ws_dh1", "ws_dyln", "ws_dZ", "ws_part",
                    "part_floats", "ws_chain", "chain_words"]


class OutblockPlan(C.Structure):
    _fields_ = [(n, C.c_int64) for n in HEAD_PLAN_FIELDS]


class AdamwTensor(C.Structure):
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p), ("numel", C.c_int64)]


class StepCounter(C.Structure):       # stgcn_step_counter
    _fields_ = [("ptr", C.c_void_p), ("inc", C.c_int64), ("mod", C.c_int64)]


class FlushBlock(C.Structure):        # stgcn_flush_block
    _fields_ = [("desc", C.POINTER(StblockDesc)), ("grads", C.POINTER(StblockGrads)), ("ws", C.c_void_p)]


class AdamwHyper(C.Structure):        # stgcn_adamw_hyper
    _fields_ = [("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float), ("weight_decay", C.c_float),
                ("step", C.c_int64), ("step_dev", C.c_void_p), ("lr_dev", C.c_void_p)]


class StgcnError(RuntimeError):
    pass


class _Lib:
    def __init__(self, path: str):
        if not os.path.exists(path):
            raise StgcnError(
                f"{path} not found: the HIP extension has not been built. Run `python -c 'import __graft_entry__ as g; "
                f"g.build()'` (or `python -m stgcn_amd.build`). There is no fallback path.")
        self.path = path
        self.dll = C.CDLL(path)
        d = self.dll
        d.stgcn_version.restype = C.c_int
        # The argument lists below are those of include/stgcn_hip.h at STGCN_ABI_VERSION: a library built from another revision of the
        # header (a stale .so left in the tree, an external build) would take shifted arguments and corrupt device memory -- refuse it.
        got = int(d.stgcn_version())
        if got != ABI_VERSION:
            raise StgcnError(f"{path} was built for ABI revision {got}, this binding needs {ABI_VERSION} (include/stgcn_hip.h "
                             f"STGCN_ABI_VERSION): rebuild with `python -m stgcn_amd.build --force`")
        d.stgcn_backend.restype = C.c_char_p
        d.stgcn_last_error.restype = C.c_char_p
        d.stgcn_stblock_plan_query.argtypes = [C.POINTER(StblockDesc), C.POINTER(StblockPlan)]
        d.stgcn_gso_prepare.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        d.stgcn_gso_layout.argtypes = [C.c_int32, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                       C.POINTER(C.c_int32)]
        d.stgcn_gso_layout.restype = C.c_int
        d.stgcn_set_gc_tiled_min_nodes.argtypes = [C.c_int32]
        d.stgcn_set_gc_tiled_min_nodes.restype = C.c_int
        d.stgcn_outblock_backward_loss.argtypes = [C.POINTER(OutblockDesc), C.POINTER(OutblockParams), C.c_void_p, C.POINTER(HeadLoss), C.c_void_p,
                                                   C.c_void_p, C.POINTER(OutblockGrads), C.c_void_p, C.POINTER(LnHook), C.c_void_p]
        d.stgcn_outblock_backward_loss.restype = C.c_int
        d.stgcn_set_bwd_precision.argtypes = [C.c_int32]
        d.stgcn_set_bwd_precision.restype = C.c_int
        d.stgcn_set_slab_gc_precision.argtypes = [C.c_int32]
        d.stgcn_set_slab_gc_precision.restype = C.c_int
        d.stgcn_set_tc1_bwd_wgs.argtypes = [C.c_int32]
        d.stgcn_set_tc1_bwd_wgs.restype = C.c_int
        d.stgcn_set_tc2ln_peers.argtypes = [C.c_int32]
        d.stgcn_set_tc2ln_peers.restype = C.c_int
        d.stgcn_stblock_chain_status.argtypes = [C.POINTER(StblockDesc), C.c_void_p, C.POINTER(C.c_uint32), C.c_void_p]
        d.stgcn_stblock_chain_status.restype = C.c_int
        d.stgcn_set_debug_stages.argtypes = [C.c_int32]
        d.stgcn_set_debug_stages.restype = C.c_int
        d.stgcn_set_gc_precision.argtypes = [C.c_int32]
        d.stgcn_set_gc_precision.restype = C.c_int
        d.stgcn_set_gc_ld_pad.argtypes = [C.c_int32]
        d.stgcn_set_gc_ld_pad.restype = C.c_int
        d.stgcn_set_gemm_big_nt.argtypes = [C.c_int32]
        d.stgcn_set_gemm_big_nt.restype = C.c_int
        d.stgcn_set_chain_spin_ticks.argtypes = [C.c_int64]
        d.stgcn_set_chain_spin_ticks.restype = C.c_int64
        d.stgcn_outblock_chain_status.argtypes = [C.POINTER(OutblockDesc), C.c_void_p, C.POINTER(C.c_uint32), C.c_void_p]
        d.stgcn_outblock_chain_status.restype = C.c_int
        d.stgcn_stblock_forward.argtypes = [C.POINTER(StblockDesc), C.POINTER(StblockParams), C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p]
        d.stgcn_stblock_backward.argtypes = [C.POINTER(StblockDesc), C.POINTER(StblockParams), C.c_void_p, C.c_void_p,
                                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(StblockGrads), C.c_void_p,
                                             C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p]
        d.stgcn_stblock_backward_hook.argtypes = [C.POINTER(StblockDesc), C.POINTER(StblockParams), C.c_void_p, C.c_void_p,
                                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(StblockGrads), C.c_void_p,
                                                  C.c_uint64, C.c_uint64, C.c_void_p, C.POINTER(LnHook), C.c_void_p]
        d.stgcn_stblock_ln_hook.argtypes = [C.POINTER(StblockDesc), C.POINTER(StblockParams), C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64,
                                            C.c_void_p, C.POINTER(LnHook)]
        d.stgcn_outblock_backward_hook.argtypes = [C.POINTER(OutblockDesc), C.POINTER(OutblockParams), C.c_void_p, C.c_void_p, C.c_void_p,
                                                   C.c_void_p, C.POINTER(OutblockGrads), C.c_void_p, C.POINTER(LnHook), C.c_void_p]
        for f in ("stgcn_stblock_backward_hook", "stgcn_stblock_ln_hook", "stgcn_outblock_backward_hook"):
            getattr(d, f).restype = C.c_int
        d.stgcn_dropout_mask.argtypes = [C.c_void_p, C.c_int64, C.c_float, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p]
        d.stgcn_outblock_plan_query.argtypes = [C.POINTER(OutblockDesc), C.POINTER(OutblockPlan)]
        d.stgcn_outblock_forward.argtypes = [C.POINTER(OutblockDesc), C.POINTER(OutblockParams), C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p]
        d.stgcn_outblock_backward.argtypes = [C.POINTER(OutblockDesc), C.POINTER(OutblockParams), C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.c_void_p, C.POINTER(OutblockGrads), C.c_void_p, C.c_void_p]
        for f in ("stgcn_outblock_plan_query", "stgcn_outblock_forward", "stgcn_outblock_backward"):
            getattr(d, f).restype = C.c_int
        d.stgcn_adamw_step.argtypes = [C.POINTER(AdamwTensor), C.c_int32, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int64,
                                       C.c_void_p, C.c_void_p, C.c_void_p]
        d.stgcn_adamw_step.restype = C.c_int
        d.stgcn_prepack.argtypes = [C.c_int32, C.POINTER(PrepackBlock), C.POINTER(OutblockDesc), C.POINTER(OutblockParams), C.c_void_p,
                                    C.c_int32, C.POINTER(StepCounter), C.c_void_p]
        d.stgcn_prepack.restype = C.c_int
        d.stgcn_prepack_park.argtypes = [C.c_int32, C.POINTER(PrepackBlock), C.POINTER(OutblockDesc), C.POINTER(OutblockParams), C.c_void_p,
                                    C.c_int32, C.POINTER(StepCounter), C.c_void_p]
        d.stgcn_prepack_park.restype = C.c_int
        d.stgcn_prepack_flush.argtypes = []
        d.stgcn_prepack_flush.restype = C.c_int
        d.stgcn_grad_flush.argtypes = [C.c_int32, C.POINTER(FlushBlock), C.POINTER(OutblockDesc), C.POINTER(OutblockGrads), C.c_void_p,
                                       C.POINTER(AdamwTensor), C.c_int32, C.POINTER(AdamwHyper), C.c_void_p]
        d.stgcn_grad_flush.restype = C.c_int
        d.stgcn_mse_loss_grad.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                          C.c_void_p]
        d.stgcn_mse_loss_grad.restype = C.c_int
        d.stgcn_profile_enable.argtypes = [C.c_int]
        d.stgcn_profile_enable.restype = C.c_int
        d.stgcn_profile_collect.argtypes = [C.c_char_p, C.c_size_t]
        d.stgcn_profile_collect.restype = C.c_int
        for f in ("stgcn_stblock_plan_query", "stgcn_gso_prepare", "stgcn_stblock_forward", "stgcn_stblock_backward",
                  "stgcn_dropout_mask"):
            getattr(d, f).restype = C.c_int
        self.backend = d.stgcn_backend().decode()
        self.is_emulator = self.backend.startswith("emu")

    def check(self, rc: int, what: str):
        if rc != STGCN_OK:
            msg = self.dll.stgcn_last_error().decode()
            if rc == 1:
                raise NotImplementedError(f"{what}: {msg}")
            if rc == 2:
                raise ValueError(f"{what}: {msg}")
            raise StgcnError(f"{what}: {msg} (code {rc})")


_lib: Optional[_Lib] = None


def use_library(path: str) -> _Lib:
    """Bind an explicit shared library (tests use this for the emulated twin)."""
    global _lib
    _lib = _Lib(path)
    return _lib


def lib() -> _Lib:
    """The bound library; binds the product HIP library on first use (raises if it is not built)."""
    global _lib
    if _lib is None:
        _lib = _Lib(os.environ.get("STGCN_AMD_LIB", DEFAULT_LIB))
    return _lib


EXPORTED_SYMBOLS = ["stgcn_version", "stgcn_backend", "stgcn_last_error", "stgcn_stblock_plan_query", "stgcn_gso_prepare",
                    "stgcn_stblock_forward", "stgcn_stblock_backward", "stgcn_dropout_mask", "stgcn_profile_enable",
                    "stgcn_profile_collect", "stgcn_outblock_plan_query", "stgcn_outblock_forward", "stgcn_outblock_backward", "stgcn_adamw_step", "stgcn_prepack",
                    "stgcn_mse_loss_grad", "stgcn_grad_flush", "stgcn_gso_layout", "stgcn_set_gc_tiled_min_nodes",
                    "stgcn_set_gc_precision", "stgcn_set_gc_ld_pad", "stgcn_set_debug_stages",
                    "stgcn_stblock_ln_hook", "stgcn_stblock_backward_hook", "stgcn_outblock_backward_hook", "stgcn_set_tc1_bwd_wgs",
                    "stgcn_set_slab_gc_precision", "stgcn_outblock_backward_loss", "stgcn_set_bwd_precision", "stgcn_set_gemm_big_nt",
                    "stgcn_set_chain_spin_ticks", "stgcn_outblock_chain_status", "stgcn_set_tc2ln_peers", "stgcn_stblock_chain_status", "stgcn_prepack_park", "stgcn_prepack_flush"]
