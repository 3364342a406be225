"""stgcn_amd -- MI355X-native (gfx950) training path for the STGCN ST-Conv block.

Drop-in for the hot path of hazdzz/STGCN: ``from stgcn_amd import layers, models`` mirrors the
reference's ``from model import layers, models``.  All arithmetic of the ST blocks runs in hand-written
HIP kernels behind a C ABI (include/stgcn_hip.h); there is no CPU / eager fallback.
"""
from . import layers, models, ops, optim  # noqa: F401
from .layers import DropoutStream  # noqa: F401

__all__ = ["layers", "models", "ops", "optim", "DropoutStream"]
