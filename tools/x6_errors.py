"""Stage errors of both ST blocks at the full C2 size against the fp64 stage oracle, for the product form in force (default: bf16x6 in
tc1_fwd / tc2_ln_fwd / tc1_bwd; STGCN_MFMA_X6=0: fp32 MFMAs everywhere).  Activations: max |diff|; gradients: max |diff| / max |reference|.

    python tools/x6_errors.py ; STGCN_MFMA_X6=0 python tools/x6_errors.py        (on an MI355X; profiles/r6-*_x6_errors.txt)"""
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.gpu_util import run_block_case  # noqa: E402
from tests.helpers import real_gso  # noqa: E402

gso = real_gso("metr_la.cheb_sym_norm_lap")
form = "fp32 MFMA" if os.environ.get("STGCN_MFMA_X6", "1") == "0" else "bf16x6   "
for blk, (c_in, T) in enumerate(((1, 12), (64, 8))):
    e = run_block_case(c_in, (64, 16, 64), 3, 3, "cheb_graph_conv", "glu", 207, 32, T, True, gso=gso)
    print(form, "block", blk, {k: float("%.3g" % v) for k, v in sorted(e.items())})
