import os, sys, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from tests.gpu_util import run_block_case
from tests.helpers import real_gso
gso = real_gso("metr_la.cheb_sym_norm_lap")
for blk, (c_in, T) in enumerate(((1, 12), (64, 8))):
    e = run_block_case(c_in, (64, 16, 64), 3, 3, "cheb_graph_conv", "glu", 207, 32, T, True, gso=gso)
    print("X6=" + os.environ.get("STGCN_MFMA_X6", "0"), "block", blk, {k: float("%.3g" % v) for k, v in e.items() if k.startswith(("fwd.", "y")) or k in ("dx",)})
