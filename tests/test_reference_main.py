"""The reference's own main.py, UNMODIFIED, driving the drop-in modules (VERDICT r1 item 5 / INTEGRATION.md section A: "nothing else
changes"): `python main.py ...` is executed twice in a scratch directory, once as is and once with `model.models` resolved to
`stgcn_amd.models` (bound to the CPU emulator of the HIP kernels), and the epoch / test lines it prints are compared.

Build-container test: needs /root/reference (skipped on the GPU box, where it does not exist).  Two deviations, both outside main.py:
the graph is a 20-node synthetic one (script.dataloader.load_adj is patched to serve it -- the reference hard-codes the node counts of
its three datasets, and the emulated kernels need a small graph to finish in seconds) and the speed series is synthetic (the
reference's vel.csv blobs are not in the repo); dropout is switched off with the reference's own --droprate flag
because the two sides draw their masks from different generators."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

DRIVER = r'''
import os, sys, runpy, types
import numpy as np, scipy.sparse as sp
REF, ROOT, mode, NV, EPOCHS = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), sys.argv[5]
sys.path.insert(0, REF)
if mode == "dropin":
    sys.path.insert(0, ROOT)
    from tests.emu_util import bind_emulator
    bind_emulator()
    import stgcn_amd.models as dropin_models
    pkg = types.ModuleType("model")
    pkg.__path__ = []                      # a package whose only member is the drop-in `models`
    pkg.models = dropin_models
    sys.modules["model"] = pkg
    sys.modules["model.models"] = dropin_models
from script import dataloader              # the reference's own data path
def load_adj(dataset_name):                # small synthetic graph instead of the hard-coded 207 / 325 / 228 nodes
    rs = np.random.RandomState(0)
    a = rs.uniform(0.1, 1.0, (NV, NV)) * (rs.uniform(size=(NV, NV)) < 0.4)
    a = np.maximum(a, a.T); np.fill_diagonal(a, 1.0)
    return sp.csc_matrix(a), NV
dataloader.load_adj = load_adj
sys.argv = ["main.py", "--dataset", "pemsd7-m", "--batch_size", "8", "--epochs", EPOCHS, "--droprate", "0", "--n_pred", "3", "--patience", "10"]
runpy.run_path(os.path.join(REF, "main.py"), run_name="__main__")
'''


def _run(tmp, mode, n, epochs):
    env = dict(os.environ, PYTHONPATH="", CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="")
    p = subprocess.run([sys.executable, "-c", DRIVER, REF, ROOT, mode, str(n), str(epochs)], cwd=tmp, env=env, capture_output=True, text=True,
                       timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    ep = [(float(a), float(b)) for a, b in re.findall(r"Train loss: ([0-9.eE+-]+) \| Val loss: ([0-9.eE+-]+)", p.stdout)]
    m = re.search(r"Test loss ([0-9.eE+-]+) \| MAE ([0-9.eE+-]+) \| RMSE ([0-9.eE+-]+) \| WMAPE ([0-9.eE+-]+)", p.stdout)
    assert len(ep) == epochs and m, p.stdout[-2000:]
    return ep, [float(v) for v in m.groups()]


# (rows, nodes, epochs): the default case is two epochs over 130 rows of a 20-node graph (~1 min, most of it in the emulator); the
# 700-row run is opt-in (STGCN_FULL_TESTS=1)
@pytest.mark.skipif(not os.path.isdir(REF), reason="needs the reference checkout (build container only)")
@pytest.mark.parametrize("rows,n,epochs", [(130, 20, 2), pytest.param(700, 20, 2, marks=pytest.mark.full)])
def test_reference_main_py_trains_the_drop_in_modules(tmp_path, rows, n, epochs):
    d = tmp_path / "data" / "pemsd7-m"
    d.mkdir(parents=True)
    rng = np.random.default_rng(0)
    t = np.arange(rows)[:, None]
    vel = np.clip(55 + 10 * np.sin(2 * np.pi * t / 288 + rng.uniform(0, 2 * np.pi, (1, n))) + rng.normal(0, 3, (rows, n)), 0, 80)
    with open(d / "vel.csv", "w") as f:          # header row: main.py:107 / dataloader.py:25 read it as such
        f.write(",".join(str(i) for i in range(n)) + "\n")
        for r in vel:
            f.write(",".join(f"{v:.6f}" for v in r) + "\n")
    ref_ep, ref_test = _run(str(tmp_path), "reference", n, epochs)
    got_ep, got_test = _run(str(tmp_path), "dropin", n, epochs)
    # The two sides differ by rounding (1e-7 per step).  On the 700-row series that is enough to take another branch of the training
    # trajectory around step 5 (losses apart by 1e-6 there, 1e-3 after 50 steps): scaling one element of every initial weight tensor
    # by 1 + 3e-7 sends the stage-per-launch kernels (STGCN_FUSE=0, which otherwise track the reference to 6 digits over all 120
    # steps) onto exactly the same alternate branch the fused kernels take -- hence the wider bound for the long run
    tol = 2e-3 if rows < 400 else 1e-2
    for (rt, rv), (gt, gv) in zip(ref_ep, got_ep):
        assert abs(gt - rt) <= tol * abs(rt) and abs(gv - rv) <= tol * abs(rv), (ref_ep, got_ep)
    for r, g in zip(ref_test, got_test):
        assert abs(g - r) <= tol * abs(r), (ref_test, got_test)
