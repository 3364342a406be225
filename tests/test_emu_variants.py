"""Launch-time kernel variants that the residency heuristic may pick on a GPU (tile rows of the gated conv, wave count of
the transposed conv) are forced one by one and run through the same emulator stage tests: on the CPU emulator every
variant "fits", so without forcing only the smallest tile would ever be exercised here."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FWD = "tests/test_emu_forward.py"
BWD = "tests/test_emu_backward.py"


def run_subset(env_extra, files, k):
    env = dict(os.environ, **env_extra)
    r = subprocess.run([sys.executable, "-m", "pytest", *files, "-x", "-q", "-p", "no:cacheprovider", "-k", k],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


@pytest.mark.parametrize("tr", [32, 48, 64])
def test_gated_conv_tile_rows(tr):
    # 1-channel first layer (scalar staging), 64-channel layers with / without an Align epilogue, ragged last tile
    run_subset({"STGCN_TCONV_TR": str(tr)}, [FWD], "21-2-7 or 17-1-6 or 35-1-5")


@pytest.mark.parametrize("waves", [4, 8])
def test_transposed_conv_wave_variants(waves):
    run_subset({"STGCN_BWD_DATA_WAVES": str(waves)}, [BWD], "17-2-6")


@pytest.mark.parametrize("parts", ["1,1", "2,3", "4,2"])
def test_graph_conv_slab_parts(parts):
    # workgroups per (b, t) slab of the graph conv, forward / backward (Chebyshev Ks = 3 and 5, Kipf, 300-node graph)
    run_subset({"STGCN_GC_PARTS": parts}, [FWD], "17-1-6 or 35-1-5 or 9-2-5 or 300-6-12")
    run_subset({"STGCN_GC_PARTS": parts}, [BWD], "17-2-6 or 35-1-5 or 9-2-5")
