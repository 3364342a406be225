"""The emulator's LDS race check (tests/emu/build_emu.py --race, tools/emu_race.sh) over the kernels of one ST block and the head.
Opt-in (STGCN_FULL_TESTS=1): the instrumented build of the one translation unit takes ~9 minutes of compile time."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.full
def test_no_cross_wave_lds_race_in_block_and_head_kernels():
    env = dict(os.environ, STGCN_EMU_RACE="1")
    env.pop("STGCN_EMU_RACE_WARN", None)      # a launch with a cross-wave race aborts the run
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", "tests/test_emu_forward.py", "tests/test_emu_backward.py",
                        "tests/test_emu_head.py", "-k", "17-1-6 or 17-2-6 or 21-2-7 or ranges_cut or head_fwd_bwd"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=3600)
    assert r.returncode == 0 and "emu-race" not in r.stderr, r.stdout[-2000:] + r.stderr[-3000:]
