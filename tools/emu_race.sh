#!/bin/bash
# The CPU-emulated kernels under the emulator's LDS race check (tests/emu/build_emu.py --race): every LDS access of every kernel is
# checked for a conflicting access by ANOTHER WAVE with no workgroup barrier in between -- a race on the hardware whatever order the
# emulator's fibers ran in.  A launch with such a race aborts the test (STGCN_EMU_RACE_WARN=1: report and go on).
#   tools/emu_race.sh [pytest args]      default: the block / head / model / optimizer tests, fp32 and bf16
set -e
cd "$(dirname "$0")/.."
python -c "from tests.emu.build_emu import build; print(build(race=True))"
export STGCN_EMU_RACE=1
if [ $# -gt 0 ]; then exec python -m pytest -x -q -p no:cacheprovider "$@"; fi
exec python -m pytest -x -q -p no:cacheprovider tests/test_emu_forward.py tests/test_emu_backward.py tests/test_emu_head.py tests/test_emu_bf16.py tests/test_emu_model.py tests/test_emu_optim.py tests/test_emu_gctile.py
