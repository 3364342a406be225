#!/bin/bash
# The CPU-emulated kernels under AddressSanitizer (SURVEY.md section 5 planned it): every global-memory and LDS access of
# a kernel is an ordinary host access in the emulator, so out-of-bounds reads / writes of the kernels show up here.
#   tools/emu_asan.sh [pytest args]      default: the head, torch-ops, backward and variant tests
# detect_stack_use_after_return=0: the fibers switch stacks by hand; detect_leaks=0: the interpreter's own allocations.
set -e
cd "$(dirname "$0")/.."
RT=$(python -c "from tests.emu.build_emu import asan_runtime; print(asan_runtime() or '')")
[ -n "$RT" ] || { echo "no shared ASan runtime next to the host clang"; exit 2; }
python -c "from tests.emu.build_emu import build; print(build(asan=True))"
export STGCN_EMU_ASAN=1 ASAN_OPTIONS=detect_stack_use_after_return=0:detect_leaks=0:abort_on_error=1:symbolize=1 LD_PRELOAD="$RT"
export ASAN_SYMBOLIZER_PATH=/opt/rocm/lib/llvm/bin/llvm-symbolizer
if [ $# -gt 0 ]; then exec python -m pytest -x -q -p no:cacheprovider "$@"; fi
exec python -m pytest -x -q -p no:cacheprovider tests/test_emu_head.py tests/test_emu_torch_ops.py tests/test_emu_backward.py tests/test_emu_forward.py tests/test_emu_optim.py
