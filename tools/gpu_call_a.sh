#!/bin/bash
# GPU call: full parity suite, then the micro-batch chain A/B
set -u
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd "$GRAFT_REPO_ROOT"
OUT="$GRAFT_REPO_ROOT/gpurun_out/${1:-a}"
mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q --maxfail=30 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log
tail -6 $OUT/pytest_gpu.log
bash tools/gpu_chains.sh ${1:-a}
