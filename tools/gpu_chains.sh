#!/bin/bash
# micro-batch chains A/B inside one GPU box: bench.py --chains k [--chain-graphs] (graph replay), alternating rounds
set -u
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd "$GRAFT_REPO_ROOT"
OUT="$GRAFT_REPO_ROOT/gpurun_out/${1:-chains}"
mkdir -p $OUT
for rep in 1 2; do
for cfg in 1 2 4; do
  timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-profile --chains $cfg 2> $OUT/err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('[chains $cfg]', d['value'], 'windows/s', d['ms_per_step'], 'ms/step', d['config']['launch'], d['config']['graph_error'])" | tee -a $OUT/chains.log
  grep -v -i "warn\|amdgpu.ids\|run_backward" $OUT/err.log | tail -3
done; done
