#!/bin/bash
# step time vs batch size (fixed vs per-window cost of the step), graph replay + per-kernel eager timers
set -u
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd "$GRAFT_REPO_ROOT"
OUT="$GRAFT_REPO_ROOT/gpurun_out/${1:-bsweep}"
mkdir -p $OUT
for b in ${BS:-4 8 16 32 64 128}; do
  STGCN_BENCH_B=$b timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-side-configs --no-gpu-baseline 2> $OUT/err.log > $OUT/bench_b$b.json
  python -c "
import json,sys
d=json.load(open('$OUT/bench_b$b.json')); r=d['roofline']['per_kernel_us_per_step']
print('[B $b]', d['value'], 'windows/s', d['ms_per_step'], 'ms/step; kernel sum', round(sum(r.values()),1), 'us')
print('   ', ' '.join(f'{k}={v:.1f}' for k,v in sorted(r.items())))" | tee -a $OUT/bsweep.log
done
