#!/bin/bash
# pass r3-30: per-kernel times of the C2 step with (timing-only builds, wrong results) 1 = activation loads replaced by an opaque zero,
# 2 = fp32 MFMAs skipped (operands kept alive), 3 = activation stores skipped (values kept alive)
OUT=$GRAFT_REPO_ROOT/$1
cd $GRAFT_REPO_ROOT
for V in 0 1 2 3; do
  if [ $V = 0 ]; then E=""; else E="STGCN_AMD_LIB=$GRAFT_REPO_ROOT/stgcn_amd/_dbg/libstgcn_ts$V.so"; fi
  env $E timeout 600 python bench.py --config ${CFG:-c2} --steps 100 --warmup 10 --no-cpu-baseline --no-gpu-baseline --no-secondary > $OUT/bench_${CFG:-c2}_ts$V.json 2> $OUT/bench_${CFG:-c2}_ts$V.err; echo "c2 ts$V exit $?"
  python -c "
import json; d=json.load(open('$OUT/bench_${CFG:-c2}_ts$V.json')); r=d['roofline']; pk=r['per_kernel_us_per_step']
print('ts$V', d['ms_per_step'], ' '.join(f'{k}={v:.1f}' for k,v in sorted(pk.items())))"
done
