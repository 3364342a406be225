#!/bin/bash
# pass r3-29 (r3-34: tc1_bwd with raw two-set prefetch; r3-35: + raw staging registers in the weight-gradient kernels, batched staging loads in gconv_fwd / gconv_bwd): static fragment ring (depth 2) in the slab graph-conv kernels + the hooked LayerNorm's dropout offset read once per kernel
OUT=$GRAFT_REPO_ROOT/$1
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q --maxfail=20 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log; tail -3 $OUT/pytest_gpu.log
for C in c2 c3; do
  timeout 600 python bench.py --config $C --steps 200 --warmup 20 --no-cpu-baseline --no-gpu-baseline > $OUT/bench_$C.json 2> $OUT/bench_$C.err; echo "$C exit $?"
  python -c "
import json; d=json.load(open('$OUT/bench_$C.json')); r=d['roofline']; pk=r['per_kernel_us_per_step']
print('$C', d['ms_per_step'], d['value'], d['config'].get('secondary_bwd_bf16x3',{}).get('value'), pk)"
done
