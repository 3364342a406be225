#!/bin/bash
# all GPU tests + the three bench lines (per-kernel us/step): gpu_pass.sh <out dir under the repo>
OUT=$GRAFT_REPO_ROOT/$1; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q --maxfail=20 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log; tail -3 $OUT/pytest_gpu.log
for C in c2 c3 c5; do
  timeout 600 python bench.py --config $C --steps 200 --warmup 20 --no-cpu-baseline --no-gpu-baseline > $OUT/bench_$C.json 2> $OUT/bench_$C.err; echo "$C exit $?"
  python -c "
import json; d=json.load(open('$OUT/bench_$C.json')); pk=d['roofline']['per_kernel_us_per_step']
print('$C', d['ms_per_step'], d['value'], d['config'].get('secondary_bwd_bf16x3',{}).get('value'), ' '.join(f'{k}={v:.1f}' for k,v in sorted(pk.items())))"
done
