#!/bin/bash
# pass r3-24: operator-fragment prefetch ring of the slab-resident graph-conv kernels, depth 4 (default build) vs 2 (stgcn_amd/_dbg/libstgcn_pf2.so)
OUT=$GRAFT_REPO_ROOT/$1
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_block.py tests/test_gpu_bf16.py -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log; tail -3 $OUT/pytest_gpu.log
for C in c2 c3; do for V in pf4 pf2; do
  if [ $V = pf4 ]; then E=""; else E="STGCN_AMD_LIB=$GRAFT_REPO_ROOT/stgcn_amd/_dbg/libstgcn_pf2.so"; fi
  env $E timeout 600 python bench.py --config $C --steps 200 --warmup 20 --no-cpu-baseline --no-gpu-baseline --no-secondary > $OUT/bench_${C}_$V.json 2> $OUT/bench_${C}_$V.err; echo "$C $V exit $?"
  python -c "
import json; d=json.load(open('$OUT/bench_${C}_$V.json')); r=d['roofline']; pk=r['per_kernel_us_per_step']
print('$C $V', d['ms_per_step'], d['value'], {k:v for k,v in pk.items() if 'gconv' in k})"
done; done
