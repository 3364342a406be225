#!/bin/bash
# pass r3-25: what the slab-resident graph-conv kernels spend their time on (timing-only builds, wrong results: STGCN_GC_DBG 1 = no operator products,
# 2 = no staging loads of the X_k / A slabs, 3 = no parameter-gradient jobs)
OUT=$GRAFT_REPO_ROOT/$1
cd $GRAFT_REPO_ROOT
for C in c2 c3; do for V in 0 1 2 3; do
  if [ $V = 0 ]; then E=""; else E="STGCN_AMD_LIB=$GRAFT_REPO_ROOT/stgcn_amd/_dbg/libstgcn_gc$V.so"; fi
  env $E timeout 600 python bench.py --config $C --steps 100 --warmup 10 --no-cpu-baseline --no-gpu-baseline --no-secondary > $OUT/bench_${C}_gc$V.json 2> $OUT/bench_${C}_gc$V.err; echo "$C gc$V exit $?"
  python -c "
import json; d=json.load(open('$OUT/bench_${C}_gc$V.json')); r=d['roofline']; pk=r['per_kernel_us_per_step']
print('$C gc$V', d['ms_per_step'], {k:v for k,v in pk.items() if 'gconv' in k})"
done; done
