#!/bin/bash
# pass r3-21: which half of the big GEMM's epilogue costs what (DBG 8: none, 9: no operand-form stores, 10: no slab-layout pass)
OUT=$GRAFT_REPO_ROOT/$1
cd $GRAFT_REPO_ROOT
for V in 0 8 9 10; do
  if [ $V = 0 ]; then E=""; else E="STGCN_AMD_LIB=$GRAFT_REPO_ROOT/stgcn_amd/_dbg/libstgcn_dbg$V.so"; fi
  env $E timeout 600 python bench.py --config c5 --steps 10 --warmup 3 --no-cpu-baseline --no-gpu-baseline > $OUT/bench_c5_dbg$V.json 2> $OUT/bench_c5_dbg$V.err; echo "c5 dbg$V exit $?"
  python -c "
import json; d=json.load(open('$OUT/bench_c5_dbg$V.json')); r=d['roofline']; pk=r['per_kernel_us_per_step']
print('dbg$V', d['ms_per_step'], {k:v for k,v in pk.items() if 'gso' in k})"
done
