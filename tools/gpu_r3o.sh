#!/bin/bash
# pass r3-19: cost of the big GEMM's epilogue (DBG 8: none)
OUT=$GRAFT_REPO_ROOT/$1
cd $GRAFT_REPO_ROOT
for V in bk64 bk64_noepi; do
  case $V in bk64) E="STGCN_GEMM_BIG_BK=64";; bk64_noepi) E="STGCN_GEMM_BIG_BK=64 STGCN_AMD_LIB=$GRAFT_REPO_ROOT/stgcn_amd/_dbg/libstgcn_dbg8.so";; esac
  env $E timeout 600 python bench.py --config c5 --steps 10 --warmup 3 --no-cpu-baseline --no-gpu-baseline > $OUT/bench_c5_$V.json 2> $OUT/bench_c5_$V.err; echo "c5 $V exit $?"
  python -c "
import json; d=json.load(open('$OUT/bench_c5_$V.json')); r=d['roofline']; pk=r['per_kernel_us_per_step']
print('$V', d['ms_per_step'], {k:v for k,v in pk.items() if 'gso' in k})"
done
