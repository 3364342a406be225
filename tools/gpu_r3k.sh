#!/bin/bash
# pass r3-14: software-pipelined big GEMM (fragment reads of step kb+1 inside step kb) vs the r3-12 form, C5
OUT=$1
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_gctile.py tests/test_gpu_bf16.py -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log; tail -4 $OUT/pytest_gpu.log
for V in swp noswp; do
  case $V in swp) E="";; noswp) E="STGCN_GEMM_BIG_SWP=0";; esac
  env $E timeout 600 python bench.py --config c5 --steps 10 --warmup 3 --no-cpu-baseline --no-gpu-baseline > $OUT/bench_c5_$V.json 2> $OUT/bench_c5_$V.err; echo "c5 $V exit $?"
  python -c "
import json; d=json.load(open('$OUT/bench_c5_$V.json')); r=d['roofline']; pk=r['per_kernel_us_per_step']
print('$V', d['ms_per_step'], d['value'], r['kernel'], r['frac'], {k:v for k,v in pk.items() if 'gso' in k})"
done
