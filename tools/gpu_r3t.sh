#!/bin/bash
# pass r3-26 (r3-27: with the static fragment ring in the product loops of gconv_fwd / gconv_bwd / gconv_bwd2): forward graph conv with two slabs per workgroup (STGCN_GC_SP: unset = launcher's rule, 1, 2 = forced for both blocks)
OUT=$GRAFT_REPO_ROOT/$1
cd $GRAFT_REPO_ROOT
STGCN_GC_SP=2 timeout 900 python -m pytest tests/test_gpu_block.py tests/test_gpu_bf16.py -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu_sp2.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu_sp2.log; tail -3 $OUT/pytest_gpu_sp2.log
for C in c2 c3; do for V in auto sp1 sp2; do
  case $V in auto) E="";; sp1) E="STGCN_GC_SP=1";; sp2) E="STGCN_GC_SP=2";; esac
  env $E timeout 600 python bench.py --config $C --steps 200 --warmup 20 --no-cpu-baseline --no-gpu-baseline --no-secondary > $OUT/bench_${C}_$V.json 2> $OUT/bench_${C}_$V.err; echo "$C $V exit $?"
  python -c "
import json; d=json.load(open('$OUT/bench_${C}_$V.json')); r=d['roofline']; pk=r['per_kernel_us_per_step']
print('$C $V', d['ms_per_step'], d['value'], {k:v for k,v in pk.items() if 'gconv' in k})"
done; done
