#!/bin/bash
# pass r3-32: C3 bf16 with the 16-wave / 6-tile tc2_ln_fwd (257 .. 384 nodes) and two tc1_fwd workgroups per CU; STGCN_TC2LN_HV=2 = the 8-wave form
OUT=$GRAFT_REPO_ROOT/$1
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_block.py -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log; tail -3 $OUT/pytest_gpu.log
for V in new hv2; do
  case $V in new) E="";; hv2) E="STGCN_TC2LN_HV=2";; esac
  env $E timeout 600 python bench.py --config c3 --steps 200 --warmup 20 --no-cpu-baseline --no-gpu-baseline > $OUT/bench_c3_$V.json 2> $OUT/bench_c3_$V.err; echo "c3 $V exit $?"
  python -c "
import json; d=json.load(open('$OUT/bench_c3_$V.json')); r=d['roofline']; pk=r['per_kernel_us_per_step']
print('c3 $V', d['ms_per_step'], d['value'], {k:v for k,v in pk.items() if 'tc2_ln' in k or 'tc1' in k})"
done
