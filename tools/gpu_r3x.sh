#!/bin/bash
# pass r3-31: two workgroups per CU for the persistent tc1 kernels (STGCN_TC1_FWD_PER_CU / STGCN_TC1_BWD_PER_CU = 2): their steps are latency chains
OUT=$GRAFT_REPO_ROOT/$1
cd $GRAFT_REPO_ROOT
for C in c2 c3; do for V in base fwd2 bwd2 both; do
  case $V in base) E="";; fwd2) E="STGCN_TC1_FWD_PER_CU=2";; bwd2) E="STGCN_TC1_BWD_PER_CU=2";; both) E="STGCN_TC1_FWD_PER_CU=2 STGCN_TC1_BWD_PER_CU=2";; esac
  env $E timeout 600 python bench.py --config $C --steps 200 --warmup 20 --no-cpu-baseline --no-gpu-baseline --no-secondary > $OUT/bench_${C}_$V.json 2> $OUT/bench_${C}_$V.err; echo "$C $V exit $?"
  python -c "
import json; d=json.load(open('$OUT/bench_${C}_$V.json')); r=d['roofline']; pk=r['per_kernel_us_per_step']
print('$C $V', d['ms_per_step'], d['value'], {k:v for k,v in pk.items() if 'tc1' in k or 'reduce' in k})"
done; done
