#!/bin/bash
# pass r3-28: ring depth of the operator-fragment prefetch for one-tile waves (4 = default build, 3, 2), forward with one slab per workgroup
OUT=$GRAFT_REPO_ROOT/$1
cd $GRAFT_REPO_ROOT
for C in c2 c3; do for V in ring4 ring3 ring2; do
  case $V in ring4) E="";; ring3) E="STGCN_AMD_LIB=$GRAFT_REPO_ROOT/stgcn_amd/_dbg/libstgcn_ring3.so";; ring2) E="STGCN_AMD_LIB=$GRAFT_REPO_ROOT/stgcn_amd/_dbg/libstgcn_ring2.so";; esac
  env STGCN_GC_SP=1 $E timeout 600 python bench.py --config $C --steps 200 --warmup 20 --no-cpu-baseline --no-gpu-baseline --no-secondary > $OUT/bench_${C}_$V.json 2> $OUT/bench_${C}_$V.err; echo "$C $V exit $?"
  python -c "
import json; d=json.load(open('$OUT/bench_${C}_$V.json')); r=d['roofline']; pk=r['per_kernel_us_per_step']
print('$C $V', d['ms_per_step'], d['value'], {k:v for k,v in pk.items() if 'gconv' in k})"
done; done
