#!/bin/bash
OUT=$1
cd $GRAFT_REPO_ROOT
for V in default head16 hv2 gcp3 gcp5; do
  case $V in default) E="";; head16) E="STGCN_HEAD_TILE=16";; hv2) E="STGCN_TC2LN_HV=2";; gcp3) E="STGCN_GC_PARTS=3,0";; gcp5) E="STGCN_GC_PARTS=5,0";; esac
  env $E timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-gpu-baseline > $OUT/bench_c2_$V.json 2> $OUT/bench_c2_$V.err; echo "c2 $V exit $?"; cut -c1-200 $OUT/bench_c2_$V.json
done
STGCN_HEAD_TILE=16 timeout 300 python -m pytest tests/test_gpu_model.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3
