#!/bin/bash
# pass r3-17 (r3-18: the same with the copies spread over the MFMA groups): big GEMM with 64-deep steps (whole 128-B lines per staged row, 2 buffers) vs 32-deep (4 buffers); + the copy stream alone (DBG 3)
OUT=$GRAFT_REPO_ROOT/$1
cd $GRAFT_REPO_ROOT
STGCN_GEMM_BIG_BK=64 timeout 900 python -m pytest tests/test_gpu_gctile.py tests/test_gpu_bf16.py -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu_bk64.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu_bk64.log; tail -3 $OUT/pytest_gpu_bk64.log
for V in bk32 bk64 bk32_nomfma bk64_nomfma; do
  case $V in bk32) E="STGCN_GEMM_BIG_BK=32";; bk64) E="STGCN_GEMM_BIG_BK=64";;
    bk32_nomfma) E="STGCN_GEMM_BIG_BK=32 STGCN_AMD_LIB=$GRAFT_REPO_ROOT/stgcn_amd/_dbg/libstgcn_dbg3.so";;
    bk64_nomfma) E="STGCN_GEMM_BIG_BK=64 STGCN_AMD_LIB=$GRAFT_REPO_ROOT/stgcn_amd/_dbg/libstgcn_dbg3.so";; esac
  env $E timeout 600 python bench.py --config c5 --steps 10 --warmup 3 --no-cpu-baseline --no-gpu-baseline > $OUT/bench_c5_$V.json 2> $OUT/bench_c5_$V.err; echo "c5 $V exit $?"
  python -c "
import json; d=json.load(open('$OUT/bench_c5_$V.json')); r=d['roofline']; pk=r['per_kernel_us_per_step']
print('$V', d['ms_per_step'], {k:v for k,v in pk.items() if 'gso' in k})"
done
