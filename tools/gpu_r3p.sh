#!/bin/bash
# pass r3-20 (r3-22: slab pass with batched 16-byte accesses): big GEMM with the LDS-transposed epilogue (default 64-deep steps); tiled / bf16 GPU tests + C5 bench + trace
OUT=$GRAFT_REPO_ROOT/$1
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_gctile.py tests/test_gpu_bf16.py -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log; tail -3 $OUT/pytest_gpu.log
for V in bk64 bk32; do
  case $V in bk32) E="STGCN_GEMM_BIG_BK=32";; bk64) E="STGCN_GEMM_BIG_BK=64";; esac
  env $E timeout 600 python bench.py --config c5 --steps 10 --warmup 3 --no-cpu-baseline --no-gpu-baseline > $OUT/bench_c5_$V.json 2> $OUT/bench_c5_$V.err; echo "c5 $V exit $?"
  python -c "
import json; d=json.load(open('$OUT/bench_c5_$V.json')); r=d['roofline']; pk=r['per_kernel_us_per_step']
print('$V', d['ms_per_step'], d['value'], {k:v for k,v in pk.items() if 'gso' in k})"
done
( cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_c5 -o trace -- python $GRAFT_REPO_ROOT/bench.py --config c5 --steps 4 --warmup 2 --no-cpu-baseline --no-gpu-baseline --no-profile > $OUT/rocprof_c5.log 2>&1; echo "trace exit $?" )
python tools/rocpd_summary.py /tmp/prof_c5/trace_results.db > $OUT/kernel_stats_c5.md 2>&1
head -16 $OUT/kernel_stats_c5.md | cut -c1-140
