#!/bin/bash
OUT=$1
cd $GRAFT_REPO_ROOT
./tools/ubench/mfma_rate > $OUT/mfma_rate.txt 2>&1; cat $OUT/mfma_rate.txt
timeout 600 python -m pytest tests/test_gpu_dp.py tests/test_gpu_bench_dp.py tests/test_gpu_block.py -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log; tail -6 $OUT/pytest_gpu.log
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-gpu-baseline > $OUT/bench_c2.json 2> $OUT/bench_c2.err; echo "c2 exit $?"; cut -c1-260 $OUT/bench_c2.json
timeout 300 python bench.py --config c3 --steps 100 --warmup 10 --no-cpu-baseline --no-gpu-baseline > $OUT/bench_c3_bf16.json 2> $OUT/bench_c3_bf16.err; echo "c3 bf16 exit $?"; cut -c1-260 $OUT/bench_c3_bf16.json
STGCN_GCBWD2_PARTS=1 timeout 300 python bench.py --config c3 --steps 100 --warmup 10 --no-cpu-baseline --no-gpu-baseline > $OUT/bench_c3_bf16_p1.json 2> $OUT/bench_c3_bf16_p1.err; echo "c3 bf16 parts1 exit $?"; cut -c1-260 $OUT/bench_c3_bf16_p1.json
