#!/bin/bash
OUT=$1
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_block.py tests/test_gpu_model.py -m gpu -q -s -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log; grep "bf16x3 backward errors" $OUT/pytest_gpu.log; tail -4 $OUT/pytest_gpu.log
for V in fp32 bf16x3; do
  timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-gpu-baseline --bwd-precision $V > $OUT/bench_c2_$V.json 2> $OUT/bench_c2_$V.err; echo "c2 $V exit $?"; cut -c1-200 $OUT/bench_c2_$V.json
done
timeout 300 python bench.py --config c3 --dtype f32 --steps 100 --warmup 10 --no-cpu-baseline --no-gpu-baseline --bwd-precision bf16x3 > $OUT/bench_c3_f32_x3.json 2> $OUT/bench_c3_f32_x3.err; echo "c3 f32 x3 exit $?"; cut -c1-200 $OUT/bench_c3_f32_x3.json
