#!/bin/bash
# round-3 pass A: GPU tests (fp32 regression after the ET refactor + the new bf16 tests), then C2 fp32 / C3 bf16 / C3 fp32 bench lines
OUT=$1
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q --maxfail=40 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log; tail -15 $OUT/pytest_gpu.log
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-gpu-baseline > $OUT/bench_c2.json 2> $OUT/bench_c2.err; echo "c2 exit $?"; cut -c1-700 $OUT/bench_c2.json
timeout 300 python bench.py --config c3 --steps 100 --warmup 10 --no-cpu-baseline --no-gpu-baseline > $OUT/bench_c3_bf16.json 2> $OUT/bench_c3_bf16.err; echo "c3 bf16 exit $?"; cut -c1-600 $OUT/bench_c3_bf16.json; tail -3 $OUT/bench_c3_bf16.err
timeout 300 python bench.py --config c3 --dtype f32 --steps 100 --warmup 10 --no-cpu-baseline --no-gpu-baseline > $OUT/bench_c3_f32.json 2> $OUT/bench_c3_f32.err; echo "c3 f32 exit $?"; cut -c1-400 $OUT/bench_c3_f32.json
