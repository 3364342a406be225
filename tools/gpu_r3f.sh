#!/bin/bash
# round-3 pass F: everything that goes into profiles/ for the current build
OUT=$1
cd $GRAFT_REPO_ROOT
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?"; tail -3 $OUT/smoke.log
timeout 1500 python -m pytest tests -m gpu -q --maxfail=40 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log; tail -4 $OUT/pytest_gpu.log
timeout 600 python bench.py --steps 200 --warmup 20 > $OUT/bench_c2_full.json 2> $OUT/bench_c2_full.err; echo "c2 full exit $?"; cut -c1-200 $OUT/bench_c2_full.json
for P in 0 1 2; do
  STGCN_GCBWD2_PARTS=$P timeout 300 python bench.py --config c3 --steps 100 --warmup 10 --no-cpu-baseline --no-gpu-baseline > $OUT/bench_c3_bf16_p$P.json 2> $OUT/bench_c3_bf16_p$P.err; echo "c3 bf16 parts=$P exit $?"; cut -c1-200 $OUT/bench_c3_bf16_p$P.json
done
timeout 600 python bench.py --config c5 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_c5_bf16.json 2> $OUT/bench_c5_bf16.err; echo "c5 bf16 exit $?"; cut -c1-200 $OUT/bench_c5_bf16.json
