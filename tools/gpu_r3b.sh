#!/bin/bash
# round-3 pass B: GPU tests, C2 bench + replay trace, C3 / C5 bf16 bench lines, batch sweep
OUT=$1
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q --maxfail=40 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log; tail -8 $OUT/pytest_gpu.log
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-gpu-baseline > $OUT/bench_c2.json 2> $OUT/bench_c2.err; echo "c2 exit $?"; cut -c1-300 $OUT/bench_c2.json
timeout 300 python bench.py --config c3 --steps 100 --warmup 10 --no-cpu-baseline --no-gpu-baseline > $OUT/bench_c3_bf16.json 2> $OUT/bench_c3_bf16.err; echo "c3 bf16 exit $?"; cut -c1-300 $OUT/bench_c3_bf16.json
timeout 600 python bench.py --config c5 --steps 10 --warmup 3 --no-cpu-baseline --no-gpu-baseline > $OUT/bench_c5_bf16.json 2> $OUT/bench_c5_bf16.err; echo "c5 bf16 exit $?"; cut -c1-300 $OUT/bench_c5_bf16.json; tail -3 $OUT/bench_c5_bf16.err
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_b -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-gpu-baseline --no-profile > $OUT/rocprof_trace.log 2>&1; echo "trace exit $?" )
python tools/rocpd_summary.py /tmp/prof_b/trace_results.db > $OUT/kernel_stats_graph.md 2>&1
head -30 $OUT/kernel_stats_graph.md | cut -c1-140
