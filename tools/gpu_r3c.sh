#!/bin/bash
# round-3 pass C: GPU tests, C2 bench (default / STGCN_GCBWD2=0 / STGCN_TC2_RECOMP=1), replay trace, C3 bf16
OUT=$1
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q --maxfail=40 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log; tail -6 $OUT/pytest_gpu.log
for V in default gcbwd2off recomp; do
  case $V in default) E="";; gcbwd2off) E="STGCN_GCBWD2=0";; recomp) E="STGCN_TC2_RECOMP=1";; esac
  env $E timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-gpu-baseline > $OUT/bench_c2_$V.json 2> $OUT/bench_c2_$V.err; echo "c2 $V exit $?"; cut -c1-260 $OUT/bench_c2_$V.json
done
timeout 300 python bench.py --config c3 --steps 100 --warmup 10 --no-cpu-baseline --no-gpu-baseline > $OUT/bench_c3_bf16.json 2> $OUT/bench_c3_bf16.err; echo "c3 bf16 exit $?"; cut -c1-260 $OUT/bench_c3_bf16.json
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_c -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-gpu-baseline --no-profile > $OUT/rocprof_trace.log 2>&1; echo "trace exit $?" )
python tools/rocpd_summary.py /tmp/prof_c/trace_results.db > $OUT/kernel_stats_graph.md 2>&1
head -28 $OUT/kernel_stats_graph.md | cut -c1-140
