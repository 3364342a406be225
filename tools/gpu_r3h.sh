#!/bin/bash
OUT=$1
cd $GRAFT_REPO_ROOT
STGCN_BWD_PRECISION=bf16x3 timeout 1500 python -m pytest tests -m gpu -q --maxfail=40 -p no:cacheprovider > $OUT/pytest_gpu_x3.log 2>&1
echo "pytest (all GPU tests, backward products bf16x3) exit $?" >> $OUT/pytest_gpu_x3.log; tail -6 $OUT/pytest_gpu_x3.log
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-gpu-baseline > $OUT/bench_c2.json 2> $OUT/bench_c2.err; echo "c2 exit $?"; python -c "
import json; d=json.load(open('$OUT/bench_c2.json')); print(d['ms_per_step'], d['value'], d['config'].get('secondary_bwd_bf16x3'))"; tail -3 $OUT/bench_c2.err
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_h -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-gpu-baseline --no-profile --no-secondary --bwd-precision bf16x3 > $OUT/rocprof_trace_x3.log 2>&1; echo "trace exit $?" )
python tools/rocpd_summary.py /tmp/prof_h/trace_results.db > $OUT/kernel_stats_graph_x3.md 2>&1
head -26 $OUT/kernel_stats_graph_x3.md | cut -c1-130
