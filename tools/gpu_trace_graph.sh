#!/bin/bash
# rocprofv3 kernel trace of the graph-replayed bench (kernel durations as they are inside the replay), summary only
set -u
export HSA_ENABLE_IPC_MODE_LEGACY=0
REPO=$GRAFT_REPO_ROOT
OUT="$REPO/gpurun_out/${1:-trace}"
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -o trace -- python $REPO/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-profile ${BENCH_ARGS:-} > $OUT/rocprof_trace.log 2>&1; echo "trace exit $?"
python $REPO/tools/rocpd_summary.py /tmp/prof/trace_results.db > $OUT/kernel_stats_graph.md 2>&1
head -75 $OUT/kernel_stats_graph.md | cut -c1-160
