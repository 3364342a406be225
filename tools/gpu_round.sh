#!/bin/bash
# One GPU pass of a build -> measure iteration (gpurun -- 'bash tools/gpu_round.sh <tag> [stages]').
#   stages (default "test bench trace"): test = pytest -m gpu | bench = bench.py 200 steps with the per-kernel timer |
#   trace = rocprofv3 kernel trace of the graph-replayed bench | pmc = FETCH/WRITE counter passes | sq = SQ busy / wait counters | smoke | side = C3/C5 configs
# Everything lands in gpurun_out/<tag>/ ; summaries worth keeping are copied to profiles/ by hand.
set -u
export HSA_ENABLE_IPC_MODE_LEGACY=0
REPO=$GRAFT_REPO_ROOT
TAG=${1:-r2}
STAGES=${2:-"test bench trace"}
OUT="$REPO/gpurun_out/$TAG"
mkdir -p $OUT
cd $REPO
rocminfo 2>/dev/null | grep -m2 -E "gfx|Marketing" > $OUT/device.txt
for S in $STAGES; do
case $S in
test)
    timeout 900 python -m pytest tests -m gpu -q --maxfail=30 -p no:cacheprovider ${PYTEST_ARGS:-} > $OUT/pytest_gpu.log 2>&1
    echo "pytest exit $?" >> $OUT/pytest_gpu.log; tail -4 $OUT/pytest_gpu.log ;;
bench)
    timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-side-configs ${BENCH_ARGS:-} > $OUT/bench.json 2> $OUT/bench.err
    echo "bench exit $?"; cat $OUT/bench.json ;;
benchfull)
    timeout 600 python bench.py --steps 200 --warmup 20 --no-side-configs ${BENCH_ARGS:-} > $OUT/bench_full.json 2> $OUT/bench_full.err
    echo "benchfull exit $?"; cat $OUT/bench_full.json ;;
smoke)
    timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?"; tail -2 $OUT/smoke.log ;;
trace)
    ( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o trace -- python $REPO/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-side-configs --no-gpu-baseline --no-profile ${BENCH_ARGS:-} > $OUT/rocprof_trace.log 2>&1; echo "trace exit $?" )
    python tools/rocpd_summary.py /tmp/prof_$TAG/trace_results.db > $OUT/kernel_stats_graph.md 2>&1
    head -60 $OUT/kernel_stats_graph.md | cut -c1-150 ;;
pmc)
    for C in FETCH_SIZE WRITE_SIZE; do
        ( cd /tmp && export TMPDIR=/tmp && STGCN_LAUNCH_LOG=$OUT/launch_$C.log timeout 300 rocprofv3 --kernel-trace --pmc $C -d /tmp/pmc_${TAG}_$C -o pmc -- python $REPO/bench.py --steps ${PMC_STEPS:-6} --warmup 2 --no-cpu-baseline --no-side-configs --no-gpu-baseline --no-profile --no-graph ${BENCH_ARGS:-} > $OUT/pmc_$C.log 2>&1; echo "pmc $C exit $?" )
        python tools/rocpd_pmc_summary.py /tmp/pmc_${TAG}_$C/pmc_results.db > $OUT/pmc_$C.md 2>&1
    done
    python tools/pmc_traffic.py /tmp/pmc_${TAG}_FETCH_SIZE/pmc_results.db /tmp/pmc_${TAG}_WRITE_SIZE/pmc_results.db $OUT/launch_FETCH_SIZE.log $OUT/launch_WRITE_SIZE.log > $OUT/pmc_traffic.json 2> $OUT/pmc_traffic.err
    python -c "import sqlite3,sys; db=sqlite3.connect('/tmp/pmc_${TAG}_FETCH_SIZE/pmc_results.db'); print([d[1] for d in db.execute('pragma table_info(counters_collection)')])" > $OUT/pmc_schema.txt 2>&1
    head -c 1200 $OUT/pmc_traffic.json; cat $OUT/pmc_traffic.err | tail -3 ;;
sq)
    ( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d /tmp/pmc_${TAG}_sq -o pmc -- python $REPO/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-side-configs --no-gpu-baseline --no-profile --no-graph > $OUT/pmc_sq.log 2>&1; echo "pmc sq exit $?" )
    python tools/rocpd_pmc_summary.py /tmp/pmc_${TAG}_sq/pmc_results.db > $OUT/pmc_sq.md 2>&1
    grep -E "tc1_bwd|tc2_bwd|tc2_ln|tc1_fwd|gconv" $OUT/pmc_sq.md | cut -c1-230 ;;
side)
    timeout 900 python tools/gpu_side_configs.py ${SIDE_ARGS:-} > $OUT/side_configs.jsonl 2> $OUT/side_configs.err; echo "side exit $?"; cat $OUT/side_configs.jsonl ;;
*)  # anything else: a script path relative to the repo
    timeout 900 bash $S $OUT; echo "$S exit $?" ;;
esac
done
