#!/bin/bash
# GPU pass: parity tests, report, bench, rocprof kernel trace (+ optional PMC passes); summaries only are kept
set -u
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd "$GRAFT_REPO_ROOT"
TAG=${1:-r02}
OUT="$GRAFT_REPO_ROOT/gpurun_out/$TAG"
mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q --maxfail=30 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log
tail -4 $OUT/pytest_gpu.log
timeout 600 python tools/gpu_report.py > $OUT/report.jsonl 2> $OUT/report.err
echo "report exit $?"
grep -E "fused_step|kernel_profile|unfused" $OUT/report.jsonl | cut -c1-1800
timeout 600 python bench.py --steps 200 --warmup 20 > $OUT/bench.json 2> $OUT/bench.err
echo "bench exit $?"; cut -c1-300 $OUT/bench.json; tail -3 $OUT/bench.err
timeout 300 python bench.py --steps 200 --warmup 20 --no-graph --no-cpu-baseline --no-profile > $OUT/bench_eager.json 2>> $OUT/bench.err
echo "bench eager exit $?"; cut -c1-200 $OUT/bench_eager.json
REPO=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-profile --no-graph"
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -o trace -- $BENCH > $OUT/rocprof_trace.log 2>&1; echo "trace exit $?"
python $REPO/tools/rocpd_summary.py /tmp/prof/trace_results.db > $OUT/kernel_stats.md 2>&1
if [ "${2:-pmc}" = "pmc" ]; then
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d /tmp/prof -o pmc_sq -- $BENCH > $OUT/rocprof_pmc_sq.log 2>&1; echo "pmc sq exit $?"
python $REPO/tools/rocpd_pmc_summary.py /tmp/prof/pmc_sq_results.db > $OUT/pmc_sq.md 2>&1
timeout 300 env STGCN_LAUNCH_LOG=/tmp/prof/launch.log rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/prof -o pmc_fetch -- $BENCH > $OUT/rocprof_pmc_fetch.log 2>&1; echo "pmc fetch exit $?"
python $REPO/tools/rocpd_pmc_summary.py /tmp/prof/pmc_fetch_results.db > $OUT/pmc_fetch.md 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/prof -o pmc_write -- $BENCH > $OUT/rocprof_pmc_write.log 2>&1; echo "pmc write exit $?"
python $REPO/tools/rocpd_pmc_summary.py /tmp/prof/pmc_write_results.db > $OUT/pmc_write.md 2>&1
python $REPO/tools/pmc_traffic.py /tmp/prof/pmc_fetch_results.db /tmp/prof/pmc_write_results.db /tmp/prof/launch.log > $OUT/pmc_traffic.json 2> $OUT/pmc_traffic.err
fi
head -30 $OUT/pmc_sq.md 2>/dev/null | cut -c1-400
du -sh $OUT
