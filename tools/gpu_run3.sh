#!/bin/bash
# One GPU pass, most important evidence first: parity suite, headline bench, smoke, side configs (C5 with the three operator
# precisions of the tiled graph conv, C3), rocprofv3 kernel traces (eager, hipGraph replay, C5), PMC passes; summaries only.
#   gpurun -- bash tools/gpu_run3.sh <tag> [nopmc]
set -u
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd "$GRAFT_REPO_ROOT"
REPO=$GRAFT_REPO_ROOT
TAG=${1:-r50}
OUT="$REPO/gpurun_out/$TAG"
mkdir -p $OUT
T0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - T0 )) s] $*"; }

timeout 700 python -m pytest tests -m gpu -q --maxfail=40 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log
tail -4 $OUT/pytest_gpu.log | cut -c1-300
grep -E "^(FAILED|ERROR)" $OUT/pytest_gpu.log | cut -c1-400 | head -20
stamp pytest

timeout 400 python bench.py --steps 200 --warmup 20 > $OUT/bench.json 2> $OUT/bench.err
echo "bench exit $?"; cut -c1-330 $OUT/bench.json; tail -3 $OUT/bench.err
stamp bench

timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?"; tail -2 $OUT/smoke.log | cut -c1-300
stamp smoke

timeout 400 python tools/gpu_side_configs.py c5 c3 --steps 5 --precision fp32 bf16x3 bf16 > $OUT/side_configs.jsonl 2> $OUT/side_configs.err
echo "side configs exit $?"; cut -c1-260 $OUT/side_configs.jsonl; tail -3 $OUT/side_configs.err
python - "$OUT/side_configs.jsonl" <<'EOF'
import json, sys
for line in open(sys.argv[1]):
    try:
        d = json.loads(line)
    except ValueError:
        continue
    print(d["config"], d.get("operator_products"), d["ms_per_step"], "ms/step", {k: (v["avg_us"], v["algorithmic_tflops"], v["frac_of_mfma_peak"]) for k, v in d.get("operator_gemm", {}).items()})
EOF
stamp side-configs

cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-profile --no-graph"
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -o trace -- $BENCH > $OUT/rocprof_trace.log 2>&1; echo "trace exit $?"
python $REPO/tools/rocpd_summary.py /tmp/prof/trace_results.db > $OUT/kernel_stats.md 2>&1
stamp trace-eager
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -o traceg -- python $REPO/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-profile > $OUT/rocprof_trace_graph.log 2>&1; echo "graph trace exit $?"
python $REPO/tools/rocpd_summary.py /tmp/prof/traceg_results.db > $OUT/kernel_stats_graph.md 2>&1
head -14 $OUT/kernel_stats_graph.md | cut -c1-200
stamp trace-graph
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -o tracec5 -- python $REPO/tools/gpu_side_configs.py c5 --steps 2 --precision fp32 bf16x3 bf16 > $OUT/rocprof_trace_c5.log 2>&1; echo "c5 trace exit $?"
python $REPO/tools/rocpd_summary.py /tmp/prof/tracec5_results.db > $OUT/kernel_stats_c5.md 2>&1
head -12 $OUT/kernel_stats_c5.md | cut -c1-200
stamp trace-c5

cd $REPO
timeout 200 python bench.py --steps 200 --warmup 20 --no-graph --no-cpu-baseline --no-profile > $OUT/bench_eager.json 2>> $OUT/bench.err
echo "bench eager exit $?"; cut -c1-200 $OUT/bench_eager.json
stamp bench-eager

if [ "${2:-pmc}" = "pmc" ]; then
cd /tmp
timeout 200 env STGCN_LAUNCH_LOG=/tmp/prof/launch.log rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/prof -o pmc_fetch -- $BENCH > $OUT/rocprof_pmc_fetch.log 2>&1; echo "pmc fetch exit $?"
python $REPO/tools/rocpd_pmc_summary.py /tmp/prof/pmc_fetch_results.db > $OUT/pmc_fetch.md 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/prof -o pmc_write -- $BENCH > $OUT/rocprof_pmc_write.log 2>&1; echo "pmc write exit $?"
python $REPO/tools/rocpd_pmc_summary.py /tmp/prof/pmc_write_results.db > $OUT/pmc_write.md 2>&1
python $REPO/tools/pmc_traffic.py /tmp/prof/pmc_fetch_results.db /tmp/prof/pmc_write_results.db /tmp/prof/launch.log > $OUT/pmc_traffic.json 2> $OUT/pmc_traffic.err
stamp pmc-traffic
timeout 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d /tmp/prof -o pmc_sq -- $BENCH > $OUT/rocprof_pmc_sq.log 2>&1; echo "pmc sq exit $?"
python $REPO/tools/rocpd_pmc_summary.py /tmp/prof/pmc_sq_results.db > $OUT/pmc_sq.md 2>&1
stamp pmc-sq
fi
du -sh $OUT
