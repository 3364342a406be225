#!/bin/bash
# first GPU pass: parity tests, report, bench, rocprof kernel trace
set -u
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT="$GRAFT_REPO_ROOT/gpurun_out"
rocminfo 2>/dev/null | grep -m2 -E "gfx|Marketing" > $OUT/device.txt
timeout 900 python -m pytest tests -m gpu -q --maxfail=30 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log
tail -5 $OUT/pytest_gpu.log
timeout 600 python tools/gpu_report.py > $OUT/report.jsonl 2> $OUT/report.err
echo "report exit $?"
timeout 600 python bench.py --steps 200 --warmup 20 > $OUT/bench.json 2> $OUT/bench.err
echo "bench exit $?"; cat $OUT/bench.json
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?"; tail -2 $OUT/smoke.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o r01 -- python $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-profile > $OUT/rocprof.log 2>&1
echo "rocprof exit $?"
ls -R $OUT/prof | head -20
