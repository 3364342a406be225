#!/bin/bash
# C5 bf16: kernel trace + FETCH_SIZE counter pass of the operator GEMMs (big tiles vs the 128 x 128 kernel)
OUT=$1
cd $GRAFT_REPO_ROOT
for V in big small; do
  case $V in big) E="";; small) E="STGCN_GEMM_BIG=0";; esac
  ( cd /tmp && export TMPDIR=/tmp && env $E timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmc_c5_$V -o pmc -- python $GRAFT_REPO_ROOT/bench.py --config c5 --steps 2 --warmup 1 --no-cpu-baseline --no-gpu-baseline --no-profile --no-graph > $OUT/pmc_c5_$V.log 2>&1; echo "pmc $V exit $?" )
  python tools/rocpd_pmc_summary.py /tmp/pmc_c5_$V/pmc_results.db > $OUT/pmc_c5_fetch_$V.md 2>&1
  grep -E "gso_gemm|kernel" $OUT/pmc_c5_fetch_$V.md | head -8 | cut -c1-220
done
( cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_c5 -o trace -- python $GRAFT_REPO_ROOT/bench.py --config c5 --steps 4 --warmup 2 --no-cpu-baseline --no-gpu-baseline --no-profile > $OUT/rocprof_c5.log 2>&1; echo "trace exit $?" )
python tools/rocpd_summary.py /tmp/prof_c5/trace_results.db > $OUT/kernel_stats_c5.md 2>&1
head -24 $OUT/kernel_stats_c5.md | cut -c1-140
