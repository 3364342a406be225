#!/bin/bash
# pass r3-16: big GEMM diagnostics, part 2: L2-resident k range (DBG 5 with / 6 without MFMAs) + SQ / TCC counter passes of the normal build
OUT=$GRAFT_REPO_ROOT/$1
cd $GRAFT_REPO_ROOT
for V in 5 6; do
  E="STGCN_AMD_LIB=$GRAFT_REPO_ROOT/stgcn_amd/_dbg/libstgcn_dbg$V.so"
  env $E timeout 600 python bench.py --config c5 --steps 10 --warmup 3 --no-cpu-baseline --no-gpu-baseline > $OUT/bench_c5_dbg$V.json 2> $OUT/bench_c5_dbg$V.err; echo "c5 dbg$V exit $?"
  python -c "
import json; d=json.load(open('$OUT/bench_c5_dbg$V.json')); r=d['roofline']; pk=r['per_kernel_us_per_step']
print('dbg$V', d['ms_per_step'], {k:v for k,v in pk.items() if 'gso' in k})"
done
( cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d /tmp/pmc_c5_sq -o pmc -- python $GRAFT_REPO_ROOT/bench.py --config c5 --steps 2 --warmup 1 --no-cpu-baseline --no-gpu-baseline --no-profile --no-graph > $OUT/pmc_c5_sq.log 2>&1; echo "pmc sq exit $?" )
python tools/rocpd_pmc_summary.py /tmp/pmc_c5_sq/pmc_results.db > $OUT/pmc_c5_sq.md 2>&1
grep -E "gso_gemm|kernel" $OUT/pmc_c5_sq.md | head -6 | cut -c1-300
( cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum -d /tmp/pmc_c5_tcc -o pmc -- python $GRAFT_REPO_ROOT/bench.py --config c5 --steps 2 --warmup 1 --no-cpu-baseline --no-gpu-baseline --no-profile --no-graph > $OUT/pmc_c5_tcc.log 2>&1; echo "pmc tcc exit $?" )
python tools/rocpd_pmc_summary.py /tmp/pmc_c5_tcc/pmc_results.db > $OUT/pmc_c5_tcc.md 2>&1
grep -E "gso_gemm|kernel" $OUT/pmc_c5_tcc.md | head -6 | cut -c1-300
tail -3 $OUT/pmc_c5_tcc.log
