#!/bin/bash
# PMC passes over the C5 operator GEMMs (tools/gpu_side_configs.py c5, one step per precision): L2 fetch size, L2 hit / miss, SQ busy / wait
#   gpurun -- bash tools/gpu_run_c5pmc.sh <tag> ["fp32 bf16"]
set -u
export HSA_ENABLE_IPC_MODE_LEGACY=0
REPO=$GRAFT_REPO_ROOT
OUT="$REPO/gpurun_out/${1:-c5pmc}"
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/tools/gpu_side_configs.py c5 --steps 1 --precision ${2:-fp32 bf16}"
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/prof -o c5fetch -- $CMD > $OUT/rocprof_c5_fetch.log 2>&1; echo "fetch exit $?"
python $REPO/tools/rocpd_pmc_summary.py /tmp/prof/c5fetch_results.db > $OUT/c5_pmc_fetch.md 2>&1
grep -E "^\| kernel|gso_gemm" $OUT/c5_pmc_fetch.md | cut -c1-300
timeout 200 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum -d /tmp/prof -o c5l2 -- $CMD > $OUT/rocprof_c5_l2.log 2>&1; echo "l2 exit $?"
python $REPO/tools/rocpd_pmc_summary.py /tmp/prof/c5l2_results.db > $OUT/c5_pmc_l2.md 2>&1
grep -E "^\| kernel|gso_gemm" $OUT/c5_pmc_l2.md | cut -c1-400
timeout 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d /tmp/prof -o c5sq -- $CMD > $OUT/rocprof_c5_sq.log 2>&1; echo "sq exit $?"
python $REPO/tools/rocpd_pmc_summary.py /tmp/prof/c5sq_results.db > $OUT/c5_pmc_sq.md 2>&1
grep -E "^\| kernel|gso_gemm" $OUT/c5_pmc_sq.md | cut -c1-400
