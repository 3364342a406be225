#!/bin/bash
# PMC passes over the C5 operator GEMMs (tools/gpu_side_configs.py c5, one step per precision): SQ busy / wait counters, L2 fetch size
set -u
export HSA_ENABLE_IPC_MODE_LEGACY=0
REPO=$GRAFT_REPO_ROOT
OUT="$REPO/gpurun_out/${1:-c5pmc}"
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/tools/gpu_side_configs.py c5 --steps 1 --precision fp32 bf16x3 bf16"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d /tmp/prof -o c5sq -- $CMD > $OUT/rocprof_c5_sq.log 2>&1; echo "sq exit $?"
python $REPO/tools/rocpd_pmc_summary.py /tmp/prof/c5sq_results.db > $OUT/c5_pmc_sq.md 2>&1
grep -E "kernel|gso_gemm" $OUT/c5_pmc_sq.md | cut -c1-400
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/prof -o c5fetch -- $CMD > $OUT/rocprof_c5_fetch.log 2>&1; echo "fetch exit $?"
python $REPO/tools/rocpd_pmc_summary.py /tmp/prof/c5fetch_results.db > $OUT/c5_pmc_fetch.md 2>&1
grep -E "kernel|gso_gemm" $OUT/c5_pmc_fetch.md | cut -c1-300
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_VMEM TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum -d /tmp/prof -o c5l2 -- $CMD > $OUT/rocprof_c5_l2.log 2>&1; echo "l2 exit $?"
python $REPO/tools/rocpd_pmc_summary.py /tmp/prof/c5l2_results.db > $OUT/c5_pmc_l2.md 2>&1
grep -E "kernel|gso_gemm" $OUT/c5_pmc_l2.md | cut -c1-400
tail -5 $OUT/rocprof_c5_l2.log | cut -c1-300
