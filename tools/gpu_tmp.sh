#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4-12; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --maxfail=10 -k "bf16 and not c5_full" > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log | cut -c1-300
bash tools/gpu_ab.sh gpurun_out/r4-12 c3 f16="STGCN_GC_B16P=0" b16p="A=1" f16b="STGCN_GC_B16P=0" b16pb="A=1" 2>&1 | grep -v exit | sed 's/adamw.*gconv_bwd@0/gconv_bwd@0/' | cut -c1-100
