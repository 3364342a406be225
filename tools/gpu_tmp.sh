#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4-07; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider --maxfail=10 -k "bf16 or gctile or block or model or graph" > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log; tail -4 $O/pytest_gpu.log | cut -c1-300
for C in c2 c3 c5; do bash tools/gpu_ab.sh gpurun_out/r4-07 $C new="A=1" 2>&1 | cut -c1-1500; done
