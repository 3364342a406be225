#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4-16
P=$GRAFT_REPO_ROOT/stgcn_amd/libstgcn_hip_prev.so
timeout 900 python -m pytest tests/test_gpu_block.py tests/test_gpu_model.py tests/test_gpu_bf16.py -x -q -m gpu -k "not c5 and not big_operator" > gpurun_out/r4-16/pytest.log 2>&1; echo "pytest exit $?"; tail -n 3 gpurun_out/r4-16/pytest.log
bash tools/gpu_ab.sh gpurun_out/r4-16 c2 prev="STGCN_AMD_LIB=$P" new="A=1" prevb="STGCN_AMD_LIB=$P" newb="A=1" 2>&1 | cut -c1-700
bash tools/gpu_ab.sh gpurun_out/r4-16 c3 prev="STGCN_AMD_LIB=$P" new="A=1" prevb="STGCN_AMD_LIB=$P" newb="A=1" 2>&1 | cut -c1-700
STEPS=50 bash tools/gpu_ab.sh gpurun_out/r4-16 c5 prev="STGCN_AMD_LIB=$P" new="A=1" 2>&1 | cut -c1-1200
