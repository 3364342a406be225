#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4-15
timeout 300 python tools/dbg_gpu_head.py 2>&1 | grep -v "bad\|^ " | tail -20
timeout 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_model.py tests/test_gpu_graph.py -x -q -m gpu -k "head or c3_model or model or graph or replay" > gpurun_out/r4-15/pytest.log 2>&1; echo "pytest exit $?"; tail -n 4 gpurun_out/r4-15/pytest.log
bash tools/gpu_ab.sh gpurun_out/r4-15 c2 base="STGCN_HEAD_FUSE=0" fuse="STGCN_HEAD_FUSE=1" baseb="STGCN_HEAD_FUSE=0" fuseb="STGCN_HEAD_FUSE=1" 2>&1 | cut -c1-400
bash tools/gpu_ab.sh gpurun_out/r4-15 c3 base="STGCN_HEAD_FUSE=0" fuse="STGCN_HEAD_FUSE=1" baseb="STGCN_HEAD_FUSE=0" fuseb="STGCN_HEAD_FUSE=1" 2>&1 | cut -c1-400
