#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4-21
timeout 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_bf16.py tests/test_gpu_rccl.py tests/test_gpu_bench_dp.py -x -q -m gpu -k "golden or head or rccl or bench" > gpurun_out/r4-21/pytest.log 2>&1; echo "pytest exit $?"; tail -n 2 gpurun_out/r4-21/pytest.log
STEPS=50 bash tools/gpu_ab.sh gpurun_out/r4-21 c5 new="A=1" 2>&1 | python3 -c "
import sys,re
for l in sys.stdin:
    if 'exit' in l: continue
    p=l.split(); print(p[0],p[1],p[2],p[3],' '.join(x for x in p[4:] if 'wgrad' in x or 'head.reduce' in x))"
