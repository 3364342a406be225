#!/bin/bash
# scratch runner of the build -> measure loop: `gpurun -- 'bash tools/gpu_tmp.sh'` (edit freely)
cd $GRAFT_REPO_ROOT
bash tools/gpu_bsweep.sh r4-24 2>&1 | tail -12
