#!/bin/bash
# scratch runner of the build -> measure loop: `gpurun -- 'bash tools/gpu_tmp.sh'` (edit freely; as committed: the evidence pass of a round)
cd $GRAFT_REPO_ROOT
bash tools/gpu_final.sh ${1:-rX} 2>&1 | grep -v "^ \|^{\|^}" | cut -c1-300
