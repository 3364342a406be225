#!/bin/bash
# scratch runner of the build -> measure loop: `gpurun -- 'bash tools/gpu_tmp.sh'` (edit freely; the committed state runs the final evidence pass)
cd $GRAFT_REPO_ROOT
bash tools/gpu_final.sh r4-22 2>&1 | grep -v "^ \|^{\|^}" | cut -c1-300
bash tools/gpu_round.sh r4-22 "sq" | cut -c1-250
BENCH_ARGS="--config c3" bash tools/gpu_round.sh r4-22c3 "trace" > /dev/null
