#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/gpu_final.sh r4-19 2>&1 | grep -v "^ \|^{\|^}" | cut -c1-300
