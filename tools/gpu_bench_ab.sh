#!/bin/bash
# alternating bench.py runs (hipGraph replay, no profile pass) under different environments:  gpu_bench_ab.sh "A=1" "A=0" ...
set -u
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
for v in "$@"; do
  env $v timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-profile 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('[$v]', d['value'], 'windows/s', d['ms_per_step'], 'ms/step', d['config']['launch'])"
done; done
