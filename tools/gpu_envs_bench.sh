#!/bin/bash
# alternating bench.py runs (graph replay + profile pass) under different environments, printing selected kernels:
#   KFILTER=regex gpu_envs_bench.sh "A=1" "A=2" ...
set -u
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
for v in "$@"; do
  env $v timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys,re,os
d=json.loads(sys.stdin.read()); r=d.get('roofline',{}).get('per_kernel_us_per_step',{})
print('[$v]', d['value'], 'windows/s', d['ms_per_step'], 'ms/step', ' '.join(f'{k}={v:.1f}' for k,v in sorted(r.items()) if re.search(os.environ.get('KFILTER','gconv'), k)))"
done; done
