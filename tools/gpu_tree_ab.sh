#!/bin/bash
# alternating bench.py runs of two source trees (the repo root and a copy of an older commit in ab_base/), same GPU box
set -u
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
for tree in ab_base .; do
  (cd $tree && timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d.get('roofline',{}).get('per_kernel_us_per_step',{})
print('[$tree]', d['value'], 'windows/s', d['ms_per_step'], 'ms/step', ' '.join(f'{k}={v:.1f}' for k,v in sorted(r.items()) if 'gconv' in k))")
done; done
