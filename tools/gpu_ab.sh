#!/bin/bash
# A/B runs of bench.py on a GPU box: gpu_ab.sh <out dir under the repo> <config: c2|c3|c5> <name>="<ENV=.. ENV=..>" ...
# (every variant: 200 replayed steps, per-kernel us/step from the hipEvent pass; one line per variant on stdout, the JSON lines under <out dir>)
OUT=$GRAFT_REPO_ROOT/$1; CFG=$2; shift 2
mkdir -p $OUT; cd $GRAFT_REPO_ROOT
for SPEC in "$@"; do
  NAME=${SPEC%%=*}; ENVS=${SPEC#*=}
  env $ENVS timeout 600 python bench.py --config $CFG --steps ${STEPS:-200} --warmup 20 --no-cpu-baseline --no-side-configs --no-gpu-baseline --no-secondary > $OUT/bench_${CFG}_$NAME.json 2> $OUT/bench_${CFG}_$NAME.err; echo "$CFG $NAME exit $?"
  python -c "
import json; d=json.load(open('$OUT/bench_${CFG}_$NAME.json')); pk=d['roofline']['per_kernel_us_per_step']
print('$CFG $NAME', d['ms_per_step'], d['value'], ' '.join(f'{k}={v:.1f}' for k,v in sorted(pk.items())))"
done
