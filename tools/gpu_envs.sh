#!/bin/bash
# run-time (environment) variants of ONE build of the library, per-kernel profile of each:  gpu_envs.sh <tag> "A=1 B=2" "A=3" ...
set -u
cd "$GRAFT_REPO_ROOT"
OUT="$GRAFT_REPO_ROOT/gpurun_out/${1:-env}"
mkdir -p $OUT
shift
i=0
for v in "$@"; do
  name="e$i"; i=$((i+1))
  env $v timeout 300 python tools/gpu_report.py --quick > $OUT/$name.jsonl 2> $OUT/$name.err
  python - "$OUT/$name.jsonl" "$v" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    d=json.loads(l)
    if 'kernel_profile' in d:
        kp=d['kernel_profile']
        print("[%s]" % sys.argv[2], "sum_us", d['sum_us'], " ".join(f"{k}={v['us_per_step']:.1f}" for k,v in sorted(kp.items()) if __import__('re').search(__import__('os').environ.get('KFILTER', 'tconv_fwd|bwd_data'), k)))
PY
done
