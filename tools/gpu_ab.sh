#!/bin/bash
# within-call A/B of env-knob settings using the shipped library: alternating quick reports
set -u
cd "$GRAFT_REPO_ROOT"
OUT="$GRAFT_REPO_ROOT/gpurun_out/${1:-ab}"; mkdir -p $OUT
shift
i=0
for rep in 1 2; do
for v in "$@"; do
  name=$(echo "$v" | tr -c 'A-Za-z0-9=\n' '_')_$rep
  env $v timeout 300 python tools/gpu_report.py --quick > $OUT/$name.jsonl 2> $OUT/$name.err
  python - "$OUT/$name.jsonl" "$name" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    d=json.loads(l)
    if 'kernel_profile' in d:
        kp=d['kernel_profile']
        print(sys.argv[2], "sum_us", d['sum_us'], " ".join(f"{k}={v['us_per_step']:.1f}" for k,v in sorted(kp.items(), key=lambda kv:-kv[1]['us_per_step']) if k in ("tconv_fwd.tc1@0","tconv_fwd.tc1@1","align_gate_bwd@0","gconv_fwd@0","tconv_bwd_weight.tc1@1")))
PY
done; done
