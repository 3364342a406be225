#!/bin/bash
# build-time variants of the library, per-kernel profile of each (experiments; summaries only)
set -u
cd "$GRAFT_REPO_ROOT"
OUT="$GRAFT_REPO_ROOT/gpurun_out/${1:-var}"
mkdir -p $OUT
run() {  # name, lib, env...
  name=$1; lib=$2; shift 2
  env STGCN_AMD_LIB=$lib "$@" timeout 300 python tools/gpu_report.py --quick > $OUT/$name.jsonl 2> $OUT/$name.err
  python - "$OUT/$name.jsonl" "$name" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    d=json.loads(l)
    if 'fused_step_ms' in d: print(sys.argv[2], "step_ms", d['fused_step_ms'])
    if 'kernel_profile' in d:
        kp=d['kernel_profile']
        print(sys.argv[2], "sum_us", d['sum_us'], " ".join(f"{k}={v['us_per_step']:.0f}" for k,v in sorted(kp.items(), key=lambda kv:-kv[1]['us_per_step'])[:14]))
PY
}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DSTGCN_BACKEND_NAME=\"hip-gfx950\""
i=0
for v in "$@"; do
  [ $i -eq 0 ] && { i=1; continue; }
  name=$(echo "$v" | tr -c 'A-Za-z0-9=\n' '_')
  hipcc $FLAGS $v stgcn_amd/csrc/stgcn_capi.hip -o /tmp/lib_$name.so 2>/dev/null || echo "build failed: $v"
  run "$name" /tmp/lib_$name.so
done
