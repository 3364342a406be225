#!/bin/bash
# A/B of the plane row padding (stgcn_set_gc_ld_pad) on the C5 operator GEMMs
set -u
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd "$GRAFT_REPO_ROOT"
OUT="$GRAFT_REPO_ROOT/gpurun_out/${1:-pad}"
mkdir -p $OUT
timeout 500 python tools/gpu_side_configs.py c5 --steps 3 --precision bf16 bf16x3 --ldpad ${PADS:-0 64 128 2176} > $OUT/side_configs.jsonl 2> $OUT/side_configs.err
echo "exit $?"; tail -3 $OUT/side_configs.err
python - "$OUT/side_configs.jsonl" <<'PY'
import json, sys
for line in open(sys.argv[1]):
    try:
        d = json.loads(line)
    except ValueError:
        continue
    print(d["config"], d.get("operator_products"), "pad", d.get("ld_pad"), d["ms_per_step"], "ms/step", {k: (v["avg_us"], v["algorithmic_tflops"]) for k, v in d.get("operator_gemm", {}).items()})
PY
