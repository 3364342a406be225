#!/bin/bash
# quick check of the bf16 operator GEMMs: the gctile GPU tests, then C5 with bf16 / bf16x3
set -u
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd "$GRAFT_REPO_ROOT"
OUT="$GRAFT_REPO_ROOT/gpurun_out/${1:-bf16}"
mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_gctile.py -m gpu -q -p no:cacheprovider > $OUT/pytest_gctile.log 2>&1
echo "pytest exit $?"; tail -2 $OUT/pytest_gctile.log | cut -c1-200; grep -E "^(FAILED|ERROR)" $OUT/pytest_gctile.log | cut -c1-300 | head
timeout 300 python tools/gpu_side_configs.py c5 --steps 3 --precision bf16 bf16x3 > $OUT/side_configs.jsonl 2> $OUT/side_configs.err
echo "exit $?"; tail -2 $OUT/side_configs.err
python - "$OUT/side_configs.jsonl" <<'PY'
import json, sys
for line in open(sys.argv[1]):
    try:
        d = json.loads(line)
    except ValueError:
        continue
    print(d["config"], d.get("operator_products"), d["ms_per_step"], "ms/step", {k: (v["avg_us"], v["algorithmic_tflops"], v["frac_of_mfma_peak"]) for k, v in d.get("operator_gemm", {}).items()})
PY
