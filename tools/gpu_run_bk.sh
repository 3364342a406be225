#!/bin/bash
# A/B of the bf16 operator GEMM's pipeline step depth (STGCN_GEMM_BF16_BK = 64 | 32) on C5, after the gctile GPU tests with BK 32
set -u
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd "$GRAFT_REPO_ROOT"
OUT="$GRAFT_REPO_ROOT/gpurun_out/${1:-bk}"
mkdir -p $OUT
STGCN_GEMM_BF16_BK=32 timeout 300 python -m pytest tests/test_gpu_gctile.py -m gpu -q -p no:cacheprovider -k "bf16" > $OUT/pytest_gctile_bk32.log 2>&1
echo "pytest(bk32) exit $?"; tail -2 $OUT/pytest_gctile_bk32.log | cut -c1-200
for bk in 64 32; do
  STGCN_GEMM_BF16_BK=$bk timeout 200 python tools/gpu_side_configs.py c5 --steps 3 --precision bf16 bf16x3 > $OUT/side_bk$bk.jsonl 2> $OUT/side_bk$bk.err
  python - "$OUT/side_bk$bk.jsonl" $bk <<'PY'
import json, sys
for line in open(sys.argv[1]):
    try:
        d = json.loads(line)
    except ValueError:
        continue
    print("BK", sys.argv[2], d.get("operator_products"), d["ms_per_step"], "ms/step", {k: (v["avg_us"], v["algorithmic_tflops"], v["frac_of_mfma_peak"]) for k, v in d.get("operator_gemm", {}).items()})
PY
done
