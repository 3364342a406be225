#!/bin/bash
# Short GPU pass: parity suite + side configs (C5 with the three operator precisions, C3); optional headline bench.
#   gpurun -- bash tools/gpu_run_side.sh <tag> [bench]
set -u
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd "$GRAFT_REPO_ROOT"
OUT="$GRAFT_REPO_ROOT/gpurun_out/${1:-side}"
mkdir -p $OUT
timeout 600 python -m pytest tests -m gpu -q --maxfail=40 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log
tail -3 $OUT/pytest_gpu.log | cut -c1-300
grep -E "^(FAILED|ERROR)" $OUT/pytest_gpu.log | cut -c1-400 | head -20
timeout 400 python tools/gpu_side_configs.py c5 c3 --steps 5 --precision fp32 bf16x3 bf16 > $OUT/side_configs.jsonl 2> $OUT/side_configs.err
echo "side configs exit $?"; tail -3 $OUT/side_configs.err
python - "$OUT/side_configs.jsonl" <<'PY'
import json, sys
for line in open(sys.argv[1]):
    try:
        d = json.loads(line)
    except ValueError:
        continue
    pk = d["per_kernel_ms_per_step"]
    print(d["config"], d.get("operator_products"), d["ms_per_step"], "ms/step", {k: (v["avg_us"], v["algorithmic_tflops"], v["frac_of_mfma_peak"]) for k, v in d.get("operator_gemm", {}).items()})
    print("   top:", sorted(((round(v, 3), k) for k, v in pk.items()), reverse=True)[:10])
PY
if [ "${2:-}" = "bench" ]; then
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
echo "bench exit $?"; cut -c1-330 $OUT/bench.json
fi
