#!/bin/bash
# A/B of the fp32 operator GEMM's column extent (STGCN_GEMM_NTW) on C5
set -u
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd "$GRAFT_REPO_ROOT"
OUT="$GRAFT_REPO_ROOT/gpurun_out/${1:-ntw}"
mkdir -p $OUT
for n in 0 3 4 5; do
  STGCN_GEMM_NTW=$n STGCN_LAUNCH_LOG=$OUT/launch_$n.log timeout 200 python tools/gpu_side_configs.py c5 --steps 2 --precision fp32 > $OUT/side_$n.jsonl 2> $OUT/side_$n.err
  python - "$OUT/side_$n.jsonl" $n <<'PY'
import json, sys
for line in open(sys.argv[1]):
    try:
        d = json.loads(line)
    except ValueError:
        continue
    print("NTW", sys.argv[2], d["ms_per_step"], "ms/step", {k: (v["avg_us"], v["algorithmic_tflops"], v["frac_of_mfma_peak"]) for k, v in d.get("operator_gemm", {}).items()})
PY
  grep gso_gemm $OUT/launch_$n.log | sort | uniq -c | head -4
  rm -f $OUT/launch_$n.log
done
