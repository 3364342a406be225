#!/bin/bash
# Everything a node with >= 2 MI355X needs to turn SURVEY.md section 8e / VERDICT r4 item 8 into MEASUREMENTS, in one command:
#     bash tools/multi_gpu.sh [out dir]           (default gpurun_out/scale)
#   1. the RCCL parity test (2 ranks == 1 rank at the global batch; tests/test_gpu_rccl.py -- skipped on 1-GPU boxes, i.e. everywhere so far)
#   2. bench.py at N = 1, 2, 4, 8 exactly as the driver launches it (one process per GPU over RCCL, weak scaling: 32 windows per GPU),
#      200 steps; every JSON line carries config.allreduce = {us (collective alone), exposed_us (step - step without the collective, two-graph
#      form), captured_in_graph, capture_probe (verdict of the watchdogged RCCL-capture probe), placement}
#   3. one summary line per N: windows/s, ms/step, scaling vs N = 1, all-reduce us / exposed us, which form ran
# STGCN_CAPTURE=off forces the two-graph form (eager all-reduce between the graphs), =auto (default) lets the probe decide.
set -u
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd "$(dirname "$0")/.."
OUT=${1:-gpurun_out/scale}
mkdir -p "$OUT"
NG=$(python -c "import torch; print(torch.cuda.device_count())")
echo "GPUs visible: $NG"
if [ "$NG" -ge 2 ]; then
    timeout 900 python -m pytest tests/test_gpu_rccl.py -m gpu -q -p no:cacheprovider > "$OUT/pytest_rccl.log" 2>&1; echo "rccl test exit $?"; tail -3 "$OUT/pytest_rccl.log"
fi
for N in 1 2 4 8; do
    [ "$N" -le "$NG" ] || continue
    if [ "$N" -eq 1 ]; then
        timeout 900 python bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --no-gpu-baseline --no-side-configs --no-secondary > "$OUT/bench_n1.json" 2> "$OUT/bench_n1.err"
    else
        timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29500 + N)) bench.py --gpus $N --steps 200 --warmup 20 \
            --capture-collective ${STGCN_CAPTURE:-auto} > "$OUT/bench_n$N.json" 2> "$OUT/bench_n$N.err"
    fi
    echo "N=$N exit $?"
done
python - "$OUT" <<'PY'
import json, os, sys
out, base = sys.argv[1], None
for n in (1, 2, 4, 8):
    f = os.path.join(out, f"bench_n{n}.json")
    if not os.path.exists(f):
        continue
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f"N={n}: no JSON line ({e})"); continue
    base = base or d["value"]
    ar = d["config"].get("allreduce", {})
    print(f"N={n}: {d['value']:.0f} windows/s, {d['ms_per_step']} ms/step, x{d['value'] / base:.2f} of N=1; all-reduce {ar.get('us')} us alone, "
          f"{ar.get('exposed_us')} us exposed, in graph: {ar.get('captured_in_graph')} ({ar.get('capture_probe')})")
PY
