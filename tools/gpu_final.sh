#!/bin/bash
# final pass of a round: gpu_final.sh <tag>  (e.g. r3-66).  All GPU tests + smoke + the C2 replay trace, then the FETCH / WRITE counter passes of
# the three configs (copied to profiles/<tag>_pmc_traffic*.json ON THE BOX, which is where bench.py's roofline.traffic looks), then the
# bench lines that get committed: C2 fp32 headline with both baselines, C3 bf16, C5 bf16, and C2 again as the driver invokes it (20 steps, 5 warm-up).
cd $GRAFT_REPO_ROOT
TAG=${1:?tag}
O=gpurun_out/$TAG
bash tools/gpu_round.sh $TAG "test smoke trace pmc" | cut -c1-400
BENCH_ARGS="--config c3" bash tools/gpu_round.sh ${TAG}c3 "pmc" > /dev/null
BENCH_ARGS="--config c5" PMC_STEPS=2 bash tools/gpu_round.sh ${TAG}c5 "pmc" > /dev/null
cp $O/pmc_traffic.json profiles/${TAG}_pmc_traffic.json
cp ${O}c3/pmc_traffic.json profiles/${TAG}_pmc_traffic_c3_bf16.json
cp ${O}c5/pmc_traffic.json profiles/${TAG}_pmc_traffic_c5_bf16.json
python bench.py > $O/bench_full.json 2> $O/bench_full.err
python bench.py --config c3 --no-cpu-baseline --no-side-configs > $O/bench_c3.json 2> $O/bench_c3.err
python bench.py --config c5 --no-cpu-baseline --no-side-configs > $O/bench_c5.json 2> $O/bench_c5.err
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_style.json 2> /dev/null
for F in bench_full bench_c3 bench_c5 bench_driver_style; do
  python -c "
import json; d=json.load(open('$O/$F.json')); r=d['roofline']
print('$F', d['ms_per_step'], d['value'], d['config'].get('secondary_bwd_bf16x3',{}).get('value'), r['kernel'], r['frac'], r.get('traffic_source'), r.get('stblock_traffic_bytes'))"
done
