#!/bin/bash
# final pass of round 3: all GPU tests + smoke + the three bench lines (C2 fp32 headline with baselines, C3 bf16, C5 bf16), the C2 replay trace and
# the FETCH / WRITE counter passes of all three (profiles/r3-62_pmc_traffic*.json feed bench.py's roofline.traffic)
cd $GRAFT_REPO_ROOT
bash tools/gpu_round.sh r3-62 "test smoke benchfull trace pmc"
BENCH_ARGS="--config c3" bash tools/gpu_round.sh r3-62c3 "bench pmc"
BENCH_ARGS="--config c5" PMC_STEPS=2 bash tools/gpu_round.sh r3-62c5 "bench pmc"
