#!/bin/bash
# shader clock / power while the graph-replayed C2 step runs for ~10 s (is the chip power-limited under this load?)
OUT=$1
( for i in $(seq 1 60); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo; sleep 0.25; done ) > $OUT/clocks.txt &
SMI=$!
python bench.py --steps 20000 --warmup 50 --no-cpu-baseline --no-gpu-baseline --no-profile > $OUT/bench_long.json 2>/dev/null
sleep 1
kill $SMI 2>/dev/null
python -c "
import json; d=json.loads(open('$OUT/bench_long.json').read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'])"
sed -n '1,3p;20,30p;55,60p' $OUT/clocks.txt | cut -c1-200
