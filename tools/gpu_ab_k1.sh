#!/bin/bash
# A/B of the fused kernels at the C2 block shapes: per-kernel times, training vs eval (no Philox), fused vs stage-per-launch
OUT=$1
for blk in 0 1; do
  python tools/gpu_block_times.py --block $blk >> $OUT/block_times.jsonl 2>> $OUT/block_times.err
  python tools/gpu_block_times.py --block $blk --eval >> $OUT/block_times.jsonl 2>> $OUT/block_times.err
done
STGCN_FUSE=0 python tools/gpu_block_times.py --block 0 >> $OUT/block_times.jsonl 2>> $OUT/block_times.err
cat $OUT/block_times.jsonl
