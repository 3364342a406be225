#!/bin/bash
# phase stamps of the fused kernels (diagnostic build): cycles per phase, and wall-clock workgroup timelines
OUT=$1
STGCN_PHASE_KIDS=${STGCN_PHASE_KIDS:-8,9} python tools/gpu_phases.py > $OUT/phases.txt 2> $OUT/phases.err
STGCN_EXTRA_FLAGS=-DSTGCN_PHASE_WALL STGCN_PHASE_KIDS=${STGCN_PHASE_KIDS:-8,9} python tools/gpu_phases.py > $OUT/phases_wall.txt 2>> $OUT/phases.err
cat $OUT/phases.txt $OUT/phases_wall.txt
