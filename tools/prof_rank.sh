#!/bin/bash
# Kernel durations of one emulated rank step: tools/prof_rank.sh <tag> <mesh> <rank> [extra env assignments...]
set -u
TAG=$1; MESH=$2; RANK=$3; shift 3
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
SUF=$(echo "$*" | tr ' =' '__')
CMD="python $PWD/tools/exp_rank_step.py --world 8 --partition interleaved --modes none --mesh $MESH --ranks $RANK --iters 6"
( cd /tmp && rocprofv3 --kernel-trace -d $OUT/trace_${MESH}_r${RANK}_$SUF -o t -- $CMD > $OUT/rank_${MESH}_r${RANK}_$SUF.log 2>&1 )
DB=$(find $OUT/trace_${MESH}_r${RANK}_$SUF -name "*.db" | head -1)
python tools/timeline.py $DB --last 40 > $OUT/timeline_${MESH}_r${RANK}_$SUF.txt 2>&1
find $OUT -name "*.db" -size +20M -delete
