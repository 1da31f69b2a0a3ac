#!/bin/bash
# Kernel timeline of one grid call with the split walk: tools/prof_split.sh <tag> <mesh> <n>
set -u
TAG=$1; MESH=$2; N=$3
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python $PWD/tools/exp_grid.py --mesh $MESH --grid $N --reps 4"
M2S_SPLIT_REPORT=1 $CMD > $OUT/report_${MESH}_$N.txt 2>&1
( cd /tmp && rocprofv3 --kernel-trace -d $OUT/trace_${MESH}_$N -o t -- $CMD > $OUT/trace_${MESH}_$N.log 2>&1 )
DB=$(find $OUT/trace_${MESH}_$N -name "*.db" | head -1)
python tools/timeline.py $DB --last 24 > $OUT/timeline_${MESH}_$N.txt 2>&1
find $OUT -name "*.db" -size +20M -delete
