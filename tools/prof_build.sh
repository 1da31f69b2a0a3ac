#!/bin/bash
# Kernel trace of the LBVH build alone (tools/exp_build.py) per mesh: per-kernel table and the timeline of the last build.
# usage: tools/prof_build.sh <tag> [meshes]     (writes gpurun_out/<tag>/build_<mesh>.{md,timeline.txt} and build.txt)
set -u
TAG=${1:-r05_build}
MESHES=${2:-blob-11k,blob-100k,blob-1M}
ROOT=$PWD
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python tools/exp_build.py --meshes $MESHES --builds 1 --iters 30 > $OUT/build.txt 2>&1
for M in ${MESHES//,/ }; do
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace_$M -o t -- python $ROOT/tools/exp_build.py --meshes $M --builds 1 --iters 8 > $OUT/trace_$M.log 2>&1 )
  DB=$(find $OUT/trace_$M -name "*.db" | head -1)
  python tools/rocpd_summary.py $DB > $OUT/build_$M.md 2>> $OUT/trace_$M.log
  python tools/timeline.py $DB --last 28 > $OUT/build_$M.timeline.txt 2>> $OUT/trace_$M.log
  rm -rf $OUT/trace_$M
done
cat $OUT/build.txt
