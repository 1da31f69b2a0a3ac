#!/bin/bash
# Timing experiment: the build's kernels with parts left out (M2S_DBG_SKIP bits), per-kernel average from a kernel trace.
# usage: tools/exp_build_skip.sh <tag> <mesh> "<skip values>"
set -u
TAG=${1:-r05_skip}; MESH=${2:-blob-100k}; VALS=${3:-"0 1 2 4 8 16 32 128 256 512"}
ROOT=$PWD; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for V in $VALS; do
  ( cd /tmp && M2S_DBG_SKIP=$V timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/t_$V -o t -- python $ROOT/tools/exp_build.py --meshes $MESH --iters 10 > $OUT/log_$V.txt 2>&1 )
  DB=$(find $OUT/t_$V -name "*.db" | head -1)
  echo "== skip $V"; python tools/rocpd_summary.py $DB | grep "m2s::" | awk -F'|' '{printf "%-40s %8s\n", substr($2,1,40), $5}'
  rm -rf $OUT/t_$V
done > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
