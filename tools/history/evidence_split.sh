#!/bin/bash
# Round-4 evidence for the split walk and the small-query path: tools/evidence_split.sh <tag>   (writes gpurun_out/<tag>/)
TAG=${1:-r04_split_v1}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
T="timeout 300"
$T python tools/exp_split.py blob-100k 64 96 128 160 192 256 384 2>&1 | grep -v amdgpu > $OUT/split_grids_100k.txt
$T python tools/exp_split.py blob-1M 128 256 2>&1 | grep -v amdgpu > $OUT/split_grids_1m.txt
$T python tools/exp_split_rank.py blob-1M 0,1,2,3,4,5,6,7 2>&1 | grep -v amdgpu > $OUT/split_ranks_1m.txt
$T python tools/exp_split_rank.py blob-100k 0,3,7 2>&1 | grep -v amdgpu > $OUT/split_ranks_100k.txt
for G in 96 128 256; do M2S_SPLIT=1 M2S_SPLIT_REPORT=1 $T python tools/exp_grid.py --grid $G --reps 2 2>&1 | grep -v amdgpu | tail -n 2; done > $OUT/split_report.txt
M2S_SPLIT=1 M2S_SPLIT_REPORT=1 $T python tools/exp_grid.py --mesh blob-1M --grid 256 --reps 2 2>&1 | grep -v amdgpu | tail -n 2 >> $OUT/split_report.txt
$T python tools/exp_small_queries.py 2>&1 | grep -v amdgpu > $OUT/small_queries.txt
$T python tools/exp_rank_step.py --world 8 --partition interleaved --modes none 2>&1 | grep -v amdgpu > $OUT/rank_step_100k.txt
$T python tools/exp_rank_step.py --world 8 --partition interleaved --modes none --mesh blob-1M 2>&1 | grep -v amdgpu > $OUT/rank_step_1m.txt
$T python tools/exp_grids.py blob-100k 64 96 128 160 192 256 384 512 2>&1 | grep -v amdgpu > $OUT/grids_100k.txt
$T python tools/exp_build.py 2>&1 | grep -v amdgpu > $OUT/build.txt
tail -n 3 $OUT/*.txt
