#!/bin/bash
# One-call evidence refresh on the GPU box: bench line (with CPU baseline), kernel trace, PMC passes.
# usage: tools/profile_all.sh <tag>     (writes gpurun_out/<tag>/...)
set -u
TAG=${1:-r01_v5}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
BENCH="python $PWD/bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-live-pmc"
( cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- $BENCH > $OUT/trace.log 2>&1 )
DB=$(find $OUT/trace -name "*.db" | head -1)
python tools/rocpd_summary.py $DB > $OUT/kernel_stats.md 2>> $OUT/trace.log
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS"; do
  i=$((i+1))
  ( cd /tmp && timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$i -o p -- $BENCH > $OUT/pmc_$i.log 2>&1 )
done
python tools/pmc_summary.py $OUT k_packet > $OUT/pmc_k_packet.txt 2>&1
python tools/pmc_summary.py $OUT k_cut > $OUT/pmc_k_cut.txt 2>&1
find $OUT -name "*.db" -size +20M -delete
du -sh $OUT
