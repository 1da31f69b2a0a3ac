#!/bin/bash
# PMC passes over k_packet of one emulated rank slab: tools/pmc_rank.sh <tag> <world> <rank>
set -u
TAG=$1; W=$2; R=$3
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python $PWD/tools/exp_one_rank.py --world $W --rank $R --iters 5"
i=0
for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES" "FETCH_SIZE"; do
  i=$((i+1))
  ( cd /tmp && timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$i -o p -- $CMD > $OUT/pmc_$i.log 2>&1 )
done
python tools/pmc_summary.py $OUT k_packet > $OUT/pmc_k_packet.txt 2>&1
rm -rf $OUT/pmc_?
cat $OUT/pmc_k_packet.txt
