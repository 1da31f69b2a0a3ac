#!/bin/bash
# PMC passes over the LBVH build alone (tools/exp_build.py): per-kernel instruction counts and wait fractions.
# usage: tools/pmc_build.sh <tag> [mesh]
set -u
TAG=${1:-r03_pmc_build}
MESH=${2:-blob-100k}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python $PWD/tools/exp_build.py --meshes $MESH --builds 1 --iters 6"
i=0
for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  ( cd /tmp && timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$i -o p -- $CMD > $OUT/pmc_$i.log 2>&1 )
done
for K in k_tri_setup k_sort_tiles k_sort_rank k_sort_buckets k_roots_from_keys k_treelet_lanes k_hierarchy k_emit k_node_ext; do
  echo "== $K"; python tools/pmc_summary.py $OUT $K
done > $OUT/summary.txt 2>&1
find $OUT -name "*.db" -delete
cat $OUT/summary.txt
