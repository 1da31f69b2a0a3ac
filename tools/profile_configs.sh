#!/bin/bash
# Counter evidence for the other BASELINE configs on one GPU: bench line, kernel trace, PMC passes per config.
# usage: tools/profile_configs.sh <tag> [configs...]    (writes gpurun_out/<tag>/cfgN/...)
set -u
TAG=${1:-r02_cfg}; shift
CFGS=${*:-"2 3 4 5"}
export TMPDIR=/tmp
for C in $CFGS; do
  OUT=$PWD/gpurun_out/$TAG/cfg$C
  mkdir -p $OUT
  python bench.py --config $C --steps 10 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err
  BENCH="python $PWD/bench.py --config $C --steps 4 --warmup 1 --no-cpu-baseline --no-live-pmc"
  ( cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- $BENCH > $OUT/trace.log 2>&1 )
  DB=$(find $OUT/trace -name "*.db" | head -1)
  python tools/rocpd_summary.py $DB > $OUT/kernel_stats.md 2>> $OUT/trace.log
  i=0
  for P in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS"; do
    i=$((i+1))
    ( cd /tmp && timeout 600 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/pmc_$i -o p -- $BENCH > $OUT/pmc_$i.log 2>&1 )
  done
  python tools/pmc_summary.py $OUT k_packet > $OUT/pmc_k_packet.txt 2>&1
  M2S_LIB=$PWD/mesh_to_sdf_amd/libm2s_stats.so M2S_STATS=1 python bench.py --config $C --steps 1 --warmup 0 --no-cpu-baseline --no-live-pmc 2>&1 | grep "m2s stats" | head -12 > $OUT/stats.txt
  find $OUT -name "*.db" -delete
  rm -rf $OUT/pmc_? $OUT/trace
done
du -sh $PWD/gpurun_out/$TAG
