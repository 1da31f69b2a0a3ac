set -u
OUT=$PWD/gpurun_out/sqc2; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
BENCH="python $PWD/bench.py --steps 4 --warmup 1 --no-cpu-baseline"
i=0
for C in "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES" "SQC_TC_REQ SQC_DCACHE_MISSES_DUPLICATE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  ( cd /tmp && timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$i -o p -- $BENCH > $OUT/pmc_$i.log 2>&1 ) || tail -3 $OUT/pmc_$i.log
done
python tools/pmc_summary.py $OUT k_packet
