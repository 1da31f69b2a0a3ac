export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_pmc_jfa
mkdir -p $OUT
CMD="python $GRAFT_REPO_ROOT/tools/exp_ab.py sheet-100k 1024 Normal"
i=0
for P in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" "TCC_HIT_sum TCC_MISS_sum" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  ( cd /tmp && REPS=2 timeout 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/pmc_$i -o p -- $CMD > $OUT/pmc_$i.log 2>&1 )
done
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $OUT k_jfa_pass32 > $OUT/pmc_k_jfa.txt 2>&1
cat $OUT/pmc_k_jfa.txt
rm -rf $OUT/pmc_?
