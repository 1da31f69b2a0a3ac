#!/bin/bash
# PMC comparison of k_packet with the split walk off / on (never checking): tools/pmc_split.sh <tag>
set -u
TAG=$1
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python $PWD/tools/exp_grid.py --mesh blob-100k --grid 512 --reps 2"
for MODE in 0 1; do
  export M2S_SPLIT=$MODE M2S_SPLIT_BUDGET=1000000000
  i=0
  for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU"; do
    i=$((i+1))
    ( cd /tmp && timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/m${MODE}_pmc_$i -o p -- $CMD > $OUT/m${MODE}_pmc_$i.log 2>&1 )
  done
done
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for mode in (0, 1):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"{out}/m{mode}_pmc_*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_packet" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("mode", mode, {k: round(sum(v) / len(v)) for k, v in sorted(acc.items())})
PY
