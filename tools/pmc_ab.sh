#!/bin/bash
# Memory-side traffic and wait share of the walk for library builds on a command:
#   tools/pmc_ab.sh <tag> "<lib paths, '-' = the default library>" -- <command...>
TAG=$1; LIBS=$2; shift 3
ROOT=$PWD; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
n=0
for L in $LIBS; do
  n=$((n+1))
  if [ "$L" == "-" ]; then unset M2S_LIB; else export M2S_LIB=$ROOT/$L; fi
  i=0
  for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    ( cd /tmp && timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/lib${n}_pmc_$i -o p -- "$@" > $OUT/lib${n}_pmc_$i.log 2>&1 )
  done
done
python3 - $OUT "$LIBS" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for n, lib in enumerate(sys.argv[2].split(), 1):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{out}/lib{n}_pmc_*/**/*counter_collection.csv", recursive=True):
        per = collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("void m2s::(anonymous namespace)::", "").split("(")[0][:48]
            per[(k, r["Counter_Name"], r["Dispatch_Id"])] += float(r["Counter_Value"])
        for (k, c, d), v in per.items(): acc[k][c].append(v)
    for k in sorted(acc, key=lambda k: -sum(sum(v) for v in acc[k].values()))[:3]:
        print(lib, k, {c: (f"{sum(v) / len(v) / 1e3:.1f} MB" if c in ("FETCH_SIZE", "WRITE_SIZE") else f"{sum(v) / len(v):.4g}") + f" (x{len(v)})" for c, v in sorted(acc[k].items())}, "(FETCH_SIZE as reported: KB, half-count on gfx950)")
PY
find $OUT -name "*.db" -delete
