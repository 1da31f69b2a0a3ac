#!/bin/bash
# HBM-side traffic of k_packet for two library builds on a command: tools/pmc_ab.sh <tag> <libA> <libB> -- <command...>
TAG=$1; A=$2; B=$3; shift 4
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for L in $A $B; do
  export M2S_LIB=$PWD/mesh_to_sdf_amd/libm2s_$L.so
  i=0
  for C in "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    ( cd /tmp && timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/${L}_pmc_$i -o p -- "$@" > $OUT/${L}_pmc_$i.log 2>&1 )
  done
done
python3 - $OUT $A $B <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for lib in sys.argv[2:]:
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{out}/{lib}_pmc_*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("void m2s::(anonymous namespace)::", "").split("(")[0][:48]
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in sorted(acc, key=lambda k: -sum(sum(v) for v in acc[k].values()))[:6]:
        print(lib, k, {c: f"{sum(v) / len(v) / 1e3:.1f} MB (x{len(v)})" if c in ("FETCH_SIZE", "WRITE_SIZE") else round(sum(v) / len(v)) for c, v in acc[k].items()}, "(FETCH_SIZE as reported: KB, half-count on gfx950)")
PY
