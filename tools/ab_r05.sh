# same-box A/B: round-5 final library (mesh_to_sdf_amd/ab/libm2s_r05.so, built from commit d1292a2) vs this tree, through bench.py
O=gpurun_out/r06_ab_r05; mkdir -p $O
R05=$PWD/mesh_to_sdf_amd/ab/libm2s_r05.so
line() { python - "$1" <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{j['ms_per_step']:8.4f} ms per step  {j['value']:10.1f} {j['unit']}  phases {j['phases_ms']}")
PY
}
for rep in 1 2 3; do
  for cfg in 0 3 2 4 5; do
    a="--config $cfg"; [ $cfg = 0 ] && a=""
    M2S_LIB=$R05 python bench.py $a --steps 20 --warmup 3 --no-cpu-baseline --no-live-pmc > $O/tmp.json 2>/dev/null; echo "config $cfg round-5 library: $(line $O/tmp.json)"
    python bench.py $a --steps 20 --warmup 3 --no-cpu-baseline --no-live-pmc > $O/tmp.json 2>/dev/null; echo "config $cfg this tree      : $(line $O/tmp.json)"
  done
done > $O/ab.txt 2>&1
cat $O/ab.txt
