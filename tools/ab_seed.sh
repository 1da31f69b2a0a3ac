# usage: bash tools/ab_seed.sh   — same-box A/B: round-5 distance.hip (ab/libm2s_base.so) vs this tree; M2S_SEED_COARSE = 0 / 2 / 4 / 8
O=gpurun_out/r06_seed2; mkdir -p $O
AB=$PWD/mesh_to_sdf_amd/ab/libm2s_base.so
for rep in 1 2; do
for c in "blob-100k 512 Raycast" "blob-100k 256 Raycast" "sheet-100k 1024 Normal"; do
  M2S_LIB=$AB python tools/exp_ab.py $c
  for k in 0 2 4 8; do M2S_SEED_COARSE=$k python tools/exp_ab.py $c 2>&1 | sed "s/^default lib/SEED_COARSE=$k/"; done
done
done > $O/ab.txt 2>&1
grep -v amdgpu.ids $O/ab.txt
M2S_STATS=1 python tools/exp_stats.py blob-100k 512 Raycast > $O/stats_512.txt 2>&1
M2S_STATS=1 python tools/exp_stats.py sheet-100k 1024 Normal > $O/stats_c5.txt 2>&1
grep "k_cut" $O/stats_512.txt $O/stats_c5.txt
