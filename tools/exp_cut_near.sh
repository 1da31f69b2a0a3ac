# emission radius of a cut-list entry next to the surface (M2S_CUT_NEAR, brick radii) by grid size: whole calls, best of 4 (tools/exp_ab.py)
for c in "blob-100k 256 Raycast" "blob-100k 192 Raycast" "blob-100k 512 Raycast" "blob-1M 512 Raycast" "sheet-100k 512 Normal"; do
  for near in 2 1.5 1 0.7; do M2S_CUT_NEAR=$near python tools/exp_ab.py $c 2>&1 | grep -v amdgpu | sed "s/^default lib/CUT_NEAR=$near/"; done
done
