O=gpurun_out/r06_cutc2; mkdir -p $O
for c in "sheet-100k 1024 Normal" "blob-100k 1024 Raycast" "blob-100k 512 Raycast" "blob-1M 512 Raycast"; do
  M2S_CUT_COARSE=0 python tools/exp_ab.py $c 2>&1 | sed "s/^default lib/one level        /"
  for cap in 0 48 96 160; do M2S_CUT_COARSE=1 M2S_CUT_COARSE_CAP=$cap python tools/exp_ab.py $c 2>&1 | sed "s/^default lib/two levels cap=$cap/"; done
done > $O/cap.txt 2>&1
grep -v amdgpu.ids $O/cap.txt
