O=gpurun_out/r06_group2; mkdir -p $O
for tw in 8192 16384 32768; do
  echo "## M2S_GROUP_TARGET_WAVES=$tw"
  M2S_GROUP_TARGET_WAVES=$tw python tools/exp_groups.py blob-100k 32 48 64 80 96 112
  M2S_GROUP_TARGET_WAVES=$tw python tools/exp_groups.py blob-11k 32 48 64 80 100
  M2S_GROUP_TARGET_WAVES=$tw python tools/exp_groups.py blob-1M 48 64 80
done > $O/sweep.txt 2>&1
grep -v amdgpu.ids $O/sweep.txt
echo "## cut lists from fewer packets on, two levels (blob-100k): M2S_CUT_MIN_PACKETS x M2S_CUT_COARSE" > $O/cutmin.txt
for n in 128 160 192 224 256; do
  for cm in 100000 20000; do for cc in 0 1; do
    M2S_CUT_MIN_PACKETS=$cm M2S_CUT_COARSE=$cc python tools/exp_ab.py blob-100k $n Raycast 2>&1 | grep -v amdgpu | sed "s/^default lib/CUT_MIN=$cm COARSE=$cc/"
  done; done
done >> $O/cutmin.txt 2>&1
cat $O/cutmin.txt
