#!/bin/bash
# sweep of the cut-list parameters (block size, emission radii, ranges per block) on the headline call
for lib in hip c15; do
  export M2S_LIB=$PWD/mesh_to_sdf_amd/libm2s_$lib.so
  for near in 1 1.5 2 3; do for far in 0.2 0.333 0.5; do
    echo -n "lib=$lib near=$near far=$far: "
    M2S_CUT_NEAR=$near M2S_CUT_FAR=$far REPS=3 timeout 100 python tools/exp_ab.py 2>&1 | grep -o "seed.*total [0-9.]* ms"
  done; done
done
