#!/bin/bash
# Round-6 evidence refresh on the GPU box (everything the docs quote): usage: bash tools/evidence_r06.sh <version>
V=${1:-v1}
O=$PWD/gpurun_out/r06_$V
mkdir -p $O
bash tools/profile_all.sh r06_$V > $O/profile_all.log 2>&1
python tools/exp_configs.py > gpurun_out/configs.txt 2>&1
python tools/exp_suzanne.py 32 64 256 > gpurun_out/suzanne.txt 2>&1
python tools/exp_grids.py blob-100k 32 48 64 96 128 192 256 > $O/grids_100k.txt 2>&1
python tools/exp_grids.py blob-11k 32 64 100 128 > $O/grids_11k.txt 2>&1
python tools/exp_rank_step.py --world 8 --modes none > $O/rank_step_100k.txt 2>&1
python tools/exp_rank_step.py --world 8 --modes none --mesh blob-1M > $O/rank_step_1m.txt 2>&1
python tools/exp_slab_cull.py blob-100k 512 8 > $O/slab_cull.txt 2>&1
python tools/exp_slab_cull.py blob-1M 512 8 >> $O/slab_cull.txt 2>&1
python tools/exp_criterion_shapes.py > $O/criterion_shapes.txt 2>&1
python tools/exp_first_call.py > $O/first_call.txt 2>&1
python tools/exp_build.py --builds 1 > $O/build.txt 2>&1
bash tools/profile_configs.sh r06_cfg_$V 2 3 4 5 > $O/profile_configs.log 2>&1
tail -c 600 $O/bench.json; cat $O/grids_100k.txt $O/rank_step_100k.txt $O/rank_step_1m.txt $O/slab_cull.txt | grep -v amdgpu
