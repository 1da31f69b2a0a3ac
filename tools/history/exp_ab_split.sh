#!/bin/bash
python tools/exp_split.py blob-100k 96 128 160 192 256 2>&1 | grep -v amdgpu
python tools/exp_split.py blob-1M 256 2>&1 | grep -v amdgpu
python tools/exp_split_rank.py blob-1M 3 2>&1 | grep -v amdgpu
python tools/exp_split_rank.py blob-1M 0 2>&1 | grep -v amdgpu
python tools/exp_split_rank.py blob-100k 3 2>&1 | grep -v amdgpu
