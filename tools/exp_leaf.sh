for L in 1 2 4 8 16; do echo "leaf_max=$L"; M2S_LEAF_MAX=$L python tools/exp_configs.py c4 2>&1 | grep -E "slab=None"; done
