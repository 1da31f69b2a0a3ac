#!/usr/bin/env python3
"""Traversal counters of k_packet (M2S_STATS=1) for one grid call: per-packet node / pre-test / exact counts, overall and by distance band."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["M2S_STATS"] = os.environ.get("M2S_STATS", "1")
os.environ.setdefault("M2S_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mesh_to_sdf_amd", "libm2s_stats.so"))   # the counting build (make -C mesh_to_sdf_amd/csrc stats)
import torch  # noqa: E402

from mesh_to_sdf_amd import Grid, SignMethod, Topology, generate_grid_sdf, meshes  # noqa: E402

mesh = sys.argv[1] if len(sys.argv) > 1 else "blob-100k"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 512
sign = SignMethod[sys.argv[3]] if len(sys.argv) > 3 else SignMethod.Raycast
v, idx = meshes.named(mesh)
lo, hi = meshes.extended_bbox(v, 0.1)
grid = Grid.from_bounding_box(lo, hi, [n, n, n])
dv = torch.as_tensor(v, device="cuda")
topo = Topology.TriangleList(torch.as_tensor(idx.astype(np.int64), device="cuda").to(torch.int32))
print(f"# {mesh} {n}^3 {sign.name}", file=sys.stderr, flush=True)
generate_grid_sdf(dv, topo, grid, sign)
torch.cuda.synchronize()
