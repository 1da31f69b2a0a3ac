#!/usr/bin/env python3
"""A few steps of ONE emulated rank (slab of `--world`/`--rank`), nothing else: the command rocprofv3 traces for the rank timeline."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ap = argparse.ArgumentParser()
ap.add_argument("--world", type=int, default=8)
ap.add_argument("--rank", type=int, default=3)
ap.add_argument("--grid", type=int, default=512)
ap.add_argument("--mesh", default="blob-100k")
ap.add_argument("--sign", default="Raycast")
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--partition", default="interleaved", choices=["contiguous", "interleaved"])
args = ap.parse_args()
import torch  # noqa: E402

from mesh_to_sdf_amd import Grid, SignMethod, Topology, generate_grid_sdf, interleaved_slab, meshes, slab_bounds  # noqa: E402

v, idx = meshes.named(args.mesh)
lo, hi = meshes.extended_bbox(v, 0.1)
grid = Grid.from_bounding_box(lo, hi, [args.grid] * 3)
dv = torch.as_tensor(v, device="cuda")
topo = Topology.TriangleList(torch.as_tensor(idx.astype(np.int64), device="cuda").to(torch.int32))
out = torch.empty(args.grid ** 3, dtype=torch.float32, device="cuda")
xs, period = slab_bounds(args.grid, args.world, args.rank), 0
if args.partition == "interleaved" and args.world > 1:
    a, b, period = interleaved_slab(grid, args.world, args.rank)
    xs = (a, b)
for _ in range(args.iters):
    generate_grid_sdf(dv, topo, grid, SignMethod[args.sign], x_slab=xs, x_period=period, out=out)
torch.cuda.synchronize()
