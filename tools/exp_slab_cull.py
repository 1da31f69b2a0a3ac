#!/usr/bin/env python3
"""Would a slab-culled build help an 8-GPU rank?  (drop the triangles farther from the rank's x-slab than the
largest seed distance of its bricks + a brick diagonal, build the tree over the survivors.)  For every slab of the BASELINE multi-GPU
configs: the largest distance any of its voxels has to the mesh (from the computed field: every seed distance is at least that), the
radius the rule would keep, and the share of triangles inside it.      python tools/exp_slab_cull.py [mesh] [n] [world]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mesh_to_sdf_amd import Grid, SignMethod, Topology, generate_grid_sdf, meshes, slab_bounds  # noqa: E402

mesh = sys.argv[1] if len(sys.argv) > 1 else "blob-100k"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 512
world = int(sys.argv[3]) if len(sys.argv) > 3 else 8
v, idx = meshes.named(mesh)
lo, hi = meshes.extended_bbox(v, 0.1)
g = Grid.from_bounding_box(lo, hi, [n] * 3)
cs = np.asarray(g.get_cell_size(), np.float64)
first = np.asarray(g.get_first_cell(), np.float64)
dv = torch.as_tensor(v, device="cuda")
topo = Topology.TriangleList(torch.as_tensor(idx.astype(np.int64), device="cuda").to(torch.int32))
sdf = generate_grid_sdf(dv, topo, g, SignMethod.Raycast).view(n, n, n).abs()
tri = v[idx.reshape(-1, 3).astype(np.int64)].astype(np.float64)
tmin, tmax = tri.min(1), tri.max(1)
diag = float(np.linalg.norm(4 * cs))
print(f"# {mesh} ({len(tri)} triangles) in {n}^3, {world} contiguous x-slabs; brick diagonal {diag / cs[0]:.1f} cells")
for r in range(world):
    x0, x1 = slab_bounds(n, world, r)
    dmax = float(sdf[x0:x1].max())
    keep_r = dmax + diag
    blo = first + np.array([x0, 0, 0]) * cs
    bhi = first + np.array([x1 - 1, n - 1, n - 1]) * cs
    gap = np.maximum(np.maximum(blo - tmax, tmin - bhi), 0.0)           # distance of the triangle's box to the slab's box, per axis
    kept = (np.linalg.norm(gap, axis=1) <= keep_r).mean()
    inside = ((tmax[:, 0] >= blo[0]) & (tmin[:, 0] <= bhi[0])).mean()
    print(f"rank {r} layers [{x0},{x1}): farthest voxel {dmax / cs[0]:6.1f} cells from the mesh -> keep radius {keep_r / cs[0]:6.1f} cells: "
          f"{100 * kept:5.1f} % of the triangles survive the rule ({100 * inside:4.1f} % lie inside the slab's own x-range)")
