import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from mesh_to_sdf_amd import Grid, SignMethod, Topology, generate_grid_sdf, meshes
for mesh, n in (("blob-100k", 512), ("blob-100k", 256), ("blob-100k", 128), ("blob-1M", 512)):
    v, idx = meshes.named(mesh)
    lo, hi = meshes.extended_bbox(v, 0.1)
    dv = torch.as_tensor(v, device="cuda")
    topo = Topology.TriangleList(torch.as_tensor(idx.astype(np.int64), device="cuda").to(torch.int32))
    grid = Grid.from_bounding_box(lo, hi, [n, n, n])
    print(f"== {mesh} {n}^3", file=sys.stderr, flush=True)
    generate_grid_sdf(dv, topo, grid, SignMethod.Raycast)
