#!/usr/bin/env python3
"""generate_grid_sdf timing over grid sizes: python tools/exp_grids.py [mesh] [sizes...]   (best of 5 calls, device-resident)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mesh_to_sdf_amd import Grid, M2STimings, SignMethod, Topology, generate_grid_sdf, meshes  # noqa: E402

mesh = sys.argv[1] if len(sys.argv) > 1 else "blob-100k"
sizes = [int(c) for c in sys.argv[2:]] or [32, 64, 96, 128, 192, 256]
v, idx = meshes.named(mesh)
lo, hi = meshes.extended_bbox(v, 0.1)
dv = torch.as_tensor(v, device="cuda")
topo = Topology.TriangleList(torch.as_tensor(idx.astype(np.int64), device="cuda").to(torch.int32))
for n in sizes:
    grid = Grid.from_bounding_box(lo, hi, [n, n, n])
    out = torch.empty(n ** 3, dtype=torch.float32, device="cuda")
    line = f"{mesh} {n:>4}^3:"
    for sign in (SignMethod.Raycast, SignMethod.Normal):
        best = None
        for _ in range(5):
            t = M2STimings()
            generate_grid_sdf(dv, topo, grid, sign, out=out, timings=t)
            if best is None or t.total_ms < best.total_ms:
                best = t
        line += f"  {sign.name}: {best.total_ms:7.3f} ms (walk {best.distance_ms:6.3f})"
    print(line, flush=True)
