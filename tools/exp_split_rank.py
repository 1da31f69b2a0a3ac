#!/usr/bin/env python3
"""Distance phase of one emulated rank slab (world 8, interleaved) under different split-walk settings."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mesh_to_sdf_amd import Grid, M2STimings, SignMethod, Topology, _lib, generate_grid_sdf, interleaved_slab, meshes

mesh = sys.argv[1] if len(sys.argv) > 1 else "blob-1M"
ranks = [int(r) for r in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 512
v, idx = meshes.named(mesh)
lo, hi = meshes.extended_bbox(v, 0.1)
grid = Grid.from_bounding_box(lo, hi, [n, n, n])
dv = torch.as_tensor(v, device="cuda")
topo = Topology.TriangleList(torch.as_tensor(idx.astype(np.int64), device="cuda").to(torch.int32))
out = torch.empty(n ** 3, dtype=torch.float32, device="cuda")
for rank in ranks:
  a, b, period = interleaved_slab(grid, 8, rank)
  for name, kn in [("off", {"M2S_SPLIT": 0}), ("on", {"M2S_SPLIT": 1}), ("automatic", {})]:
    with _lib.knobs(**kn):
      best = None
      for _ in range(7):
        t = M2STimings()
        generate_grid_sdf(dv, topo, grid, SignMethod.Raycast, x_slab=(a, b), x_period=period, out=out, timings=t)
        if best is None or t.distance_ms < best.distance_ms:
          best = t
    print(f"{mesh} {n}^3 rank {rank} slab ({a},{b})+{period}: {name}: distance {best.distance_ms:.3f} total {best.total_ms:.3f}", flush=True)
