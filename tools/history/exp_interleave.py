#!/usr/bin/env python3
"""Would interleaved half-slabs balance the ranks?  Rank r of N computes the half-slabs r and N + r (of 2N) as two asynchronous
calls through one persistent mesh on two streams (distributed.run_pieces), emulated alone on one GPU."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch  # noqa: E402

from mesh_to_sdf_amd import Grid, Mesh, SignMethod, Topology, meshes, slab_bounds  # noqa: E402
from mesh_to_sdf_amd.distributed import run_pieces  # noqa: E402

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = 512
v, idx = meshes.named("blob-100k")
lo, hi = meshes.extended_bbox(v, 0.1)
grid = Grid.from_bounding_box(lo, hi, [n, n, n])
dv = torch.as_tensor(v, device="cuda")
topo = Topology.TriangleList(torch.as_tensor(idx.astype(np.int64), device="cuda").to(torch.int32))
out = torch.empty(n ** 3, dtype=torch.float32, device="cuda")
worst = 0
for r in range(world):
    pieces = [slab_bounds(n, 2 * world, r), slab_bounds(n, 2 * world, world + r)]

    def step():
        m = Mesh(dv, topo)
        run_pieces(m, grid, SignMethod.Raycast, out, pieces)
        torch.cuda.current_stream().synchronize()
        m.close()

    for _ in range(3):
        step()
    ts = []
    for _ in range(15):
        t0 = time.perf_counter()
        step()
        ts.append((time.perf_counter() - t0) * 1e3)
    worst = max(worst, float(np.median(ts)))
    print(f"world {world} rank {r} half-slabs {pieces}: wall median {np.median(ts):.3f} ms (min {np.min(ts):.3f})")
print(f"## slowest rank {worst:.3f} ms")
