import sys
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import numpy as np, torch
from mesh_to_sdf_amd import *
from mesh_to_sdf_amd import meshes
v, idx = meshes.named('blob-100k'); lo, hi = meshes.extended_bbox(v, 0.1)
n = 512
g = Grid.from_bounding_box(lo, hi, [n] * 3)
dv = torch.as_tensor(v, device='cuda'); di = torch.as_tensor(idx.astype(np.int64), device='cuda').to(torch.int32)
out = torch.empty(n ** 3, device='cuda')
res = []
for s in range(8):
    best = 1e9
    for rep in range(2):
        t = M2STimings(); generate_grid_sdf(dv, Topology.TriangleList(di), g, SignMethod.Raycast, x_slab=(64 * s, 64 * s + 64), out=out, timings=t)
        best = min(best, t.distance_ms)
    res.append(best)
print("per 64-layer x-slab k_packet ms:", " ".join(f"{r:.2f}" for r in res), " sum", f"{sum(res):.2f}")
