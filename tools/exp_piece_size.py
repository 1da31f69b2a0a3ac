"""Sum of the per-piece phase times when the 512^3 grid is computed in x-pieces of L layers (one GPU, one stream)."""
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
m = Mesh(dv, Topology.TriangleList(di))
for L in [int(a) for a in sys.argv[1:]] or (512, 256, 128, 64, 32, 16, 8):
    for rep in range(2):
        seed = dist = 0.0; per = []
        for a in range(0, n, L):
            t = M2STimings()
            m.generate_grid_sdf(g, SignMethod.Raycast, x_slab=(a, a + L), out=out, timings=t)
            seed += t.seed_ms; dist += t.distance_ms; per.append(t.distance_ms)
    print(f"L={L:4d}: pieces {n // L:3d}  seed sum {seed:6.2f}  distance sum {dist:6.2f} ms  (min piece {min(per):.3f}, max {max(per):.3f})", flush=True)
