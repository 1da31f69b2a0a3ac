"""Quick experiment driver: one 512^3 (or --grid N) generate_grid_sdf with phase timings."""
import argparse, sys
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import numpy as np, torch
from mesh_to_sdf_amd import *
from mesh_to_sdf_amd import meshes
ap = argparse.ArgumentParser(); ap.add_argument('--grid', type=int, default=512); ap.add_argument('--mesh', default='blob-100k')
ap.add_argument('--reps', type=int, default=3); ap.add_argument('--sign', default='Raycast')
a = ap.parse_args()
v, idx = meshes.named(a.mesh); lo, hi = meshes.extended_bbox(v, 0.1)
g = Grid.from_bounding_box(lo, hi, [a.grid] * 3)
dv = torch.as_tensor(v, device='cuda'); di = torch.as_tensor(idx.astype(np.int64), device='cuda').to(torch.int32)
out = torch.empty(a.grid ** 3, device='cuda')
best = None
for r in range(a.reps):
    t = M2STimings()
    generate_grid_sdf(dv, Topology.TriangleList(di), g, SignMethod[a.sign], timings=t, out=out)
    if best is None or t.distance_ms < best.distance_ms: best = t
print(f"grid {a.grid}^3 {a.mesh} {a.sign}: build {best.accel_build_ms:.3f} sign {best.sign_ms:.3f} distance {best.distance_ms:.3f} ms  checksum {float(out.double().abs().sum()):.6f}")
