"""A/B helper: distance-phase time of the headline call (or another mesh / size) with whatever library M2S_LIB names."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mesh_to_sdf_amd import Grid, Topology, SignMethod, M2STimings, generate_grid_sdf, meshes
mesh = sys.argv[1] if len(sys.argv) > 1 else "blob-100k"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 512
sign = SignMethod[sys.argv[3]] if len(sys.argv) > 3 else SignMethod.Raycast
v, idx = meshes.named(mesh); lo, hi = meshes.extended_bbox(v, 0.1)
g = Grid.from_bounding_box(lo, hi, [n] * 3)
dv = torch.as_tensor(v, device="cuda"); di = torch.as_tensor(idx.astype(np.int64), device="cuda").to(torch.int32)
out = torch.empty(n ** 3, device="cuda"); best = None
for r in range(int(os.environ.get("REPS", "4"))):
    t = M2STimings()
    generate_grid_sdf(dv, Topology.TriangleList(di), g, sign, timings=t, out=out)
    if best is None or t.distance_ms < best.distance_ms: best = t
print(f"{os.environ.get('M2S_LIB', 'default lib')}: {mesh} {n}^3 {sign.name}: build {best.accel_build_ms:.2f} seed {best.seed_ms:.2f} distance {best.distance_ms:.2f} total {best.total_ms:.2f} ms; checksum {float(out.double().sum()):.6f}", flush=True)
