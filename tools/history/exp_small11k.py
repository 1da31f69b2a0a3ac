import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from mesh_to_sdf_amd import Grid, M2STimings, SignMethod, Topology, generate_grid_sdf, meshes
v, idx = meshes.named("blob-11k")
lo, hi = v.min(0), v.max(0)
dv = torch.as_tensor(v, device="cuda")
topo = Topology.TriangleList(torch.as_tensor(idx.astype(np.int64), device="cuda").to(torch.int32))
for n in (16, 32, 64, 100):
    g = Grid.from_bounding_box(lo, hi, [n]*3)
    out = torch.empty(n**3, dtype=torch.float32, device="cuda")
    best = None
    for _ in range(6):
        t = M2STimings()
        generate_grid_sdf(dv, topo, g, SignMethod.Raycast, out=out, timings=t)
        if best is None or t.total_ms < best.total_ms: best = t
    print(f"{n}^3: build {best.accel_build_ms:.3f} sign {best.sign_ms:.3f} seed {best.seed_ms:.3f} distance {best.distance_ms:.3f} total {best.total_ms:.3f}")
