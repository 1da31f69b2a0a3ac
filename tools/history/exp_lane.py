import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mesh_to_sdf_amd import Grid, M2STimings, SignMethod, Topology, generate_grid_sdf, meshes
for mesh in ("blob-1M", "blob-100k"):
    v, idx = meshes.named(mesh); lo, hi = meshes.extended_bbox(v, 0.1)
    dv, di = torch.as_tensor(v, device="cuda"), torch.as_tensor(idx.astype(np.int64), device="cuda").to(torch.int32)
    for n in (64, 128, 256, 512):
        g = Grid.from_bounding_box(lo, hi, [n] * 3); out = torch.empty(n ** 3, device="cuda")
        best = None
        for _ in range(2):
            t = M2STimings(); generate_grid_sdf(dv, Topology.TriangleList(di), g, SignMethod.Raycast, out=out, timings=t)
            if best is None or t.total_ms < best.total_ms: best = t
        print(f"{mesh} {n}^3 LANE_WALK={os.environ.get('M2S_LANE_WALK','auto')}: distance {best.distance_ms:.3f} total {best.total_ms:.3f} ms; tris per surface brick ~ {v.shape[0]*2/(6*((n/4)**3)**(2/3)):.1f}", flush=True)
