import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from mesh_to_sdf_amd import Grid, M2STimings, SignMethod, Topology, generate_grid_sdf, meshes
d = np.load(os.path.join(ROOT, "tests/golden/suzanne.npz"))
v, idx = d["vertices"].astype(np.float32), d["indices"].astype(np.uint32)
lo, hi = meshes.extended_bbox(v, 0.1)
dv, di = torch.as_tensor(v, device="cuda"), torch.as_tensor(idx.astype(np.int64), device="cuda").to(torch.int32)
for n in ([int(a) for a in sys.argv[1:]] or [256, 512]):
    g = Grid.from_bounding_box(lo, hi, [n] * 3)
    out = torch.empty(n ** 3, device="cuda")
    for sign in (SignMethod.Raycast, SignMethod.Normal):
        best = None
        for _ in range(3):
            t = M2STimings(); generate_grid_sdf(dv, Topology.TriangleList(di), g, sign, out=out, timings=t)
            if best is None or t.total_ms < best.total_ms: best = t
        print(f"suzanne(968 tris) {n}^3 {sign.name}: total {best.total_ms:.3f} ms (build {best.accel_build_ms:.3f}, sign {best.sign_ms:.3f}, seeds {best.seed_ms:.3f}, distance {best.distance_ms:.3f}) -> {n**3/best.total_ms/1e3:.0f} Mvox/s")
