import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mesh_to_sdf_amd import Grid, M2STimings, SignMethod, Topology, _lib, generate_grid_sdf, meshes
d = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests/golden/suzanne.npz"))
v, idx = d["vertices"].astype(np.float32), d["indices"].astype(np.uint32)
lo, hi = meshes.extended_bbox(v, 0.1)
dv, di = torch.as_tensor(v, device="cuda"), torch.as_tensor(idx.astype(np.int64), device="cuda").to(torch.int32)
for n in (48, 64, 96, 128):
    g = Grid.from_bounding_box(lo, hi, [n] * 3)
    out = torch.empty(n ** 3, device="cuda")
    for sign in (SignMethod.Raycast, SignMethod.Normal):
        line = f"suzanne {n}^3 {sign.name}:"
        for knob in (0, 1):
            with _lib.knobs(M2S_GROUP=knob):
                best = None
                for _ in range(7):
                    t = M2STimings(); generate_grid_sdf(dv, Topology.TriangleList(di), g, sign, out=out, timings=t)
                    if best is None or t.total_ms < best.total_ms: best = t
            line += f"  GROUP={knob}: total {best.total_ms:.3f} (walk {best.distance_ms:.3f})"
        print(line, flush=True)
