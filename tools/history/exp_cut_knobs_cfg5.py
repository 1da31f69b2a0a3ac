import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from mesh_to_sdf_amd import Grid, M2STimings, SignMethod, Topology, _lib, generate_grid_sdf, meshes
mesh, n = "sheet-100k", 1024
v, idx = meshes.named(mesh)
lo, hi = meshes.extended_bbox(v, 0.1)
dv = torch.as_tensor(v, device="cuda")
topo = Topology.TriangleList(torch.as_tensor(idx.astype(np.int64), device="cuda").to(torch.int32))
grid = Grid.from_bounding_box(lo, hi, [n, n, n])
out = torch.empty(n ** 3, dtype=torch.float32, device="cuda")
configs = [("defaults", {})] + [(f"far 1/{b}", {"M2S_CUT_FAR": 1.0 / b}) for b in (64, 128, 16)] + [(f"near {a}", {"M2S_CUT_NEAR": a}) for a in (1.5, 3.0)] + [(f"wave cap {c}", {"M2S_CUT_WAVE_CAP": c}) for c in (250, 500, 700)] + [("far 1/64 cap 500", {"M2S_CUT_FAR": 1 / 64, "M2S_CUT_WAVE_CAP": 500})]
for name, kn in configs:
    with _lib.knobs(**kn):
        best = None
        for _ in range(3):
            t = M2STimings()
            generate_grid_sdf(dv, topo, grid, SignMethod.Normal, out=out, timings=t)
            if best is None or t.total_ms < best.total_ms:
                best = t
    print(f"{mesh} {n}^3 Normal {name:18}: total {best.total_ms:7.3f} seed+cut {best.seed_ms:6.3f} walk {best.distance_ms:7.3f}", flush=True)
