import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from mesh_to_sdf_amd import Grid, M2STimings, SignMethod, Topology, _lib, generate_grid_sdf, meshes
for mesh, sizes in (("blob-100k", (128, 160, 192, 224, 256, 320)), ("blob-1M", (160, 192, 256)), ("blob-11k", (128, 192, 256))):
    v, idx = meshes.blob(80, 71) if mesh == "blob-11k" else meshes.named(mesh)
    lo, hi = meshes.extended_bbox(v, 0.1)
    dv = torch.as_tensor(v, device="cuda")
    topo = Topology.TriangleList(torch.as_tensor(idx.astype(np.int64), device="cuda").to(torch.int32))
    for n in sizes:
        grid = Grid.from_bounding_box(lo, hi, [n, n, n])
        out = torch.empty(n ** 3, dtype=torch.float32, device="cuda")
        line = f"{mesh} {n}^3 ({(n // 4) ** 3} packets):"
        for name, kn in (("no lists", {"M2S_CUT_MIN_PACKETS": 1000000000}), ("lists", {"M2S_CUT_MIN_PACKETS": 1}), ("automatic", {})):
            with _lib.knobs(**kn):
                best = None
                for _ in range(7):
                    t = M2STimings()
                    generate_grid_sdf(dv, topo, grid, SignMethod.Raycast, out=out, timings=t)
                    if best is None or t.total_ms < best.total_ms:
                        best = t
            line += f"  {name}: {best.total_ms:6.3f} (seed+cut {best.seed_ms:5.3f} walk {best.distance_ms:6.3f}) |"
        print(line, flush=True)
