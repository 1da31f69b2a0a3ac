import numpy as np, torch, sys
sys.path.insert(0,'/root/repo')
from mesh_to_sdf_amd import *
from mesh_to_sdf_amd import meshes
import os
for mesh, n in (("blob-6k", (128, 40, 36)), ("blob-100k", (256, 128, 128))):
    v, idx = meshes.named(mesh); lo, hi = meshes.extended_bbox(v, 0.1)
    g = Grid.from_bounding_box(lo, hi, list(n))
    dv = torch.as_tensor(v, device="cuda"); di = torch.as_tensor(idx.astype(np.int64), device="cuda")
    for sign in (SignMethod.Raycast, SignMethod.Normal):
        want = generate_grid_sdf(dv, Topology.TriangleList(di), g, sign)
        for world in (2, 4):
            out = torch.full_like(want, float("nan"))
            for k in range(world):
                sl = interleaved_slab(g, world, k)
                generate_grid_sdf(dv, Topology.TriangleList(di), g, sign, x_slab=sl[:2], x_period=sl[2], out=out)
            ok = torch.equal(out.view(torch.int32), want.view(torch.int32))
            print(mesh, n, sign.name, world, sl, "OK" if ok else f"MISMATCH {(out.view(torch.int32) != want.view(torch.int32)).sum().item()} nan {torch.isnan(out).sum().item()}")
