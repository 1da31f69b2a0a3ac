#!/usr/bin/env python3
"""The build's treelet pass on / off for one-shot grid calls over coarse grids (leaves of 8 - 16): python tools/exp_treelets.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mesh_to_sdf_amd import Grid, M2STimings, SignMethod, Topology, _lib, generate_grid_sdf, meshes

for mesh, sizes in ((("blob-100k", (160, 192, 224)), ("blob-1M", (320, 384)), ("blob-11k", (64, 80, 96, 112))) if os.environ.get("LEAF4") else (("blob-100k", (32, 64, 96, 128)), ("blob-1M", (64, 128, 192, 256)), ("blob-11k", (16, 32, 48, 64)))):
    v, idx = meshes.blob(80, 71) if mesh == "blob-11k" else meshes.named(mesh)
    lo, hi = meshes.extended_bbox(v, 0.1)
    dv = torch.as_tensor(v, device="cuda")
    topo = Topology.TriangleList(torch.as_tensor(idx.astype(np.int64), device="cuda").to(torch.int32))
    for n in sizes:
        grid = Grid.from_bounding_box(lo, hi, [n, n, n])
        out = torch.empty(n ** 3, dtype=torch.float32, device="cuda")
        line = f"{mesh} {n:>4}^3 ({idx.size // 3 / (n / 4) ** 3:6.1f} triangles per brick) Raycast:"
        ref = None
        for name, kn in (("treelets", {"M2S_TREELETS": 1}), ("no treelets", {"M2S_TREELETS": 0}), ("automatic", {})):
            with _lib.knobs(**kn):
                best = None
                for _ in range(9):
                    t = M2STimings()
                    generate_grid_sdf(dv, topo, grid, SignMethod.Raycast, out=out, timings=t)
                    if best is None or t.total_ms < best.total_ms:
                        best = t
            if ref is None:
                ref = out.clone()
            same = bool(torch.equal(out.view(torch.int32), ref.view(torch.int32)))
            line += f"  {name}: total {best.total_ms:6.3f} (build {best.accel_build_ms:5.3f} walk {best.distance_ms:6.3f}){'' if same else ' DIFFERENT'} |"
        print(line, flush=True)
