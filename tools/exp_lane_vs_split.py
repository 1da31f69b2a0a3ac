#!/usr/bin/env python3
"""Lane walk against packet walk with the split walk, coarse grids over fine meshes: python tools/exp_lane_vs_split.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mesh_to_sdf_amd import Grid, M2STimings, SignMethod, Topology, _lib, generate_grid_sdf, meshes

CASES = (("blob-100k", (24, 32, 48, 64)), ("blob-1M", (48, 64, 80)), ("blob-11k", (12, 16, 24, 32))) if os.environ.get("DENSE") else None
for mesh, sizes in CASES or (("blob-100k", (64, 80, 88, 96, 112)), ("blob-1M", (96, 112, 128, 160, 192, 224)), ("blob-11k", (32, 40, 48, 56, 64, 96))):
    v, idx = meshes.blob(80, 71) if mesh == "blob-11k" else meshes.named(mesh)
    lo, hi = meshes.extended_bbox(v, 0.1)
    dv = torch.as_tensor(v, device="cuda")
    topo = Topology.TriangleList(torch.as_tensor(idx.astype(np.int64), device="cuda").to(torch.int32))
    for n in sizes:
        grid = Grid.from_bounding_box(lo, hi, [n, n, n])
        out = torch.empty(n ** 3, dtype=torch.float32, device="cuda")
        line = f"{mesh} {n:>4}^3 ({idx.size // 3 / (n / 4) ** 3:6.1f} triangles per brick) Raycast:"
        ref = None
        for name, kn in (("lane walk", {"M2S_LANE_WALK": 1, "M2S_BRUTE_MAX": 0}), ("packet", {"M2S_LANE_WALK": 0, "M2S_SPLIT": 0, "M2S_BRUTE_MAX": 0}),
                         ("packet + split", {"M2S_LANE_WALK": 0, "M2S_SPLIT": 1, "M2S_BRUTE_MAX": 0}), ("automatic", {})):
            with _lib.knobs(**kn):
                best = None
                for _ in range(7):
                    t = M2STimings()
                    generate_grid_sdf(dv, topo, grid, SignMethod.Raycast, out=out, timings=t)
                    if best is None or t.total_ms < best.total_ms:
                        best = t
            if ref is None:
                ref = out.clone()
            same = bool(torch.equal(out.view(torch.int32), ref.view(torch.int32)))
            line += f"  {name}: {best.total_ms:6.3f} (walk {best.distance_ms:6.3f}){'' if same else ' DIFFERENT'} |"
        print(line, flush=True)
