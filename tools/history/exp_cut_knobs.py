#!/usr/bin/env python3
"""Cut-list knobs on a grid call: python tools/exp_cut_knobs.py [mesh] [n]   (best of 7, device-resident, Raycast)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mesh_to_sdf_amd import Grid, M2STimings, SignMethod, Topology, _lib, generate_grid_sdf, meshes

mesh = sys.argv[1] if len(sys.argv) > 1 else "blob-100k"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 512
v, idx = meshes.named(mesh)
lo, hi = meshes.extended_bbox(v, 0.1)
dv = torch.as_tensor(v, device="cuda")
topo = Topology.TriangleList(torch.as_tensor(idx.astype(np.int64), device="cuda").to(torch.int32))
grid = Grid.from_bounding_box(lo, hi, [n, n, n])
out = torch.empty(n ** 3, dtype=torch.float32, device="cuda")
ref = None
configs = [("defaults", {})]
configs += [(f"near {a}", {"M2S_CUT_NEAR": a}) for a in (1.0, 1.5, 3.0, 4.0)]
configs += [(f"far 1/{b}", {"M2S_CUT_FAR": 1.0 / b}) for b in (64, 48, 24, 16)]
configs += [(f"wave cap {c}", {"M2S_CUT_WAVE_CAP": c}) for c in (200, 260, 450, 600, 900)]
for name, kn in configs:
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
    print(f"{mesh} {n}^3 {name:14}: total {best.total_ms:7.3f} seed+cut {best.seed_ms:6.3f} walk {best.distance_ms:7.3f}{'' if same else ' DIFFERENT'}", flush=True)
