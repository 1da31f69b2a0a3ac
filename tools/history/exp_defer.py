#!/usr/bin/env python3
"""Dense (queued) exact evaluations in the packet walk, off / on: python tools/exp_defer.py   (best of 7 calls, device-resident)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mesh_to_sdf_amd import Grid, M2STimings, SignMethod, Topology, _lib, generate_grid_sdf, meshes

cases = (("blob-100k", (96, 128, 160, 192, 256, 384, 512)), ("blob-1M", (128, 192, 256, 512)), ("blob-11k", (64, 96, 128, 256)))
if len(sys.argv) > 1:
    cases = ((sys.argv[1], tuple(int(c) for c in sys.argv[2:])),)
sign = SignMethod.Normal if os.environ.get("SIGN") == "Normal" else SignMethod.Raycast
for mesh, sizes in cases:
    v, idx = meshes.blob(80, 71) if mesh == "blob-11k" else meshes.named(mesh)
    lo, hi = meshes.extended_bbox(v, 0.1)
    dv = torch.as_tensor(v, device="cuda")
    topo = Topology.TriangleList(torch.as_tensor(idx.astype(np.int64), device="cuda").to(torch.int32))
    for n in sizes:
        grid = Grid.from_bounding_box(lo, hi, [n, n, n])
        out = torch.empty(n ** 3, dtype=torch.float32, device="cuda")
        line = f"{mesh} {n:>4}^3 ({idx.size // 3 / (n / 4) ** 3:6.2f} triangles per brick) {sign.name}:"
        ref = None
        for name, kn in (("packets", {"M2S_LANE_WALK": 0, "M2S_DEFER": 0}), ("packets, dense evaluations", {"M2S_LANE_WALK": 0, "M2S_DEFER": 1}), ("packets, dense pre-tests and evaluations", {"M2S_LANE_WALK": 0, "M2S_DEFER": 3}), ("automatic", {})):
            with _lib.knobs(**kn):
                best = None
                for _ in range(7):
                    t = M2STimings()
                    generate_grid_sdf(dv, topo, grid, sign, out=out, timings=t)
                    if best is None or t.distance_ms < best.distance_ms:
                        best = t
            if ref is None:
                ref = out.clone()
            same = bool(torch.equal(out.view(torch.int32), ref.view(torch.int32)))
            line += f"  {name}: walk {best.distance_ms:6.3f} total {best.total_ms:6.3f}{'' if same else ' DIFFERENT'} |"
        print(line, flush=True)
