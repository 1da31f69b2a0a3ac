#!/usr/bin/env python3
"""Collapsed-leaf size (M2S_LEAF_MAX) over regimes, whole one-shot calls: python tools/exp_leaf.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mesh_to_sdf_amd import Grid, M2STimings, SignMethod, Topology, _lib, generate_grid_sdf, meshes

for mesh, sizes in (("blob-100k", (96, 128, 192, 256, 512)), ("blob-1M", (128, 256, 512)), ("sheet-100k", (256,))):
    v, idx = meshes.named(mesh)
    lo, hi = meshes.extended_bbox(v, 0.1)
    dv = torch.as_tensor(v, device="cuda")
    topo = Topology.TriangleList(torch.as_tensor(idx.astype(np.int64), device="cuda").to(torch.int32))
    for n in sizes:
        grid = Grid.from_bounding_box(lo, hi, [n, n, n])
        out = torch.empty(n ** 3, dtype=torch.float32, device="cuda")
        line = f"{mesh} {n:>4}^3 Raycast:"
        ref = None
        for leaf in (1, 2, 3, 4, 6, 8):
            with _lib.knobs(M2S_LEAF_MAX=leaf):
                best = None
                for _ in range(5):
                    t = M2STimings()
                    generate_grid_sdf(dv, topo, grid, SignMethod.Raycast, out=out, timings=t)
                    if best is None or t.total_ms < best.total_ms:
                        best = t
            if ref is None:
                ref = out.clone()
            same = bool(torch.equal(out.view(torch.int32), ref.view(torch.int32)))
            line += f"  leaf {leaf}: {best.total_ms:7.3f} (walk {best.distance_ms:6.3f}){'' if same else ' DIFFERENT'}"
        print(line, flush=True)
