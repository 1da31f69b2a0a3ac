#!/usr/bin/env python3
"""Tiny problems: the tree-less k_brute_split (cells x triangles <= M2S_BRUTE_MAX) against the build + walk, whole call, device resident.
    python tools/exp_tiny.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mesh_to_sdf_amd import _lib, Grid, M2STimings, SignMethod, Topology, generate_grid_sdf, meshes  # noqa: E402

cases = [("blob-11k", 16), ("blob-11k", 20), ("blob-11k", 24), ("blob-11k", 32), ("blob-100k", 8), ("blob-100k", 12), ("blob-100k", 16), ("blob-6k", 16), ("blob-6k", 32), ("blob-6k", 48)]
for name, n in cases:
    v, idx = meshes.named(name)
    lo, hi = v.min(0), v.max(0)
    g = Grid.from_bounding_box(lo, hi, [n] * 3)
    dv = torch.as_tensor(v, device="cuda")
    topo = Topology.TriangleList(torch.as_tensor(idx.astype(np.int64), device="cuda").to(torch.int32))
    line = f"{name} {n}^3 (cells x triangles = {n ** 3 * (idx.size // 3):.2e}):"
    outs = []
    for label, limit in (("brute", "1e30"), ("tree", "0")):
        _lib.set_knob("M2S_BRUTE_MAX", limit)
        best = None
        for sign in (SignMethod.Raycast, SignMethod.Normal):
            for _ in range(5):
                t = M2STimings()
                out = generate_grid_sdf(dv, topo, g, sign, timings=t)
                if sign == SignMethod.Raycast and (best is None or t.total_ms < best):
                    best = t.total_ms
            outs.append(out.clone())
        line += f"  {label} {best:.3f} ms"
    _lib.set_knob("M2S_BRUTE_MAX", None)
    same = bool(torch.equal(outs[0].view(torch.int32), outs[2].view(torch.int32)) and torch.equal(outs[1].view(torch.int32), outs[3].view(torch.int32)))
    print(line + f"  identical: {same}", flush=True)
