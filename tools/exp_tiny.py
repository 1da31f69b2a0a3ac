#!/usr/bin/env python3
"""Tiny problems: the tree-less k_brute_split (cells x triangles <= M2S_BRUTE_MAX) against the build + walk, whole call, device resident.
    python tools/exp_tiny.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mesh_to_sdf_amd import _lib, Grid, M2STimings, SignMethod, Topology, generate_grid_sdf, meshes  # noqa: E402

cases = [("suzanne", 16), ("suzanne", 24), ("suzanne", 32), ("suzanne", 40), ("suzanne", 48), ("blob-1k", 32), ("blob-1k", 48), ("blob-3k", 24), ("blob-3k", 32), ("blob-6k", 16), ("blob-6k", 24), ("blob-6k", 32), ("blob-11k", 12), ("blob-11k", 16), ("blob-11k", 20), ("blob-11k", 24), ("blob-100k", 6), ("blob-100k", 8), ("blob-100k", 12), ("blob-100k", 16)]
for name, n in cases:
    if name == "suzanne":
        d = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests/golden/suzanne.npz"))
        v, idx = d["vertices"].astype(np.float32), d["indices"].astype(np.uint32)
    elif name == "blob-1k":
        v, idx = meshes.blob(24, 21)
    elif name == "blob-3k":
        v, idx = meshes.blob(42, 36)
    else:
        v, idx = meshes.named(name)
    lo, hi = v.min(0), v.max(0)
    g = Grid.from_bounding_box(lo, hi, [n] * 3)
    dv = torch.as_tensor(v, device="cuda")
    topo = Topology.TriangleList(torch.as_tensor(idx.astype(np.int64), device="cuda").to(torch.int32))
    line = f"{name} {n}^3 (cells x triangles = {n ** 3 * (idx.size // 3):.2e}):"
    outs = []
    for label, limit in (("brute", "1e30"), ("tree", "0"), ("auto", None)):
        _lib.set_knob("M2S_BRUTE_MAX", limit)
        for sign in (SignMethod.Raycast, SignMethod.Normal):
            best = None
            for _ in range(7):
                t = M2STimings()
                out = generate_grid_sdf(dv, topo, g, sign, timings=t)
                if best is None or t.total_ms < best:
                    best = t.total_ms
            outs.append(out.clone())
            line += f"  {label} {sign.name} {best:.3f}"
    _lib.set_knob("M2S_BRUTE_MAX", None)
    same = bool(torch.equal(outs[0].view(torch.int32), outs[2].view(torch.int32)) and torch.equal(outs[1].view(torch.int32), outs[3].view(torch.int32)))
    print(line + f"  identical: {same}", flush=True)
