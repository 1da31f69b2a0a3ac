#!/usr/bin/env python3
"""Cut lists in one level (M2S_CUT_COARSE=0) vs two (1): whole calls, best of 5; the results must be identical.
python tools/exp_cut_coarse.py [case ...]   cases: head c2 rank c4 c4slab c5slab c5 g192 b1024"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mesh_to_sdf_amd import Grid, M2STimings, SignMethod, Topology, _lib, generate_grid_sdf, meshes  # noqa: E402

CASES = {"head": ("blob-100k", 512, "Raycast", None), "c2": ("blob-100k", 256, "Raycast", None), "g192": ("blob-100k", 192, "Raycast", None),
         "c4": ("blob-1M", 512, "Raycast", None), "c4slab": ("blob-1M", 512, "Raycast", (192, 256)), "rank": ("blob-100k", 512, "Raycast", (192, 256)),
         "rank0": ("blob-100k", 512, "Raycast", (0, 64)),
         "c5slab": ("sheet-100k", 1024, "Normal", (448, 576)), "c5": ("sheet-100k", 1024, "Normal", None),
         "b1024": ("blob-100k", 1024, "Raycast", None), "headN": ("blob-100k", 512, "Normal", None)}
for name in (sys.argv[1:] or ["head", "c2", "rank", "rank0", "c4", "c4slab", "c5slab", "c5"]):
    mesh, n, sign, slab = CASES[name]
    v, idx = meshes.named(mesh)
    lo, hi = meshes.extended_bbox(v, 0.1)
    g = Grid.from_bounding_box(lo, hi, [n] * 3)
    dv = torch.as_tensor(v, device="cuda")
    topo = Topology.TriangleList(torch.as_tensor(idx.astype(np.int64), device="cuda").to(torch.int32))
    out = torch.empty(n ** 3, dtype=torch.float32, device="cuda")
    ref = None
    for knob in (0, 1, 0, 1):
        with _lib.knobs(M2S_CUT_COARSE=knob):
            best = None
            for _ in range(5):
                t = M2STimings()
                generate_grid_sdf(dv, topo, g, SignMethod[sign], out=out, timings=t, x_slab=slab)
                if best is None or t.total_ms < best.total_ms:
                    best = t
        same = ""
        if ref is None:
            ref = out.clone()
        else:
            same = f" identical={bool(torch.equal(out.view(torch.int32), ref.view(torch.int32)))}"
        print(f"{name:7s} {mesh} {n}^3 {sign} slab={slab} CUT_COARSE={knob}: build {best.accel_build_ms:.3f} sign {best.sign_ms:.3f} seed+cut {best.seed_ms:.3f} "
              f"walk {best.distance_ms:.3f} total {best.total_ms:.3f} ms{same}", flush=True)
    del out, ref
