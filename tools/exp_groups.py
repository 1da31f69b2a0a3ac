#!/usr/bin/env python3
"""One wave per packet (M2S_GROUP=0) vs four (1) on shallow launches: python tools/exp_groups.py [mesh] [sizes...]  (best of 7 calls)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mesh_to_sdf_amd import Grid, M2STimings, SignMethod, Topology, _lib, generate_grid_sdf, meshes  # noqa: E402

mesh = sys.argv[1] if len(sys.argv) > 1 else "blob-100k"
sizes = [int(c) for c in sys.argv[2:]] or [32, 48, 64, 80, 96, 128, 160, 192]
v, idx = meshes.named(mesh)
lo, hi = meshes.extended_bbox(v, 0.1)
dv = torch.as_tensor(v, device="cuda")
topo = Topology.TriangleList(torch.as_tensor(idx.astype(np.int64), device="cuda").to(torch.int32))
for n in sizes:
    grid = Grid.from_bounding_box(lo, hi, [n, n, n])
    out = torch.empty(n ** 3, dtype=torch.float32, device="cuda")
    for sign in (SignMethod.Raycast, SignMethod.Normal):
        line = f"{mesh} {n:>4}^3 {sign.name:8s}"
        ref = None
        for knob in (0, 1, -1):
            with _lib.knobs(M2S_GROUP=knob):
                best = None
                for _ in range(7):
                    t = M2STimings()
                    generate_grid_sdf(dv, topo, grid, sign, out=out, timings=t)
                    if best is None or t.total_ms < best.total_ms:
                        best = t
            if ref is None:
                ref = out.clone()
            same = bool(torch.equal(out.view(torch.int32), ref.view(torch.int32)))
            line += f"  GROUP={knob:2d}: {best.total_ms:6.3f} ms (walk {best.distance_ms:6.3f}){'' if same else ' DIFFERENT'}"
        print(line, flush=True)
