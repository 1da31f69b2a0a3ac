#!/usr/bin/env python3
"""Split walk on/off over grid sizes: python tools/exp_split.py [mesh] [sizes...]   (best of 7 calls, device-resident; M2S_SPLIT switched per call)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mesh_to_sdf_amd import Grid, M2STimings, SignMethod, Topology, _lib, generate_grid_sdf, meshes  # noqa: E402

mesh = sys.argv[1] if len(sys.argv) > 1 else "blob-100k"
sizes = [int(c) for c in sys.argv[2:]] or [64, 96, 128, 160, 192, 256, 384, 512]
configs = [("off", {"M2S_SPLIT": 0}), ("on", {"M2S_SPLIT": 1}), ("on, nothing suspended", {"M2S_SPLIT": 1, "M2S_SPLIT_PATIENCE": 1e9}), ("automatic", {})]
if os.environ.get("SPLIT_SWEEP"):
    configs += [(f"rounds {r} patience {pt} sub {lo}-{hi}", {"M2S_SPLIT": 1, "M2S_SPLIT_ROUNDS": r, "M2S_SPLIT_PATIENCE": pt, "M2S_SPLIT_MIN_RECORDS": lo, "M2S_SPLIT_MAX_RECORDS": hi})
                for r, pt, lo, hi in ((3, 1.5, 16, 128), (2, 1.0, 16, 128), (2, 2.0, 16, 128), (2, 3.0, 16, 128), (2, 1.5, 32, 256), (2, 1.5, 64, 512))]
v, idx = meshes.named(mesh)
lo, hi = meshes.extended_bbox(v, 0.1)
dv = torch.as_tensor(v, device="cuda")
topo = Topology.TriangleList(torch.as_tensor(idx.astype(np.int64), device="cuda").to(torch.int32))
for n in sizes:
    grid = Grid.from_bounding_box(lo, hi, [n, n, n])
    out = torch.empty(n ** 3, dtype=torch.float32, device="cuda")
    ref = None
    for sign in (SignMethod.Raycast, SignMethod.Normal):
        line = f"{mesh} {n:>4}^3 {sign.name:8}:"
        for name, kn in configs:
            with _lib.knobs(**kn):
                best = None
                for _ in range(7):
                    t = M2STimings()
                    generate_grid_sdf(dv, topo, grid, sign, out=out, timings=t)
                    if best is None or t.distance_ms < best.distance_ms:
                        best = t
            if name == "off":
                ref = out.clone()
            same = bool(torch.equal(out.view(torch.int32), ref.view(torch.int32)))
            line += f"  {name}: walk {best.distance_ms:6.3f} total {best.total_ms:6.3f}{'' if same else ' DIFFERENT'} |"
        print(line, flush=True)
