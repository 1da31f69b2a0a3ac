#!/usr/bin/env python3
"""M2S_MIX_CELLS sweep: packets within c cells of their seed triangle through the lane walk (k_lane), the rest through the packet
walk — what a per-packet choice of the walk could win.  python tools/exp_mix.py [mesh] [n] [cells...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mesh_to_sdf_amd import Grid, M2STimings, SignMethod, Topology, generate_grid_sdf, meshes  # noqa: E402

mesh = sys.argv[1] if len(sys.argv) > 1 else "blob-1M"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 512
cells = [float(c) for c in sys.argv[3:]] or [0, 0.5, 1, 2, 4, 8]
v, idx = meshes.named(mesh)
lo, hi = meshes.extended_bbox(v, 0.1)
g = Grid.from_bounding_box(lo, hi, [n] * 3)
dv = torch.as_tensor(v, device="cuda")
topo = Topology.TriangleList(torch.as_tensor(idx.astype(np.int64), device="cuda").to(torch.int32))
ref = None
for c in cells:
    if c > 0:
        os.environ["M2S_MIX_CELLS"] = str(c)
    else:
        os.environ.pop("M2S_MIX_CELLS", None)
    best = None
    for _ in range(3):
        t = M2STimings()
        out = generate_grid_sdf(dv, topo, g, SignMethod.Raycast, timings=t)
        if best is None or t.total_ms < best.total_ms:
            best = t
    if ref is None:
        ref = out.clone()
    same = bool(torch.equal(out.view(torch.int32), ref.view(torch.int32)))
    print(f"{mesh} {n}^3 mix below {c} cells: distance {best.distance_ms:.3f} ms, total {best.total_ms:.3f} ms, identical to the packet walk: {same}", flush=True)
