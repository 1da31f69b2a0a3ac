#!/usr/bin/env python3
"""Step time of small / medium grids with and without cut lists (M2S_CUT_MIN_PACKETS switched through m2s_tuning_set)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mesh_to_sdf_amd import _lib, Grid, M2STimings, SignMethod, Topology, generate_grid_sdf, meshes  # noqa: E402

for mesh in ("blob-100k", "blob-6k"):
    v, idx = meshes.named(mesh)
    lo, hi = meshes.extended_bbox(v, 0.1)
    dv = torch.as_tensor(v, device="cuda")
    topo = Topology.TriangleList(torch.as_tensor(idx.astype(np.int64), device="cuda").to(torch.int32))
    for n in (64, 96, 128, 160, 192, 256):
        grid = Grid.from_bounding_box(lo, hi, [n, n, n])
        out = torch.empty(n ** 3, dtype=torch.float32, device="cuda")
        res = []
        for cutmin in ("8", "100000000"):
            _lib.set_knob("M2S_CUT_MIN_PACKETS", cutmin)
            ts = []
            for i in range(12):
                t0 = time.perf_counter()
                generate_grid_sdf(dv, topo, grid, SignMethod.Raycast, out=out)
                ts.append((time.perf_counter() - t0) * 1e3)
            res.append(float(np.median(ts[2:])))
        print(f"{mesh} {n}^3 ({(n // 4) ** 3} packets): with cut lists {res[0]:.3f} ms, without {res[1]:.3f} ms")
