#!/usr/bin/env python3
"""One-shot calls against calls on a persistent mesh (tree resident, leaves re-marked per grid): python tools/exp_persistent.py [mesh] [sizes...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mesh_to_sdf_amd import Grid, M2STimings, Mesh, SignMethod, Topology, _lib, generate_grid_sdf, meshes

mesh = sys.argv[1] if len(sys.argv) > 1 else "blob-100k"
sizes = [int(c) for c in sys.argv[2:]] or [32, 64, 96, 128, 192, 256]
v, idx = meshes.blob(80, 71) if mesh == "blob-11k" else meshes.named(mesh)
lo, hi = meshes.extended_bbox(v, 0.1)
dv = torch.as_tensor(v, device="cuda")
topo = Topology.TriangleList(torch.as_tensor(idx.astype(np.int64), device="cuda").to(torch.int32))
with Mesh(dv, topo) as m:
    for n in sizes:
        grid = Grid.from_bounding_box(lo, hi, [n, n, n])
        out = torch.empty(n ** 3, dtype=torch.float32, device="cuda")
        res = {}
        for name, call in (("one-shot", lambda: generate_grid_sdf(dv, topo, grid, SignMethod.Raycast, out=out)),
                           ("persistent mesh", lambda: m.generate_grid_sdf(grid, SignMethod.Raycast, out=out)),
                           ("persistent mesh, leaves of 2", None)):
            kn = {"M2S_LEAF_MAX": 2} if call is None else {}
            fn = call or (lambda: m.generate_grid_sdf(grid, SignMethod.Raycast, out=out))
            with _lib.knobs(**kn):
                ts = []
                for _ in range(9):
                    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
            res[name] = min(ts[2:])
        print(f"{mesh} {n:>4}^3 Raycast, wall per call:  " + "  |  ".join(f"{k}: {x:6.3f} ms" for k, x in res.items()), flush=True)
