#!/usr/bin/env python3
"""Small calls: wall time per synchronous device-resident call beside the device time between the call's first and last event, and the
kernel launches per call — is the host's launch rate the bound?   python tools/exp_small_wall.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mesh_to_sdf_amd import Grid, M2STimings, SignMethod, Topology, generate_grid_sdf, meshes

d = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests/golden/suzanne.npz"))
cases = [("suzanne", d["vertices"].astype(np.float32), d["indices"].astype(np.uint32))]
v, idx = meshes.blob(80, 71); cases.append(("blob-11k", v, idx))
v, idx = meshes.named("blob-100k"); cases.append(("blob-100k", v, idx))
for name, v, idx in cases:
    lo, hi = meshes.extended_bbox(v, 0.1)
    dv, di = torch.as_tensor(v, device="cuda"), torch.as_tensor(idx.astype(np.int64), device="cuda").to(torch.int32)
    for n in (32, 64, 128):
        g = Grid.from_bounding_box(lo, hi, [n] * 3)
        out = torch.empty(n ** 3, device="cuda")
        for sign in (SignMethod.Raycast, SignMethod.Normal):
            for _ in range(5):
                generate_grid_sdf(dv, Topology.TriangleList(di), g, sign, out=out)
            torch.cuda.synchronize()
            ts, dev = [], []
            for _ in range(30):
                t = M2STimings()
                t0 = time.perf_counter()
                generate_grid_sdf(dv, Topology.TriangleList(di), g, sign, out=out, timings=t)
                ts.append((time.perf_counter() - t0) * 1e3); dev.append(t.total_ms)
            print(f"{name} {n}^3 {sign.name}: wall median {np.median(ts):.3f} ms (min {min(ts):.3f}), device total median {np.median(dev):.3f} ms", flush=True)
