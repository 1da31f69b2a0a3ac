#!/usr/bin/env python3
"""generate_sdf timing over query counts (uniform queries in the extended bbox): python tools/exp_queries.py [mesh] [counts...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mesh_to_sdf_amd import AccelerationMethod, M2STimings, Topology, generate_sdf, meshes  # noqa: E402

mesh = sys.argv[1] if len(sys.argv) > 1 else "blob-100k"
counts = [int(c) for c in sys.argv[2:]] or [100_000, 300_000, 1_000_000, 3_000_000, 10_000_000]
v, idx = meshes.named(mesh)
lo, hi = meshes.extended_bbox(v, 0.1)
dv = torch.as_tensor(v, device="cuda")
topo = Topology.TriangleList(torch.as_tensor(idx.astype(np.int64), device="cuda").to(torch.int32))
for n in counts:
    dq = torch.as_tensor(meshes.uniform_queries(lo, hi, n), device="cuda")
    line = f"{mesh} {n:>9} queries:"
    for am in (AccelerationMethod.RtreeBvh, AccelerationMethod.Rtree):
        best = 1e9
        for _ in range(4):
            t = M2STimings()
            generate_sdf(dv, topo, dq, am, timings=t)
            best = min(best, t.total_ms)
        line += f"  {am.kind}: {best:7.3f} ms ({n / best / 1e3:7.1f} Mq/s)"
    print(line, flush=True)
