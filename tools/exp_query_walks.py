#!/usr/bin/env python3
"""Generic queries: lane walk against packet walk, leaves of 2 / 4 / 8: python tools/exp_query_walks.py [mesh]   (one-shot calls, device-resident, best of 5)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mesh_to_sdf_amd import AccelerationMethod, M2STimings, SignMethod, Topology, _lib, generate_sdf, meshes

mesh = sys.argv[1] if len(sys.argv) > 1 else "blob-100k"
v, idx = meshes.blob(80, 71) if mesh == "blob-11k" else meshes.named(mesh)
lo, hi = meshes.extended_bbox(v, 0.1)
dv = torch.as_tensor(v, device="cuda")
topo = Topology.TriangleList(torch.as_tensor(idx.astype(np.int64), device="cuda").to(torch.int32))
am = AccelerationMethod.RtreeBvh
for nq in (30000, 100000, 300000, 1000000, 3000000, 10000000):
    q = torch.as_tensor(meshes.uniform_queries(lo, hi, nq), device="cuda")
    line = f"{mesh} x {nq:>8} queries RtreeBvh:"
    ref = None
    for name, kn in (("automatic", {}), ("automatic, no treelets", {"M2S_TREELETS": 0}), ("lane", {"M2S_LANE_WALK": 1, "M2S_BRUTE_MAX": 0}), ("packets, leaves of 2", {"M2S_LANE_WALK": 0, "M2S_BRUTE_MAX": 0, "M2S_LEAF_MAX": 2}),
                     ("leaves of 4", {"M2S_LANE_WALK": 0, "M2S_BRUTE_MAX": 0, "M2S_LEAF_MAX": 4}), ("leaves of 8", {"M2S_LANE_WALK": 0, "M2S_BRUTE_MAX": 0, "M2S_LEAF_MAX": 8})):
        with _lib.knobs(**kn):
            best = None
            for _ in range(5):
                t = M2STimings()
                out = generate_sdf(dv, topo, q, am, timings=t)
                if best is None or t.total_ms < best.total_ms:
                    best = t
        if ref is None:
            ref = out.clone()
        same = bool(torch.equal(out.view(torch.int32), ref.view(torch.int32)))
        line += f"  {name}: {best.total_ms:7.3f}{'' if same else ' DIFFERENT'} |"
    print(line, flush=True)
