#!/usr/bin/env python3
"""Small query sets of generate_sdf: the tree-less k_brute_split_q (queries x triangles <= M2S_BRUTE_MAX) against build + walk, whole call,
device resident, every AccelerationMethod.   python tools/exp_small_queries.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mesh_to_sdf_amd import AccelerationMethod, M2STimings, SignMethod, Topology, _lib, generate_sdf, meshes  # noqa: E402

ACCELS = [("RtreeBvh", AccelerationMethod.RtreeBvh), ("Bvh(Raycast)", AccelerationMethod.Bvh(SignMethod.Raycast)), ("Bvh(Normal)", AccelerationMethod.Bvh(SignMethod.Normal)),
          ("Rtree", AccelerationMethod.Rtree)]
for name, (sl, st_) in (("blob-11k", (80, 71)), ("blob-100k", (250, 201))):
    v, idx = meshes.blob(sl, st_)
    lo, hi = meshes.extended_bbox(v, 0.1)
    dv = torch.as_tensor(v, device="cuda")
    topo = Topology.TriangleList(torch.as_tensor(idx.astype(np.int64), device="cuda").to(torch.int32))
    for nq in (1, 64, 1000, 10000):
        q = torch.as_tensor(meshes.uniform_queries(lo, hi, nq), device="cuda")
        for aname, am in ACCELS:
            line = f"{name} ({idx.size // 3} triangles) x {nq:>5} queries {aname:13}:"
            outs = []
            for label, limit in (("all pairs", 1e30), ("tree", 0)):
                with _lib.knobs(M2S_BRUTE_MAX=limit):
                    best = None
                    for _ in range(7):
                        t = M2STimings()
                        out = generate_sdf(dv, topo, q, am, timings=t)
                        if best is None or t.total_ms < best:
                            best = t.total_ms
                outs.append(out.clone())
                line += f"  {label} {best:.3f} ms"
            same = bool(torch.equal(outs[0].view(torch.int32), outs[1].view(torch.int32)))
            print(line + f"  identical: {same}", flush=True)
