#!/usr/bin/env python3
"""The shapes of the reference's own criterion benches (mesh_to_sdf/benches/generate_grid_sdf.rs:8-34,94-129; generate_sdf.rs:34-49,
126-138) on this box: an 11 200-triangle mesh (knight.glb has 11 184) in its TIGHT bounding box at 16^3 and 100^3, and the query
lattice of step 0.01 x the box (~1 M points) — GPU call (device-resident, best of 5) beside the CPU port of the reference's
algorithm (oracle, 1 thread and all threads, best of 3).  The .glb itself stays in the reference tree; the synthetic mesh has
its triangle count."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import oracle as orc  # noqa: E402
from mesh_to_sdf_amd import (AccelerationMethod, Grid, M2STimings, SignMethod, Topology, generate_grid_sdf, generate_sdf,  # noqa: E402
                             meshes)

v, idx = meshes.named("blob-11k")
lo, hi = v.min(0), v.max(0)
dv = torch.as_tensor(v, device="cuda")
topo = Topology.TriangleList(torch.as_tensor(idx.astype(np.int64), device="cuda").to(torch.int32))
cores = orc.hardware_threads()
print(f"# blob-11k: {len(idx) // 3} triangles, tight bounding box; host threads {cores}")
for n in (16, 100):
    grid = Grid.from_bounding_box(lo, hi, [n, n, n])
    first, size, cnt = meshes.grid_from_bounding_box(lo, hi, [n] * 3)
    for sign, sname in ((SignMethod.Normal, "normal"), (SignMethod.Raycast, "raycast")):
        out = torch.empty(n ** 3, dtype=torch.float32, device="cuda")
        gpu = 1e9
        for _ in range(5):
            t0 = time.perf_counter()
            generate_grid_sdf(dv, topo, grid, sign, out=out)
            torch.cuda.synchronize()
            gpu = min(gpu, time.perf_counter() - t0)
        cpu = {}
        for th in (1, cores):
            best = 1e9
            for _ in range(3 if n == 16 else 1):
                t0 = time.perf_counter()
                orc.generate_grid_sdf(v, idx, first, size, cnt, sign=0 if sign == SignMethod.Raycast else 1, semantics=orc.PROPAGATE, heaps=th, threads=th)
                best = min(best, time.perf_counter() - t0)
            cpu[th] = best
        print(f"generate_grid_sdf_{sname} {n}^3: GPU call {gpu * 1e3:8.3f} ms (wall, incl. the accel build) | CPU port 1 thread {cpu[1] * 1e3:9.1f} ms, "
              f"{cores} threads {cpu[cores] * 1e3:9.1f} ms", flush=True)
step = 0.01 * float((hi - lo).max())
ax = [np.arange(lo[k], hi[k], step, dtype=np.float32) for k in range(3)]
q = np.stack(np.meshgrid(*ax, indexing="ij"), -1).reshape(-1, 3).astype(np.float32)
dq = torch.as_tensor(q, device="cuda")
for am, name in ((AccelerationMethod.Bvh(SignMethod.Normal), "bvh_normal"), (AccelerationMethod.Bvh(SignMethod.Raycast), "bvh_raycast"),
                 (AccelerationMethod.Rtree, "rtree"), (AccelerationMethod.RtreeBvh, "rtree_bvh")):
    gpu = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        generate_sdf(dv, topo, dq, am)
        torch.cuda.synchronize()
        gpu = min(gpu, time.perf_counter() - t0)
    t0 = time.perf_counter()
    sub = q[:: max(1, len(q) // 20000)]
    orc.generate_sdf(v, idx, sub, accel=am.kind, sign=int(am.sign.value), threads=1, fast=True)
    cpu = (time.perf_counter() - t0) * len(q) / len(sub)
    print(f"generate_sdf_{name} {len(q)} lattice queries: GPU call {gpu * 1e3:8.3f} ms | CPU port (BVH-accelerated exact search, 1 thread, extrapolated from {len(sub)} queries) {cpu * 1e3:9.1f} ms", flush=True)
