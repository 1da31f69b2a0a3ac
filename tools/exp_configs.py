"""Timing / sanity of the other BASELINE.json configs on one GPU."""
import sys, time
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import numpy as np, torch
from mesh_to_sdf_amd import *
from mesh_to_sdf_amd import meshes

def grid_case(mesh, n, sign, slab=None):
    v, idx = meshes.named(mesh); lo, hi = meshes.extended_bbox(v, 0.1)
    g = Grid.from_bounding_box(lo, hi, [n] * 3)
    dv = torch.as_tensor(v, device='cuda'); di = torch.as_tensor(idx.astype(np.int64), device='cuda').to(torch.int32)
    out = torch.empty(n ** 3, device='cuda')
    best = None
    for r in range(2):
        t = M2STimings()
        generate_grid_sdf(dv, Topology.TriangleList(di), g, sign, timings=t, out=out, x_slab=slab)
        if best is None or t.total_ms < best.total_ms: best = t
    units = best.n_units
    print(f"{mesh} {n}^3 {sign.name} slab={slab}: build {best.accel_build_ms:.2f} sign {best.sign_ms:.2f} seed {best.seed_ms:.2f} "
          f"distance {best.distance_ms:.2f} total {best.total_ms:.2f} ms -> {units / best.total_ms / 1e3:.0f} Mvox/s; "
          f"neg frac {float((out < 0).float().mean()) if slab is None else -1:.4f}", flush=True)
    del out

def query_case(mesh, nq, am):
    v, idx = meshes.named(mesh); lo, hi = meshes.extended_bbox(v, 0.1)
    q = meshes.uniform_queries(lo, hi, nq)
    dv = torch.as_tensor(v, device='cuda'); di = torch.as_tensor(idx.astype(np.int64), device='cuda').to(torch.int32)
    dq = torch.as_tensor(q, device='cuda')
    best = None
    for r in range(2):
        t = M2STimings()
        out = generate_sdf(dv, Topology.TriangleList(di), dq, am, timings=t)
        if best is None or t.total_ms < best.total_ms: best = t
    print(f"{mesh} {nq} queries accel={am.kind}: build {best.accel_build_ms:.2f} distance {best.distance_ms:.2f} total {best.total_ms:.2f} ms "
          f"-> {nq / best.total_ms / 1e3:.1f} Mq/s; neg frac {float((out < 0).float().mean()):.4f}", flush=True)

which = sys.argv[1:] or ['c2', 'c3', 'c4', 'c5']
if 'c2' in which: grid_case('blob-100k', 256, SignMethod.Raycast)
if 'c3' in which:
    query_case('blob-100k', 10_000_000, AccelerationMethod.RtreeBvh)
    query_case('blob-100k', 10_000_000, AccelerationMethod.Rtree)
if 'c4' in which:
    grid_case('blob-1M', 512, SignMethod.Raycast)
    grid_case('blob-1M', 512, SignMethod.Raycast, slab=(0, 64))
if 'c5' in which:
    grid_case('sheet-100k', 1024, SignMethod.Normal, slab=(0, 128))
    grid_case('sheet-100k', 1024, SignMethod.Normal, slab=(448, 576))
if 'c5full' in which: grid_case('sheet-100k', 1024, SignMethod.Normal)
