"""Phase timings over unusual regimes (looking for pathologies): open surface, mixed triangle sizes, anisotropic
grids, triangles much smaller than voxels, queries far outside the mesh."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mesh_to_sdf_amd import AccelerationMethod, Grid, M2STimings, SignMethod, Topology, generate_grid_sdf, generate_sdf, meshes


def run_grid(name, v, idx, counts, sign, frac=0.1):
    lo, hi = meshes.extended_bbox(v, frac)
    g = Grid.from_bounding_box(lo, hi, counts)
    dv, di = torch.as_tensor(v, device="cuda"), torch.as_tensor(idx.astype(np.int64), device="cuda").to(torch.int32)
    out = torch.empty(int(np.prod(counts)), device="cuda")
    best = None
    for _ in range(3):
        t = M2STimings(); generate_grid_sdf(dv, Topology.TriangleList(di), g, sign, out=out, timings=t)
        if best is None or t.total_ms < best.total_ms: best = t
    n = int(np.prod(counts))
    print(f"{name} {counts} {sign.name}: total {best.total_ms:.3f} ms (build {best.accel_build_ms:.3f}, sign {best.sign_ms:.3f}, seeds {best.seed_ms:.3f}, "
          f"distance {best.distance_ms:.3f}) -> {n / best.total_ms / 1e3:.0f} Mvox/s", flush=True)


def run_q(name, v, idx, q, am):
    dv, di = torch.as_tensor(v, device="cuda"), torch.as_tensor(idx.astype(np.int64), device="cuda").to(torch.int32)
    dq = torch.as_tensor(q, device="cuda")
    best = None
    for _ in range(3):
        t = M2STimings(); generate_sdf(dv, Topology.TriangleList(di), dq, am, timings=t)
        if best is None or t.total_ms < best.total_ms: best = t
    print(f"{name} {q.shape[0]} queries accel={am.kind}: total {best.total_ms:.3f} ms (build {best.accel_build_ms:.3f}, distance {best.distance_ms:.3f}) -> {q.shape[0] / best.total_ms / 1e3:.0f} Mq/s", flush=True)


v, idx = meshes.named("sheet-100k")
run_grid("sheet-100k", v, idx, [512] * 3, SignMethod.Raycast)
run_grid("sheet-100k", v, idx, [512] * 3, SignMethod.Normal)
# mixed sizes: blob-100k plus 12 huge triangles (a big box around it)
bv, bi = meshes.named("blob-100k")
lo, hi = bv.min(0) * 3, bv.max(0) * 3
c = np.array([[lo[0], lo[1], lo[2]], [hi[0], lo[1], lo[2]], [hi[0], hi[1], lo[2]], [lo[0], hi[1], lo[2]],
              [lo[0], lo[1], hi[2]], [hi[0], lo[1], hi[2]], [hi[0], hi[1], hi[2]], [lo[0], hi[1], hi[2]]], np.float32)
f = np.array([0, 2, 1, 0, 3, 2, 4, 5, 6, 4, 6, 7, 0, 1, 5, 0, 5, 4, 2, 3, 7, 2, 7, 6, 1, 2, 6, 1, 6, 5, 3, 0, 4, 3, 4, 7], np.uint32)
mv = np.concatenate([bv, c]); mi = np.concatenate([bi.reshape(-1), f + bv.shape[0]]).astype(np.uint32)
run_grid("blob-100k + box of 12 huge triangles", mv, mi, [512] * 3, SignMethod.Raycast, frac=0.02)
run_grid("blob-100k", bv, bi, [2048, 64, 64], SignMethod.Raycast)
run_grid("blob-100k", bv, bi, [64, 64, 2048], SignMethod.Raycast)
v1, i1 = meshes.named("blob-1M")
run_grid("blob-1M", v1, i1, [128] * 3, SignMethod.Raycast)
run_grid("blob-1M", v1, i1, [128] * 3, SignMethod.Normal)
lo, hi = meshes.extended_bbox(bv, 5.0)
run_q("blob-100k, queries in a box 11x the mesh", bv, bi, meshes.uniform_queries(lo, hi, 10_000_000), AccelerationMethod.RtreeBvh)
run_q("blob-100k, queries in a box 11x the mesh", bv, bi, meshes.uniform_queries(lo, hi, 10_000_000), AccelerationMethod.Bvh(SignMethod.Normal))
