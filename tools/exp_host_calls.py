"""Wall time of the drop-in (host pointer) calls: numpy in, numpy out — what a caller of the reference's API sees."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mesh_to_sdf_amd import AccelerationMethod, Grid, Mesh, SignMethod, Topology, generate_grid_sdf, generate_sdf, meshes

v, idx = meshes.named("blob-100k")
lo, hi = meshes.extended_bbox(v, 0.1)
q = meshes.uniform_queries(lo, hi, 10_000_000)
for am, name in ((AccelerationMethod.RtreeBvh, "RtreeBvh"), (AccelerationMethod.Rtree, "Rtree")):
    ts = []
    for _ in range(4):
        t0 = time.perf_counter(); d = generate_sdf(v, Topology.TriangleList(idx), q, am); ts.append((time.perf_counter() - t0) * 1e3)
    print(f"generate_sdf 10M queries x 100k tris, {name}, host pointers: first {ts[0]:.1f} ms, then {min(ts[1:]):.1f} ms -> {10e6 / min(ts[1:]) / 1e3:.0f} Mq/s")
for n in (256, 512):
    g = Grid.from_bounding_box(lo, hi, [n] * 3)
    out = np.empty(n ** 3, np.float32)
    ts = []
    for _ in range(4):
        t0 = time.perf_counter(); generate_grid_sdf(v, Topology.TriangleList(idx), g, SignMethod.Raycast, out=out); ts.append((time.perf_counter() - t0) * 1e3)
    print(f"generate_grid_sdf {n}^3 x 100k tris Raycast, host pointers: first {ts[0]:.1f} ms, then {min(ts[1:]):.1f} ms -> {n**3 / min(ts[1:]) / 1e3:.0f} Mvoxels/s")
m = Mesh(v, Topology.TriangleList(idx))
g = Grid.from_bounding_box(lo, hi, [512] * 3)
ts = []
for _ in range(3):
    t0 = time.perf_counter(); m.generate_grid_sdf(g, SignMethod.Raycast, out=out); ts.append((time.perf_counter() - t0) * 1e3)
print(f"persistent mesh, 512^3, host out: {min(ts):.1f} ms")
