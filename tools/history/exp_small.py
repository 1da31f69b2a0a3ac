"""Latency of small grids (the reference client's interactive case): launch-bound?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mesh_to_sdf_amd import Grid, M2STimings, Mesh, SignMethod, Topology, generate_grid_sdf, meshes
d = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "suzanne.npz"))
v, idx = d["vertices"].astype(np.float32), d["indices"].astype(np.uint32)
lo, hi = meshes.extended_bbox(v, 0.1)
dv, di = torch.as_tensor(v, device="cuda"), torch.as_tensor(idx.astype(np.int64), device="cuda").to(torch.int32)
for n in (16, 32, 64, 128):
    g = Grid.from_bounding_box(lo, hi, [n] * 3)
    out = torch.empty(n ** 3, device="cuda")
    for name, fn in (("one-shot", lambda: generate_grid_sdf(dv, Topology.TriangleList(di), g, SignMethod.Raycast, out=out)),):
        fn(); torch.cuda.synchronize()
        ts = []
        for _ in range(20):
            t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        t = M2STimings(); generate_grid_sdf(dv, Topology.TriangleList(di), g, SignMethod.Raycast, out=out, timings=t)
        print(f"suzanne {n}^3 {name}: wall median {np.median(ts):.3f} ms (min {min(ts):.3f}); device total {t.total_ms:.3f} ms (build {t.accel_build_ms:.3f}, sign {t.sign_ms:.3f}, seeds {t.seed_ms:.3f}, distance {t.distance_ms:.3f})")
    m = Mesh(dv, Topology.TriangleList(di))
    m.generate_grid_sdf(g, SignMethod.Raycast, out=out); torch.cuda.synchronize()
    ts = []
    for _ in range(20):
        t0 = time.perf_counter(); m.generate_grid_sdf(g, SignMethod.Raycast, out=out); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print(f"suzanne {n}^3 persistent mesh (same grid: planes cached): wall median {np.median(ts):.3f} ms (min {min(ts):.3f})")
    m.close()
