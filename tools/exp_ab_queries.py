"""A/B helper for generic queries: whole-call time of generate_sdf (10 M uniform queries x blob-100k by default) with the library M2S_LIB names."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mesh_to_sdf_amd import AccelerationMethod, M2STimings, Topology, generate_sdf, meshes
nq = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
v, idx = meshes.named("blob-100k"); lo, hi = meshes.extended_bbox(v, 0.1)
dq = torch.as_tensor(meshes.uniform_queries(lo, hi, nq), device="cuda")
dv = torch.as_tensor(v, device="cuda"); topo = Topology.TriangleList(torch.as_tensor(idx.astype(np.int64), device="cuda").to(torch.int32))
for am in (AccelerationMethod.RtreeBvh, AccelerationMethod.Rtree):
    best = 1e9
    for _ in range(5):
        t = M2STimings(); out = generate_sdf(dv, topo, dq, am, timings=t); best = min(best, t.total_ms)
    print(f"{os.environ.get('M2S_LIB', 'default lib')}: {nq} queries accel {am.kind}: {best:.3f} ms; checksum {float(out.double().sum()):.6f}", flush=True)
