"""A mesh at the documented limit (2^25 triangles): build + every walk against the all-pairs kernel on a small grid."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mesh_to_sdf_amd import _lib, Grid, M2STimings, SignMethod, Topology, generate_grid_sdf, meshes
slices, stacks = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (5792, 2897)
t0 = time.time(); v, idx = meshes.blob(slices, stacks, detail=True); print(f"{len(idx)//3} triangles ({time.time()-t0:.1f} s on the host), limit {1<<25}", flush=True)
lo, hi = meshes.extended_bbox(v, 0.1)
g = Grid.from_bounding_box(lo, hi, [20, 18, 22])
dv = torch.as_tensor(v, device="cuda"); topo = Topology.TriangleList(torch.as_tensor(idx.astype(np.int64), device="cuda").to(torch.int32))
for sign in (SignMethod.Raycast, SignMethod.Normal):
    t = M2STimings(); want = generate_grid_sdf(dv, topo, g, sign, algorithm=1, timings=t).cpu().numpy(); print(f"{sign.name}: all-pairs {t.total_ms:.1f} ms", flush=True)
    for name, env in (("default", {}), ("packet walk", {"M2S_LANE_WALK": "0"}), ("packet walk + cut lists", {"M2S_LANE_WALK": "0", "M2S_CUT_MIN_PACKETS": "1"}), ("lane walk", {"M2S_LANE_WALK": "1"})):
        with _lib.knobs(**env):
            t = M2STimings(); got = generate_grid_sdf(dv, topo, g, sign, timings=t).cpu().numpy()
        same = np.array_equal(got.view(np.uint32), want.view(np.uint32))
        print(f"  {name}: build {t.accel_build_ms:.2f} ms, total {t.total_ms:.2f} ms, identical {same}", flush=True)
