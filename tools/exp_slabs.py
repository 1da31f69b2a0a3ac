"""Per-piece timings as an 8-GPU, 4-chunk run would issue them (rank r of 8), on one GPU."""
import sys, time
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import numpy as np, torch
from mesh_to_sdf_amd import *
from mesh_to_sdf_amd import meshes
from mesh_to_sdf_amd.distributed import chunk_plan, piece_bounds, run_pieces
v, idx = meshes.named('blob-100k'); lo, hi = meshes.extended_bbox(v, 0.1)
n = 512
g = Grid.from_bounding_box(lo, hi, [n] * 3)
dv = torch.as_tensor(v, device='cuda'); di = torch.as_tensor(idx.astype(np.int64), device='cuda').to(torch.int32)
out = torch.empty(n ** 3, device='cuda')
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
chunks = int(sys.argv[2]) if len(sys.argv) > 2 else 4
plan = chunk_plan(n, world, chunks)
for rank in (0, 3, 7):
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        m = Mesh(dv, Topology.TriangleList(di))
        run_pieces(m, g, SignMethod.Raycast, out, [piece_bounds(ch, world, rank) for ch in plan])
        t = m.drain_timings(); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
        m.close()
    print(f"world {world} chunks {chunks} rank {rank}: wall {dt:.2f} ms; build {t.accel_build_ms:.2f}; k_packet launches {t.distance_launches} sum {t.distance_ms:.2f} ms")
# sync one-shot per piece, to see phase split
a, b = piece_bounds(plan[1], world, 3)
t = M2STimings(); generate_grid_sdf(dv, Topology.TriangleList(di), g, SignMethod.Raycast, x_slab=(a, b), out=out, timings=t)
print(f"one piece [{a},{b}): build {t.accel_build_ms:.3f} sign {t.sign_ms:.3f} seed {t.seed_ms:.3f} distance {t.distance_ms:.3f} total {t.total_ms:.3f}")
