"""Host-side timeline of one rank's step of an N-GPU run (emulated on one GPU, no collectives)."""
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
rank = min(3, world - 1)
acc = np.zeros(5)
reps = 20
for rep in range(reps + 3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m = Mesh(dv, Topology.TriangleList(di)); t1 = time.perf_counter()
    run_pieces(m, g, SignMethod.Raycast, out, [piece_bounds(ch, world, rank) for ch in plan]); t2 = time.perf_counter()
    torch.cuda.synchronize(); t3 = time.perf_counter()
    t = m.drain_timings(); t4 = time.perf_counter()
    m.close(); t5 = time.perf_counter()
    if rep >= 3: acc += np.array([t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4]) * 1e3
acc /= reps
print(f"world {world} chunks {chunks} rank {rank}: create {acc[0]:.3f} (build on the GPU {t.accel_build_ms:.3f}) | enqueue pieces {acc[1]:.3f} | wait {acc[2]:.3f} | drain {acc[3]:.3f} | close {acc[4]:.3f} | total {acc.sum():.3f} ms")
