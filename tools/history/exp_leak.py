"""Steady-state memory check: device memory in use before / after many calls of every kind (hipMemGetInfo through torch)."""
import sys
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import numpy as np, torch
from mesh_to_sdf_amd import *
from mesh_to_sdf_amd import meshes
from mesh_to_sdf_amd.distributed import generate_grid_sdf_sharded
v, idx = meshes.named('blob-100k'); lo, hi = meshes.extended_bbox(v, 0.1)
dv = torch.as_tensor(v, device='cuda'); di = torch.as_tensor(idx.astype(np.int64), device='cuda').to(torch.int32)
topo = Topology.TriangleList(di)
q = torch.as_tensor(meshes.uniform_queries(lo, hi, 200000), device='cuda')
q3 = torch.as_tensor(meshes.uniform_queries(lo, hi, 3000000), device='cuda')
def used():
    torch.cuda.synchronize(); f, t = torch.cuda.mem_get_info(); return (t - f) / 2**20
def round_():
    for n in (96, 160, 128):
        g = Grid.from_bounding_box(lo, hi, [n] * 3)
        out = torch.empty(n ** 3, device='cuda')
        generate_grid_sdf(dv, topo, g, SignMethod.Raycast, out=out)
        generate_grid_sdf(dv, topo, g, SignMethod.Normal, out=out)
        generate_grid_sdf(v, Topology.TriangleList(idx), g, SignMethod.Raycast)            # host pointers
        generate_grid_sdf_sharded(dv, topo, g, SignMethod.Raycast, out=out, chunks=4)    # persistent mesh, pieces on two streams
        del out
    generate_sdf(dv, topo, q, AccelerationMethod.RtreeBvh)                               # lane walk (sparse set)
    generate_sdf(dv, topo, q, AccelerationMethod.Rtree)
    generate_sdf(dv, topo, q3, AccelerationMethod.RtreeBvh)                              # bucket packets + per-packet cut lists
    generate_sdf_multi(dv, topo, q, AccelerationMethod.RtreeBvh, devices=[0, 0])
    g32 = Grid.from_bounding_box(lo, hi, [32] * 3)
    generate_grid_sdf(dv, topo, g32, SignMethod.Raycast)                                 # lane walk with work sharing
    generate_grid_sdf_multi(dv, topo, Grid.from_bounding_box(lo, hi, [128] * 3), SignMethod.Raycast, devices=[0, 0])
for _ in range(3): round_()
torch.cuda.empty_cache(); a = used()
for _ in range(40): round_()
torch.cuda.empty_cache(); b = used()
print(f"device memory in use: {a:.1f} MiB after warm-up, {b:.1f} MiB after 40 more rounds of 19 calls each -> growth {b - a:+.1f} MiB")
