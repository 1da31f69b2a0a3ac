import sys, time
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import numpy as np, oracle as orc
from mesh_to_sdf_amd import meshes
v, idx = meshes.named('blob-100k'); lo, hi = meshes.extended_bbox(v, 0.1)
print("cores", orc.hardware_threads())
for n in (128, 256):
    first, size, cnt = meshes.grid_from_bounding_box(lo, hi, [n] * 3)
    for th in (8, 16, 32, 64, 128, 256):
        if th > orc.hardware_threads(): continue
        t = time.perf_counter()
        out, st = orc.generate_grid_sdf(v, idx, first, size, cnt, sign=0, semantics=orc.PROPAGATE, heaps=th, threads=th, return_stats=True)
        dt = time.perf_counter() - t
        print(f"n={n} threads={th}: {dt:.2f}s {n**3/dt/1e6:.2f} Mvox/s pops={int(st[1])}", flush=True)
