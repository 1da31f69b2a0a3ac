import sys; sys.path.insert(0, '.')
import numpy as np, oracle as orc
from mesh_to_sdf_amd import *
from mesh_to_sdf_amd import meshes
F = np.float32
v = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [2, 2, 2], [3, 2.5, 2]], F)
q = meshes.uniform_queries([-1, -1, -1], [4, 4, 4], 5000)
cases = {"reg+point": [0, 1, 2, 3, 3, 3], "reg+a==b": [0, 1, 2, 3, 3, 4], "reg+b==c": [0, 1, 2, 3, 4, 4],
         "reg+a==c": [0, 1, 2, 4, 3, 4], "point only": [3, 3, 3], "a==b only": [3, 3, 4], "reg only": [0, 1, 2],
         "two reg": [0, 1, 2, 2, 3, 4]}
for name, idx in cases.items():
    want = orc.generate_sdf(v, idx, q, accel=1, sign=1)
    for alg in (0, 1):
        got = generate_sdf(v, Topology.TriangleList(idx), q, AccelerationMethod.Bvh(SignMethod.Normal), algorithm=alg)
        print(f"{name:12s} alg={alg} mismatches={int(np.sum(got.view(np.uint32) != want.view(np.uint32)))} "
              f"sign-only={int(np.sum((np.abs(got) == np.abs(want)) & (got != want)))}")
