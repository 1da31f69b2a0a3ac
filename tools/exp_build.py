#!/usr/bin/env python3
"""Build time of the LBVH alone (m2s_mesh_create: records + sort + hierarchy + treelets + bounds), per mesh, for the lean build and
the round-2 sequence (M2S_BUILD=1 / 0):  python tools/exp_build.py [--meshes blob-100k,blob-1M] [--iters 30]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--meshes", default="blob-11k,blob-100k,blob-1M")
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--builds", default="0,1")
    args = ap.parse_args()
    import torch

    from mesh_to_sdf_amd import Mesh, Topology, meshes

    for name in args.meshes.split(","):
        if name == "blob-11k":
            v, idx = meshes.blob(80, 71)
        else:
            v, idx = meshes.named(name)
        dv = torch.as_tensor(v, device="cuda")
        topo = Topology.TriangleList(torch.as_tensor(idx.astype(np.int64), device="cuda").to(torch.int32))
        for b in ("lean",):
            ts, dev = [], []
            for i in range(args.iters + 3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                m = Mesh(dv, topo)
                t1 = time.perf_counter()
                d = m.drain_timings().accel_build_ms
                m.close()
                if i >= 3:
                    ts.append((t1 - t0) * 1e3)
                    dev.append(d)
            print(f"{name} ({idx.size // 3} triangles) {b} build: device build median {np.median(dev):.3f} ms (min {np.min(dev):.3f}); "
                  f"m2s_mesh_create wall median {np.median(ts):.3f} ms", flush=True)


if __name__ == "__main__":
    main()
