#!/usr/bin/env python3
"""First call of a fresh process through the drop-in (host pointer) entry points, with and without m2s_warmup — the reference's
documented usage is ONE call per process (examples/demo.rs:29-54).  Each case runs in a process of its own.
    python tools/exp_first_call.py            # all cases
    python tools/exp_first_call.py grid 1     # one case (used by the driver loop below): kind, warmup flag"""
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def one(kind, warm):
    import numpy as np

    t_imp = time.perf_counter()
    from mesh_to_sdf_amd import AccelerationMethod, Grid, SignMethod, Topology, generate_grid_sdf, generate_sdf, meshes, warmup
    t_imp = (time.perf_counter() - t_imp) * 1e3
    v, idx = meshes.named("blob-100k")
    lo, hi = meshes.extended_bbox(v, 0.1)
    if kind == "grid":
        g = Grid.from_bounding_box(lo, hi, [512] * 3)
        out = np.empty(512 ** 3, np.float32)
        call = lambda: generate_grid_sdf(v, Topology.TriangleList(idx), g, SignMethod.Raycast, out=out)
        ws, ring = 700 << 20, 32 << 20
    else:
        q = meshes.uniform_queries(lo, hi, 10_000_000)
        call = lambda: generate_sdf(v, Topology.TriangleList(idx), q, AccelerationMethod.RtreeBvh)
        ws, ring = 1600 << 20, 32 << 20
    t_w = 0.0
    if warm:
        t0 = time.perf_counter()
        warmup(0, ws, ring)
        t_w = (time.perf_counter() - t0) * 1e3
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        call()
        ts.append((time.perf_counter() - t0) * 1e3)
    print(f"{kind:5s} warmup={warm}: import {t_imp:.0f} ms, m2s_warmup {t_w:.1f} ms, first call {ts[0]:.1f} ms, then {min(ts[1:]):.1f} ms", flush=True)


if __name__ == "__main__":
    if len(sys.argv) == 3:
        one(sys.argv[1], int(sys.argv[2]))
    else:
        for kind in ("grid", "query"):
            for warm in (0, 1):
                subprocess.run([sys.executable, os.path.abspath(__file__), kind, str(warm)], env=dict(os.environ, M2S_HOST_TIMES="1"))
