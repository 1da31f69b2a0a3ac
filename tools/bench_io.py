#!/usr/bin/env python3
"""Throughput of the SURVEY §8(f) kernels at the BASELINE size (512^3 cells, device resident), against the
HBM roofline: container encode / decode (9 B per cell algorithmic) and the cell ordering (radix sort).
usage: python tools/bench_io.py [--n 512] [--reps 5]   -> one JSON line per op"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from mesh_to_sdf_amd import Grid
from mesh_to_sdf_amd.client import merge_instances, order_cells_by_distance
from mesh_to_sdf_amd.serde import SerializeGrid, deserialize, serialize

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=512)
ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
n = a.n**3
grid = Grid.new([0.1, 0.2, 0.3], [0.01, 0.01, 0.01], [a.n] * 3)
d = torch.randn(n, device="cuda") * 0.4 + 0.3
PEAK = 8000.0


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
        del r
    return float(np.median(ts))


def line(op, ms, alg_bytes, extra=None):
    gbs = alg_bytes / ms / 1e6
    out = {"op": op, "cells": n, "ms": round(ms, 4), "algorithmic_bytes": alg_bytes, "achieved_GBps": round(gbs, 1),
           "frac_of_8TBps": round(gbs / PEAK, 4)}
    out.update(extra or {})
    print(json.dumps(out), flush=True)


ms = timed(lambda: serialize(SerializeGrid(grid, d), synchronous=False), a.reps)
line("sdf_encode_grid (k_encode_f32)", ms, 9 * n)
data = serialize(SerializeGrid(grid, d))
ms = timed(lambda: deserialize(data), a.reps)
line("sdf_decode (k_decode_f32, incl. envelope probe + error-flag sync)", ms, 9 * n)
del data
ms = timed(lambda: order_cells_by_distance(d, want_limits=False), a.reps)
line("order_cells_by_distance (rocPRIM radix sort of (key,index) pairs)", ms, 8 * n, {"Mcells_per_s": round(n / ms / 1e3, 1)})
ms = timed(lambda: order_cells_by_distance(d, want_limits=True), a.reps)
line("order_cells_by_distance + iso_limits", ms, 8 * n, {"Mcells_per_s": round(n / ms / 1e3, 1)})
# instance merge: 64 instances x 1M vertices
nv = 1 << 20
v = torch.randn(nv, 3, device="cuda")
i = torch.randint(0, nv, (3 * nv,), device="cuda", dtype=torch.int32)
m = np.eye(4, dtype=np.float32).reshape(-1)
inst = [(v, i, m)] * 64
ms = timed(lambda: merge_instances(inst), a.reps)
line("merge_instances (64 x 1M vertices, 3M indices) + bbox", ms, 64 * (nv * 24 + 3 * nv * 8), {"cells": 64 * nv})
# host-side flavours: numpy in -> bytes out / file out (PCIe + page cache inclusive)
import tempfile, time
from mesh_to_sdf_amd.serde import read_from_file, save_to_file
dh = d.cpu().numpy()
for label, fn in (("serialize numpy -> bytes", lambda: serialize(SerializeGrid(grid, dh))),):
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); r = fn(); ts.append((time.perf_counter() - t0) * 1e3); del r
    print(json.dumps({"op": label, "cells": n, "ms_first": round(ts[0], 1), "ms": round(min(ts[1:]), 1)}), flush=True)
with tempfile.TemporaryDirectory() as td:
    path = os.path.join(td, "sdf.bin")
    for label, src in (("save_to_file from device tensor", d), ("save_to_file from numpy", dh)):
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); save_to_file(SerializeGrid(grid, src), path); ts.append((time.perf_counter() - t0) * 1e3)
        print(json.dumps({"op": label, "cells": n, "file_MB": round(os.path.getsize(path) / 1e6, 1), "ms_first": round(ts[0], 1), "ms": round(min(ts[1:]), 1)}), flush=True)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); de = read_from_file(path); ts.append((time.perf_counter() - t0) * 1e3)
    print(json.dumps({"op": "read_from_file -> numpy", "cells": n, "ms_first": round(ts[0], 1), "ms": round(min(ts[1:]), 1)}), flush=True)
    assert np.array_equal(de.distances.view(np.uint32), dh.view(np.uint32))
