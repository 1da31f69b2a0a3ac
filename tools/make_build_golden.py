#!/usr/bin/env python3
"""Writes tests/golden/build_digests.json: FNV-1a digests of the resident arrays of the LBVH (m2s_debug_mesh_digest) for every mesh of
tests/build_cases.py, computed on an MI355X.  First made with the round-3 library, whose lean build and round-2 kernel sequence agreed on
every one of them byte for byte (tests/test_gpu_build.py of that round); the tests now hold any later build to these trees.

    python tools/make_build_golden.py [out.json]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import build_cases  # noqa: E402
from mesh_to_sdf_amd import Mesh, Topology  # noqa: E402


def digest(v, idx):
    dv = torch.as_tensor(np.ascontiguousarray(v, np.float32), device="cuda")
    topo = Topology.TriangleList(torch.as_tensor(idx.astype(np.int64), device="cuda").to(torch.int32))
    with Mesh(dv, topo) as m:
        return m.debug_digest()


out = {}
both = os.environ.get("M2S_GOLDEN_CHECK_OLD") == "1"      # round-3 library only: also run M2S_BUILD=0 and insist on the same tree
for name, v, idx in build_cases.cases():
    d = digest(v, idx)
    if both:
        os.environ["M2S_BUILD"] = "0"
        old = digest(v, idx)
        os.environ.pop("M2S_BUILD")
        assert old == d, f"{name}: the two builds differ"
    out[name] = [f"{x:016x}" for x in d]
    print(name, out[name][3], flush=True)
path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden", "build_digests.json")
with open(path, "w") as f:
    json.dump(out, f, indent=1)
print("wrote", path, len(out), "cases")
