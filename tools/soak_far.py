#!/usr/bin/env python3
"""Far-field soak: a small mesh in a grid (and query cloud) tens of mesh sizes wide — where a voxel sees many triangles at almost the same
distance and the pruning margins (prune_bound, the bounds' outward roundings) decide what is evaluated.  Every walk form against the
all-pairs kernel, bit for bit; both sign rules; off-centre boxes (large coordinates), anisotropic cells.
    python tools/soak_far.py [--seeds 300] [--first 0] [--seconds 300]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mesh_to_sdf_amd import AccelerationMethod, Grid, SignMethod, Topology, _lib, generate_grid_sdf, generate_sdf, meshes

FORMS = {"all pairs": {"M2S_BRUTE_MAX": 1e30}, "default": {"M2S_BRUTE_MAX": 0},
         "packets, queued, leaves of 2": {"M2S_BRUTE_MAX": 0, "M2S_LANE_WALK": 0, "M2S_DEFER": 3, "M2S_LEAF_MAX": 2},
         "packets, wave-wide, cut lists": {"M2S_BRUTE_MAX": 0, "M2S_LANE_WALK": 0, "M2S_DEFER": 0, "M2S_CUT_MIN_PACKETS": 8, "M2S_QUERY_CUT_MIN": 1},
         "packets, mixed, leaves of 16": {"M2S_BRUTE_MAX": 0, "M2S_LANE_WALK": 0, "M2S_DEFER": 2, "M2S_LEAF_MAX": 16},
         "lane walk": {"M2S_BRUTE_MAX": 0, "M2S_LANE_WALK": 1, "M2S_LEAF_MAX": 2},
         # round 6: cut lists in two levels; packets as groups of four waves
         "packets, two-level cut lists": {"M2S_BRUTE_MAX": 0, "M2S_LANE_WALK": 0, "M2S_CUT_MIN_PACKETS": 8, "M2S_CUT_COARSE": 1, "M2S_SPLIT": 0},
         "packet groups": {"M2S_BRUTE_MAX": 0, "M2S_LANE_WALK": 0, "M2S_GROUP": 1, "M2S_CUT_MIN_PACKETS": 4000000000}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=300)
    ap.add_argument("--first", type=int, default=0)
    ap.add_argument("--seconds", type=float, default=300)
    a = ap.parse_args()
    t0, bad, done = time.time(), 0, 0
    for seed in range(a.first, a.first + a.seeds):
        if time.time() - t0 > a.seconds:
            break
        rng = np.random.default_rng(991000 + seed)
        kind = rng.choice(["blob", "sheet", "blob-detail"])
        nt_target = int(10 ** rng.uniform(2.0, 4.0))
        su = max(4, int(np.sqrt(nt_target / 2 * 1.25)))
        sv = max(3, nt_target // (2 * su) + 1)
        v, idx = (meshes.sheet(su + 1, sv + 1) if kind == "sheet" else meshes.blob(su, sv, detail=kind == "blob-detail"))
        v = (v * np.float32(10 ** rng.uniform(-2, 2))).astype(np.float32)
        lo, hi = meshes.extended_bbox(v, 0.0)
        size = float(np.max(np.asarray(hi) - np.asarray(lo)))
        far = float(10 ** rng.uniform(0.5, 2.0))                              # box half-width in mesh sizes: 3 ... 100
        centre = 0.5 * (np.asarray(lo) + np.asarray(hi)) + rng.uniform(-1, 1, 3) * size * far * (0.8 if seed % 2 else 0.0)
        half = size * far * rng.uniform(0.5, 1.0, 3)
        counts = [int(c) for c in rng.integers(8, 41, 3)]
        grid = Grid.from_bounding_box((centre - half).astype(np.float32), (centre + half).astype(np.float32), counts)
        dv = torch.as_tensor(v, device="cuda")
        topo = Topology.TriangleList(torch.as_tensor(idx.astype(np.int64), device="cuda").to(torch.int32))
        line = f"seed {seed}: {kind} {len(idx) // 3} triangles, box {far:.1f} mesh sizes, grid {counts}:"
        for sign in (SignMethod.Raycast, SignMethod.Normal):
            ref = None
            for name, kn in FORMS.items():
                with _lib.knobs(**kn):
                    out = generate_grid_sdf(dv, topo, grid, sign)
                if ref is None:
                    ref = out.view(torch.int32).clone()
                    continue
                same = bool(torch.equal(out.view(torch.int32), ref))
                bad += 0 if same else 1
                if not same:
                    line += f" {sign.name} {name} DIFFERS"
        nq = int(rng.integers(200, 4000))
        q = torch.as_tensor((centre + rng.uniform(-1, 1, (nq, 3)) * half).astype(np.float32), device="cuda")
        for am in (AccelerationMethod.RtreeBvh, AccelerationMethod.Rtree):
            ref = None
            for name, kn in FORMS.items():
                with _lib.knobs(**kn):
                    out = generate_sdf(dv, topo, q, am)
                if ref is None:
                    ref = out.view(torch.int32).clone()
                    continue
                same = bool(torch.equal(out.view(torch.int32), ref))
                bad += 0 if same else 1
                if not same:
                    line += f" queries {am.kind} {name} DIFFERS"
        if "DIFFERS" in line or done % 50 == 0:
            print(line + ("" if "DIFFERS" in line else " ="), flush=True)
        done += 1
    print(f"## {done} cases ({len(FORMS) - 1} walk forms x 2 sign rules x grid + 2 query rules each) in {time.time() - t0:.0f} s, {bad} differences", flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
