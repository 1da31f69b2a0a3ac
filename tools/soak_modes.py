#!/usr/bin/env python3
"""Soak: random structured meshes (blobs / sheets of 200 … 400 000 triangles, some with duplicated or degenerate triangles), random grids
(8 … 200 cells per axis, anisotropic, some x-slabs / interleaved slabs), both sign rules, device-resident: the default call, the packet
walk with cut lists forced, the lane walk and (where cells x triangles allows) the all-pairs kernel must agree bit for bit; generic
queries likewise.   python tools/soak_modes.py [--seeds 60] [--first 0] [--seconds 600]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mesh_to_sdf_amd import AccelerationMethod, Grid, SignMethod, Topology, _lib, generate_grid_sdf, generate_sdf, interleaved_slab, meshes

MODES = {"default": {}, "packet + cut lists": {"M2S_CUT_MIN_PACKETS": 8, "M2S_QUERY_CUT_MIN": 1, "M2S_LANE_WALK": 0, "M2S_BRUTE_MAX": 0, "M2S_SPLIT": 0, "M2S_TREELETS": 0},
         "packet + split": {"M2S_LANE_WALK": 0, "M2S_BRUTE_MAX": 0, "M2S_SPLIT": 2, "M2S_SPLIT_BUDGET": 40, "M2S_TREELETS": 1},
         "packet + cut lists + split": {"M2S_CUT_MIN_PACKETS": 8, "M2S_QUERY_CUT_MIN": 1, "M2S_LANE_WALK": 0, "M2S_BRUTE_MAX": 0, "M2S_SPLIT": 2, "M2S_SPLIT_BUDGET": 60},
         "lane walk": {"M2S_LANE_WALK": 1, "M2S_BRUTE_MAX": 0, "M2S_LEAF_MAX": 2},
         # the packet walk's pre-tests and exact evaluations: wave-wide at once (round 3), evaluations queued, both queued, queued + direct
         "packet, direct evaluations": {"M2S_LANE_WALK": 0, "M2S_BRUTE_MAX": 0, "M2S_DEFER": 0},
         "packet, queued evaluations, leaves of 2": {"M2S_LANE_WALK": 0, "M2S_BRUTE_MAX": 0, "M2S_DEFER": 1, "M2S_LEAF_MAX": 2},
         "packet, leaves of 16": {"M2S_LANE_WALK": 0, "M2S_BRUTE_MAX": 0, "M2S_LEAF_MAX": 16},
         "packet + cut lists, queued pre-tests and evaluations": {"M2S_CUT_MIN_PACKETS": 8, "M2S_QUERY_CUT_MIN": 1, "M2S_LANE_WALK": 0, "M2S_BRUTE_MAX": 0, "M2S_SPLIT": 0, "M2S_DEFER": 3},
         # round 6: cut lists made in two levels (k_cut LEVEL 1 + 2), packets as groups of four waves that share their minima in LDS
         "packet + two-level cut lists": {"M2S_CUT_MIN_PACKETS": 8, "M2S_CUT_COARSE": 1, "M2S_LANE_WALK": 0, "M2S_BRUTE_MAX": 0, "M2S_SPLIT": 0},
         "packet + two-level cut lists + split": {"M2S_CUT_MIN_PACKETS": 8, "M2S_CUT_COARSE": 1, "M2S_LANE_WALK": 0, "M2S_BRUTE_MAX": 0, "M2S_SPLIT": 2, "M2S_SPLIT_BUDGET": 60},
         "packet groups": {"M2S_GROUP": 1, "M2S_LANE_WALK": 0, "M2S_BRUTE_MAX": 0, "M2S_CUT_MIN_PACKETS": 4000000000},
         "packet groups, leaves of 2": {"M2S_GROUP": 1, "M2S_LANE_WALK": 0, "M2S_BRUTE_MAX": 0, "M2S_CUT_MIN_PACKETS": 4000000000, "M2S_LEAF_MAX": 2},
         "packet + cut lists + split, queued + direct evaluations": {"M2S_CUT_MIN_PACKETS": 8, "M2S_QUERY_CUT_MIN": 1, "M2S_LANE_WALK": 0, "M2S_BRUTE_MAX": 0, "M2S_SPLIT": 2,
                                                                     "M2S_SPLIT_BUDGET": 60, "M2S_DEFER": 2}}


def with_mode(env, fn):
    with _lib.knobs(**env):
        return fn()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=60)
    ap.add_argument("--first", type=int, default=0)
    ap.add_argument("--seconds", type=float, default=600)
    a = ap.parse_args()
    t0, bad, done = time.time(), 0, 0
    for seed in range(a.first, a.first + a.seeds):
        if time.time() - t0 > a.seconds:
            break
        rng = np.random.default_rng(77000 + seed)
        kind = rng.choice(["blob", "blob-detail", "sheet"])
        nt_target = int(10 ** rng.uniform(2.3, 5.6))
        su = max(4, int(np.sqrt(nt_target / 2 * 1.25)))
        sv = max(3, nt_target // (2 * su) + 1)
        v, idx = (meshes.sheet(su + 1, sv + 1) if kind == "sheet" else meshes.blob(su, sv, detail=kind == "blob-detail"))
        v = (v * np.float32(10 ** rng.uniform(-2, 2)) + (rng.uniform(-1, 1, 3) * 10 ** rng.uniform(-1, 2) * (seed % 3 == 0))).astype(np.float32)
        tri = idx.reshape(-1, 3).copy()
        if seed % 4 == 1:
            tri[::9, 2] = tri[::9, 1]                                  # degenerate
            tri[3::13] = tri[2::13][: tri[3::13].shape[0]]            # duplicated
        idx = tri.reshape(-1).astype(np.uint32)
        lo, hi = meshes.extended_bbox(v, float(rng.uniform(0.0, 0.4)))
        counts = [int(c) for c in np.clip(10 ** rng.uniform(0.9, 2.3, 3), 8, 200)]
        if seed % 5 == 0:
            counts = [int(rng.choice([64, 128, 192]))] * 3                  # shapes the interleaved slabs accept
        grid = Grid.from_bounding_box(lo, hi, counts)
        sign = SignMethod.Raycast if rng.random() < 0.6 else SignMethod.Normal
        dv = torch.as_tensor(v, device="cuda")
        topo = Topology.TriangleList(torch.as_tensor(idx.astype(np.int64), device="cuda").to(torch.int32))
        cells = counts[0] * counts[1] * counts[2]
        nt = len(idx) // 3
        kw, what = {}, "whole grid"
        if seed % 5 == 0:
            world = int(rng.choice([2, 4]))
            r = int(rng.integers(0, world))
            sl = interleaved_slab(grid, world, r)
            if sl is not None and sl[2]:
                kw, what = {"x_slab": (sl[0], sl[1]), "x_period": sl[2]}, f"interleaved slab {sl}"
        elif seed % 5 == 3:
            x0 = int(rng.integers(0, counts[0]))
            x1 = int(rng.integers(x0 + 1, counts[0] + 1))
            kw, what = {"x_slab": (x0, x1)}, f"slab ({x0}, {x1})"
        res = {}
        for name, env in MODES.items():
            def run():
                out = torch.full((cells,), float("nan"), device="cuda")
                generate_grid_sdf(dv, topo, grid, sign, out=out, **kw)
                return out
            res[name] = with_mode(env, run)
        if cells * nt <= 3e10:
            def run_b():
                out = torch.full((cells,), float("nan"), device="cuda")
                generate_grid_sdf(dv, topo, grid, sign, out=out, algorithm=1, **kw)
                return out
            res["all pairs"] = with_mode({}, run_b)
        ref = res["default"].view(torch.int32)
        line = f"seed {seed}: {kind} {nt} triangles, grid {counts}, {sign.name}, {what}:"
        for name, r_ in res.items():
            same = bool(torch.equal(r_.view(torch.int32), ref))
            line += f" {name} {'=' if same else 'DIFFERS'}"
            bad += 0 if same else 1
        nq = int(10 ** rng.uniform(2, 5.7))
        q = torch.as_tensor((lo + rng.uniform(-0.2, 1.2, (nq, 3)) * (hi - lo)).astype(np.float32), device="cuda")
        am = [AccelerationMethod.RtreeBvh, AccelerationMethod.Rtree, AccelerationMethod.Bvh(SignMethod.Normal), AccelerationMethod.Bvh(SignMethod.Raycast)][seed % 4]
        qres = {name: with_mode(env, lambda: generate_sdf(dv, topo, q, am)) for name, env in MODES.items()}
        if nq * nt <= 3e10:
            qres["all pairs"] = with_mode({}, lambda: generate_sdf(dv, topo, q, am, algorithm=1))
        qref = qres["default"].view(torch.int32)
        line += f" | {nq} queries accel {am.kind}:"
        for name, r_ in qres.items():
            same = bool(torch.equal(r_.view(torch.int32), qref))
            line += f" {name} {'=' if same else 'DIFFERS'}"
            bad += 0 if same else 1
        print(line, flush=True)
        done += 1
    print(f"## {done} cases in {time.time() - t0:.0f} s, {bad} differences", flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
