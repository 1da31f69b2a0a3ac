#!/usr/bin/env python3
"""One rank's whole step of an N-GPU run, emulated alone on one GPU: the rank's contiguous x-slab through ONE
m2s_generate_grid_sdf call (build + sign planes + seeds + cut lists + walk) that also writes the slab into N-1 peer
buffers (here: other buffers on the same GPU, so the push costs HBM bandwidth instead of xGMI).  Wall time per step
vs the 1-GPU step gives the scaling the compute side allows:  speedup(N) <= t(1) / max_r t_rank(r).

    python tools/exp_rank_step.py [--world 8] [--grid 512] [--mesh blob-100k] [--modes push,store,none]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", default="8,4,2")
    ap.add_argument("--grid", type=int, default=512)
    ap.add_argument("--mesh", default="blob-100k")
    ap.add_argument("--sign", default="Raycast")
    ap.add_argument("--modes", default="none,trail,push,store")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--ranks", default="")
    ap.add_argument("--partition", default="contiguous", choices=["contiguous", "interleaved", "interleaved4", "adaptive"])
    ap.add_argument("--adapt-rounds", type=int, default=4, help="adaptive: rounds of (all ranks measured alone, slabs re-cut by m2s_balanced_slabs)")
    args = ap.parse_args()
    import torch

    from mesh_to_sdf_amd import Grid, M2STimings, PeerMode, SignMethod, Topology, balanced_slabs, generate_grid_sdf, interleaved_slab, meshes, slab_bounds

    v, idx = meshes.named(args.mesh)
    lo, hi = meshes.extended_bbox(v, 0.1)
    n = args.grid
    grid = Grid.from_bounding_box(lo, hi, [n, n, n])
    sign = SignMethod[args.sign]
    dv = torch.as_tensor(v, device="cuda")
    topo = Topology.TriangleList(torch.as_tensor(idx.astype(np.int64), device="cuda").to(torch.int32))
    out = torch.empty(n ** 3, dtype=torch.float32, device="cuda")

    def timed(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(args.iters):
            t0 = time.perf_counter()
            fn()
            ts.append((time.perf_counter() - t0) * 1e3)
        return float(np.median(ts)), float(np.min(ts))

    t = M2STimings()
    med1, min1 = timed(lambda: generate_grid_sdf(dv, topo, grid, sign, out=out, timings=t))
    print(f"# {args.mesh} {n}^3 {args.sign}: 1-GPU step wall median {med1:.3f} ms (min {min1:.3f}); build {t.accel_build_ms:.3f} sign {t.sign_ms:.3f} "
          f"seed+cut {t.seed_ms:.3f} distance {t.distance_ms:.3f} total {t.total_ms:.3f}")
    for world in [int(w) for w in args.world.split(",")]:
        peers = [torch.empty(n ** 3, dtype=torch.float32, device="cuda") for _ in range(world - 1)]
        ranks = [int(r) for r in args.ranks.split(",")] if args.ranks else (list(range(world)) if args.partition.startswith("interleaved") else sorted({0, world // 2 - 1, world // 2, world - 1}))
        if args.partition == "adaptive":
            # what M2S_PART_ADAPTIVE / the bench's adaptive slabs converge to: every round measures all ranks (alone, no delivery)
            # and re-cuts the slabs from their device times without the build
            bounds = [slab_bounds(n, world, r)[0] for r in range(world)] + [n]
            for rnd in range(args.adapt_rounds + 1):
                walls, costs = [], []
                for r in range(world):
                    t = M2STimings()
                    med, mn = timed(lambda: generate_grid_sdf(dv, topo, grid, sign, x_slab=(bounds[r], bounds[r + 1]), out=out, timings=t))
                    walls.append(med)
                    costs.append(max(t.total_ms - t.accel_build_ms, 0.0))
                print(f"world {world} adaptive round {rnd}: bounds {bounds} wall per rank {[round(w, 3) for w in walls]} -> slowest {max(walls):.3f} ms, "
                      f"bound {med1 / max(walls):.2f}x of {world}", flush=True)
                bounds = balanced_slabs(n, 4, bounds, costs)
            continue
        for mode in args.modes.split(","):
            worst = 0.0
            for r in ranks:
                xs, period = slab_bounds(n, world, r), 0
                if args.partition == "interleaved":
                    a, b, period = interleaved_slab(grid, world, r)
                    xs = (a, b)
                if args.partition == "interleaved4":       # four chunks per rank: chunk r of every quarter of the grid
                    c = n // (4 * world)
                    xs, period = (r * c, (r + 1) * c), world * c
                kw = {} if mode == "none" else {"peer_out": peers, "peer_mode": {"push": PeerMode.Push, "store": PeerMode.Store, "trail": PeerMode.Trail}[mode]}
                t = M2STimings()
                med, mn = timed(lambda: generate_grid_sdf(dv, topo, grid, sign, x_slab=xs, x_period=period, out=out, timings=t, **kw))
                worst = max(worst, med)
                print(f"world {world} rank {r} slab {xs}{' period ' + str(period) if period else ''} delivery={mode}: wall median {med:.3f} ms (min {mn:.3f}); build {t.accel_build_ms:.3f} sign {t.sign_ms:.3f} "
                      f"seed+cut {t.seed_ms:.3f} distance {t.distance_ms:.3f} ({t.distance_launches} launches) device total {t.total_ms:.3f}")
            print(f"## world {world} delivery={mode}: slowest rank {worst:.3f} ms -> compute-side speedup bound {med1 / worst:.2f}x of {world}")
        del peers


if __name__ == "__main__":
    main()
