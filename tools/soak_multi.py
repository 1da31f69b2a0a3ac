#!/usr/bin/env python3
"""Soak of the multi-shard entry point on ONE GPU (devices = [0] * n): random meshes, grids, shard counts, partitions, exchanges and peer
modes, device-resident and host results — every buffer must equal the single call bit for bit (with M2S_XCHG_NONE: on the shard's own
slab, as m2s_multi_opts.slabs reports it).   python tools/soak_multi.py [--seeds 200] [--first 0] [--seconds 600]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mesh_to_sdf_amd import Exchange, Grid, M2SError, Partition, PeerMode, SignMethod, Topology, generate_grid_sdf, generate_grid_sdf_multi, meshes


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=200)
    ap.add_argument("--first", type=int, default=0)
    ap.add_argument("--seconds", type=float, default=600)
    a = ap.parse_args()
    t0, bad, done, refused = time.time(), 0, 0, 0
    for seed in range(a.first, a.first + a.seeds):
        if time.time() - t0 > a.seconds:
            break
        rng = np.random.default_rng(91000 + seed)
        kind = rng.choice(["blob", "sheet"])
        nt_target = int(10 ** rng.uniform(2.3, 5.2))
        su = max(4, int(np.sqrt(nt_target / 2 * 1.25)))
        sv = max(3, nt_target // (2 * su) + 1)
        v, idx = (meshes.sheet(su + 1, sv + 1) if kind == "sheet" else meshes.blob(su, sv))
        lo, hi = meshes.extended_bbox(v, float(rng.uniform(0.0, 0.3)))
        if seed % 3 == 0:
            counts = [int(rng.choice([32, 64, 96, 128, 160, 256])), int(rng.integers(8, 130)), int(rng.integers(8, 130))]
        else:
            counts = [int(c) for c in np.clip(10 ** rng.uniform(0.9, 2.25, 3), 8, 180)]
        sign = SignMethod.Raycast if rng.random() < 0.6 else SignMethod.Normal
        world = int(rng.integers(1, 9))
        part = Partition(int(rng.integers(0, 4)))
        if part in (Partition.Interleaved, Partition.Auto) and rng.random() < 0.8:      # a shape the interleaved chunks accept: 2 n chunks of 16 m layers
            counts[0] = world * 2 * 16 * int(rng.integers(1, 3))
            counts[1], counts[2] = int(rng.integers(8, 70)), int(rng.integers(8, 70))
        grid = Grid.from_bounding_box(lo, hi, counts)
        xchg = [Exchange.Auto, Exchange.Peer, Exchange.Nothing][int(rng.integers(0, 3))]
        pmode = PeerMode(int(rng.integers(0, 3)))
        host = seed % 4 == 1
        cells = counts[0] * counts[1] * counts[2]
        nt = len(idx) // 3
        line = f"seed {seed}: {kind} {nt} triangles, grid {counts}, {sign.name}, {world} shards, {part.name}, {xchg.name}, {pmode.name}, {'host' if host else 'device'}:"
        try:
            if host:
                want = generate_grid_sdf(v, Topology.TriangleList(idx), grid, sign)
                info = {}
                got = generate_grid_sdf_multi(v, Topology.TriangleList(idx), grid, sign, devices=[0] * world, partition=part, peer_mode=pmode, info=info)
                same = bool(np.array_equal(got.view(np.uint32), want.view(np.uint32)))
                line += f" {'=' if same else 'DIFFERS'} ({info.get('partition')})"
                bad += 0 if same else 1
            else:
                dv = torch.as_tensor(v, device="cuda")
                topo = Topology.TriangleList(torch.as_tensor(idx.astype(np.int64), device="cuda").to(torch.int32))
                want = generate_grid_sdf(dv, topo, grid, sign)
                outs = [torch.full((cells,), float("nan"), device="cuda") for _ in range(world)]
                info = {}
                generate_grid_sdf_multi(dv, topo, grid, sign, devices=[0] * world, outs=outs, exchange=xchg, partition=part, peer_mode=pmode, info=info)
                torch.cuda.synchronize()
                ok = True
                row = counts[1] * counts[2]
                for k, o in enumerate(outs):
                    if xchg == Exchange.Nothing and world > 1:
                        x0, x1, period = info["slabs"][k]
                        mask = torch.zeros(counts[0], dtype=torch.bool, device="cuda")
                        if period:
                            c = x1 - x0
                            for s in range(x0, counts[0], period):
                                mask[s:s + c] = True
                        else:
                            mask[x0:x1] = True
                        sel = mask.repeat_interleave(row)
                        ok &= bool(torch.equal(o.view(torch.int32)[sel], want.view(torch.int32)[sel]))
                    else:
                        ok &= bool(torch.equal(o.view(torch.int32), want.view(torch.int32)))
                line += f" {'=' if ok else 'DIFFERS'} ({info.get('partition')}, {info.get('exchange')})"
                bad += 0 if ok else 1
        except M2SError as e:
            line += f" refused: {e}"
            refused += 1
        print(line, flush=True)
        done += 1
    print(f"## {done} cases in {time.time() - t0:.0f} s, {bad} differences, {refused} refused", flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
