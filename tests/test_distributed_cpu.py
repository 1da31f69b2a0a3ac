"""The N>1 path (x-slab shards + all-gather) under gloo, world_size 2 and 3, on CPU.  The slab
computation is injected (the oracle's exact semantics restricted to the slab) so that the
partition / gather logic of mesh_to_sdf_amd.distributed runs without a GPU."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle as orc
from mesh_to_sdf_amd import Grid, SignMethod, Topology, meshes
from mesh_to_sdf_amd.distributed import chunk_plan, generate_grid_sdf_sharded, piece_bounds, slab_bounds


def test_slab_bounds_cover_and_are_contiguous():
    for nx in (1, 5, 7, 64, 512, 513):
        for world in (1, 2, 3, 4, 8):
            prev = 0
            for r in range(world):
                x0, x1 = slab_bounds(nx, world, r)
                assert x0 == prev and x1 >= x0
                prev = x1
            assert prev == nx
            sizes = [slab_bounds(nx, world, r)[1] - slab_bounds(nx, world, r)[0] for r in range(world)]
            assert max(sizes) - min(sizes) <= 1
    assert [slab_bounds(512, 8, r) for r in (0, 7)] == [(0, 64), (448, 512)]


def test_chunk_plan_partitions_every_layer_once():
    for nx in (1, 5, 16, 15, 64, 512, 513, 1024):
        for world in (1, 2, 3, 8):
            for chunks in (1, 2, 4, 7):
                plan = chunk_plan(nx, world, chunks)
                assert plan[0][0] == 0 and plan[-1][1] == nx
                owner = [-1] * nx
                for (c0, c1), nxt in zip(plan, plan[1:] + [(nx, nx)]):
                    assert c1 == nxt[0] and c1 > c0
                    for r in range(world):
                        a, b = piece_bounds((c0, c1), world, r)
                        for x in range(a, b):
                            assert owner[x] == -1
                            owner[x] = r
                assert all(o >= 0 for o in owner)
                if world > 1 and nx >= chunks * world:
                    assert all((c1 - c0) % world == 0 for c0, c1 in plan[:-1])  # in-place all-gather applies


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, nx, chunks, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        v, idx = meshes.blob(24, 13)
        lo, hi = meshes.extended_bbox(v, 0.1)
        grid = Grid.from_bounding_box(lo, hi, [nx, 10, 12])
        first, size, cnt = grid.get_first_cell(), grid.get_cell_size(), grid.get_cell_count()
        whole = orc.generate_grid_sdf(v, idx, first, size, cnt, sign=0, semantics=orc.EXACT, threads=1)
        row = cnt[1] * cnt[2]

        full = torch.full((nx * row,), float("nan"))

        def compute_slab(out, x0, x1):   # writes ONLY this rank's piece, like the HIP slab call
            out[x0 * row : x1 * row] = torch.from_numpy(whole[x0 * row : x1 * row].copy())

        out = generate_grid_sdf_sharded(torch.from_numpy(v), Topology.TriangleList(idx), grid, SignMethod.Raycast,
                                        compute_slab=compute_slab, chunks=chunks, out=full)
        ok = np.array_equal(out.numpy().view(np.uint32), whole.view(np.uint32))
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,nx,chunks", [(2, 16, 1), (2, 16, 4), (2, 15, 2), (3, 16, 2)])
def test_sharded_grid_gloo(world, nx, chunks):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, nx, chunks, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res), res


def test_query_bounds_cover_all_queries():
    from mesh_to_sdf_amd.distributed import query_bounds

    for nq in (0, 1, 7, 64, 1000, 10_000_001):
        for world in (1, 2, 3, 8):
            prev = 0
            for r in range(world):
                q0, q1 = query_bounds(nq, world, r)
                assert q0 == prev and q1 >= q0
                prev = q1
            assert prev == nq


def _query_worker(rank, world, port, nq, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mesh_to_sdf_amd import AccelerationMethod
        from mesh_to_sdf_amd.distributed import generate_sdf_sharded

        v, idx = meshes.blob(24, 13)
        lo, hi = meshes.extended_bbox(v, 0.1)
        pts = meshes.uniform_queries(lo, hi, nq)
        whole = orc.generate_sdf(v, idx, pts, accel=3, fast=True) if nq else np.zeros(0, np.float32)

        def compute_range(out, q0, q1):   # writes ONLY this rank's queries, like the HIP call on the sub-array
            out[q0:q1] = torch.from_numpy(whole[q0:q1].copy())

        out = generate_sdf_sharded(torch.from_numpy(v), Topology.TriangleList(idx), torch.from_numpy(pts), AccelerationMethod.RtreeBvh,
                                   compute_range=compute_range)
        q.put((rank, bool(np.array_equal(out.numpy().view(np.uint32), whole.view(np.uint32)))))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,nq", [(2, 1000), (2, 1001), (3, 10), (3, 2), (2, 0)])
def test_sharded_queries_gloo(world, nq):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_query_worker, args=(r, world, port, nq, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res), res
