"""Multi-GPU generate_grid_sdf behind the C ABI (include/m2s.h: m2s_generate_grid_sdf_multi, m2s_opts.peer_out,
m2s_ipc_*), exercised on ONE GPU: `devices` may name a device twice (two shards, two contexts, two host threads),
peers are then other buffers on the same device; the cross-process exchange runs as two processes sharing the GPU;
RCCL runs as a 1-rank communicator.  A device-count-gated test covers real peers where the box has them.
Everything is compared bit for bit with the single-device call (itself bit-identical to the oracle, test_gpu_parity.py).
"""
import os
import subprocess
import sys

import numpy as np
import pytest

from mesh_to_sdf_amd import _lib
from mesh_to_sdf_amd import (AccelerationMethod, Exchange, Grid, Partition, PeerMode, SignMethod, Topology, generate_grid_sdf,
                             generate_grid_sdf_multi, generate_sdf, generate_sdf_multi, interleaved_slab, meshes, slab_bounds)

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _case(n=(70, 40, 36), mesh="blob-6k"):
    v, idx = meshes.named(mesh)
    lo, hi = meshes.extended_bbox(v, 0.1)
    return v, idx, Grid.from_bounding_box(lo, hi, list(n))


def _device_inputs(v, idx):
    import torch

    return torch.as_tensor(v, device="cuda:0"), torch.as_tensor(idx.astype(np.int64), device="cuda:0")


@pytest.mark.parametrize("sign", [SignMethod.Raycast, SignMethod.Normal])
@pytest.mark.parametrize("devices", [[0], [0, 0], [0, 0, 0]])
def test_multi_host_result_equals_single_call(sign, devices):
    v, idx, g = _case()
    want = generate_grid_sdf(v, Topology.TriangleList(idx), g, sign)
    info = {}
    got = generate_grid_sdf_multi(v, Topology.TriangleList(idx), g, sign, devices=devices, info=info)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert info["exchange"] == "Nothing"                       # host result: no collective at all
    nx = g.get_cell_count()[0]
    assert sum(int(t.n_units) for t in info["timings"]) == g.get_total_cell_count()
    assert [int(t.n_units) for t in info["timings"]] == [(b - a) * 40 * 36 for a, b in (slab_bounds(nx, len(devices), k) for k in range(len(devices)))]


@pytest.mark.parametrize("mode", [PeerMode.Push, PeerMode.Store, PeerMode.Trail])
@pytest.mark.parametrize("sign", [SignMethod.Raycast, SignMethod.Normal])
def test_multi_device_resident_peer_exchange_two_shards_one_gpu(mode, sign):
    import torch

    v, idx, g = _case((83, 40, 36))   # uneven slabs, x not a multiple of the brick
    dv, di = _device_inputs(v, idx)
    want = generate_grid_sdf(dv, Topology.TriangleList(di), g, sign)
    outs = [torch.full_like(want, float("nan")) for _ in range(3)]
    info = {}
    got = generate_grid_sdf_multi(dv, Topology.TriangleList(di), g, sign, devices=[0, 0, 0], outs=outs, exchange=Exchange.Peer,
                                  peer_mode=mode, info=info)
    assert info["exchange"] == "Peer"
    for k, o in enumerate(got):     # EVERY buffer holds the whole grid
        assert torch.equal(o.view(torch.int32), want.view(torch.int32)), f"buffer {k}"


@pytest.mark.parametrize("sign", [SignMethod.Raycast, SignMethod.Normal])
def test_multi_peer_trailing_push_large_slab(sign):
    """M2S_PEER_TRAIL on slabs large enough for cut lists and many progress units, with an x extent that is not a multiple of
    the unit (ragged last unit) and y/z extents that are not multiples of the brick."""
    import torch

    v, idx = meshes.named("blob-100k")
    lo, hi = meshes.extended_bbox(v, 0.1)
    g = Grid.from_bounding_box(lo, hi, [270, 250, 254])
    dv, di = _device_inputs(v, idx)
    want = generate_grid_sdf(dv, Topology.TriangleList(di), g, sign)
    info = {}
    outs = generate_grid_sdf_multi(dv, Topology.TriangleList(di), g, sign, devices=[0, 0], exchange=Exchange.Peer, peer_mode=PeerMode.Trail, info=info)
    assert [int(t.distance_launches) for t in info["timings"]] == [1, 1]
    for o in outs:
        assert torch.equal(o.view(torch.int32), want.view(torch.int32))


@pytest.mark.parametrize("mode", [PeerMode.Push, PeerMode.Store, PeerMode.Trail])
@pytest.mark.parametrize("sign", [SignMethod.Raycast, SignMethod.Normal])
def test_multi_interleaved_partition(mode, sign):
    """m2s_partition INTERLEAVED: shard k computes the chunks k and n + k of 2n (m2s_opts.x_period) in one call; every
    delivery mode; cut lists, seeds and walk are numbered along the virtual slab, addresses and geometry along the grid."""
    import torch

    v, idx = meshes.named("blob-100k")
    lo, hi = meshes.extended_bbox(v, 0.1)
    g = Grid.from_bounding_box(lo, hi, [256, 250, 254])
    assert interleaved_slab(g, 4, 1) == (32, 64, 128) and interleaved_slab(g, 3, 1)[2] == 0
    dv, di = _device_inputs(v, idx)
    want = generate_grid_sdf(dv, Topology.TriangleList(di), g, sign)
    info = {}
    outs = generate_grid_sdf_multi(dv, Topology.TriangleList(di), g, sign, devices=[0, 0, 0, 0], exchange=Exchange.Peer, peer_mode=mode,
                                   partition=Partition.Interleaved, info=info)
    for o in outs:
        assert torch.equal(o.view(torch.int32), want.view(torch.int32))
    assert sum(int(t.n_units) for t in info["timings"]) == g.get_total_cell_count()


@pytest.mark.parametrize("case", range(8))
def test_interleaved_calls_fuzz(case):
    """Random meshes / grids (anisotropic and negative cell sizes, ragged y and z, non-cubic packet bricks) cut into interleaved
    chunks, one call per shard, with the cut lists forced on: the union must equal the single call bit for bit."""
    import torch

    rng = np.random.default_rng(7000 + case)
    nt = int(rng.integers(50, 3000))
    centers = rng.uniform(-1, 1, (nt, 1, 3))
    tri = (centers + rng.standard_normal((nt, 3, 3)) * 10.0 ** rng.uniform(-2.5, -0.5, (nt, 1, 1))).astype(np.float32)
    v, idx = tri.reshape(-1, 3), np.arange(3 * nt, dtype=np.uint32)
    world = int(rng.choice([2, 4]))
    nx = int(rng.choice([64, 128, 256]))
    counts = [nx, int(rng.integers(5, 60)), int(rng.integers(5, 60))]
    lo, hi = v.min(0) - rng.uniform(0, 0.5, 3).astype(np.float32), v.max(0) + rng.uniform(0, 0.5, 3).astype(np.float32)
    if case % 3 == 1:
        lo[0], hi[0] = hi[0], lo[0]           # negative cell size along x
    if case % 4 == 2:
        hi[0] = lo[0] + (hi[0] - lo[0]) * 8   # anisotropic: long cells along x, so packet bricks thin along x
    g = Grid.from_bounding_box(lo, hi, counts)
    with _lib.knobs(M2S_CUT_MIN_PACKETS=8, M2S_CUT_COARSE=case % 2):   # odd cases: two-level cut lists (blocks inside the chunks of an interleaved slab)
        dv, di = torch.as_tensor(v, device="cuda:0"), torch.as_tensor(idx.astype(np.int64), device="cuda:0")
        for sign in (SignMethod.Raycast, SignMethod.Normal):
            want = generate_grid_sdf(dv, Topology.TriangleList(di), g, sign)
            out = torch.full_like(want, float("nan"))
            used = 0
            for k in range(world):
                a, b, period = interleaved_slab(g, world, k)
                used += period != 0
                generate_grid_sdf(dv, Topology.TriangleList(di), g, sign, x_slab=(a, b), x_period=period, out=out)
            assert torch.equal(out.view(torch.int32), want.view(torch.int32)), (case, sign, counts, world, used)


def test_multi_interleaved_where_the_grid_does_not_allow_it():
    from mesh_to_sdf_amd import M2SPanic

    v, idx, g = _case((70, 40, 36))
    dv, di = _device_inputs(v, idx)
    with pytest.raises(M2SPanic):
        generate_grid_sdf_multi(dv, Topology.TriangleList(di), g, SignMethod.Raycast, devices=[0, 0], exchange=Exchange.Peer, partition=Partition.Interleaved)
    with pytest.raises(M2SPanic):   # host results stream out slab by slab: no interleaving
        generate_grid_sdf(v, Topology.TriangleList(idx), Grid.from_bounding_box([-1, -1, -1], [1, 1, 1], [64, 64, 64]), SignMethod.Raycast, x_slab=(0, 16), x_period=32)
    with pytest.raises(M2SPanic):   # chunk not a power of two
        generate_grid_sdf(dv, Topology.TriangleList(di), Grid.from_bounding_box([-1, -1, -1], [1, 1, 1], [96, 64, 64]), SignMethod.Raycast, x_slab=(0, 24), x_period=48)


def test_one_brick_chunks_are_refused_and_auto_partition_stays_correct():
    """with 16-layer packet bricks (cell_size 0.125 along x) a 16-layer chunk is ONE brick; the peer push used to round
    its piece up to two bricks and walk two chunks as one range.  Explicit x_period with such a chunk is now M2S_ERR_BAD_ARG, and the
    multi call's AUTO partition falls back to contiguous slabs there: every buffer equals the single call."""
    import torch

    from mesh_to_sdf_amd import M2SPanic

    v, idx = meshes.blob(48, 25)
    g = Grid.new([-2.0, -1.6, -1.6], [4.0 / 128, 3.2 / 16, 3.2 / 16], [128, 16, 16])
    dv, di = _device_inputs(v, idx)
    want = generate_grid_sdf(dv, Topology.TriangleList(di), g, SignMethod.Raycast)
    with pytest.raises(M2SPanic):
        generate_grid_sdf(dv, Topology.TriangleList(di), g, SignMethod.Raycast, x_slab=(16, 32), x_period=64)
    for mode in (PeerMode.Push, PeerMode.Store):
        outs = generate_grid_sdf_multi(dv, Topology.TriangleList(di), g, SignMethod.Raycast, devices=[0, 0, 0, 0], exchange=Exchange.Peer, peer_mode=mode)
        for o in outs:
            assert torch.equal(o.view(torch.int32), want.view(torch.int32))


def test_adaptive_partition_and_persistent_workers():
    """M2S_PART_ADAPTIVE: contiguous slabs re-cut after every call from the per-shard times of the call before (m2s_balanced_slabs);
    whatever the boundaries, every buffer equals the single call.  The shard threads are the library's persistent workers: many
    calls in a row, shard counts going up and down, and two host threads calling at once (the second falls back to its own threads)."""
    import threading

    import torch

    v, idx = meshes.named("blob-100k")
    lo, hi = meshes.extended_bbox(v, 0.1)
    g = Grid.from_bounding_box(lo, hi, [256, 64, 64])
    dv, di = _device_inputs(v, idx)
    want = generate_grid_sdf(dv, Topology.TriangleList(di), g, SignMethod.Raycast)
    seen = set()
    for it in range(5):
        info = {}
        outs = generate_grid_sdf_multi(dv, Topology.TriangleList(di), g, SignMethod.Raycast, devices=[0, 0, 0, 0], exchange=Exchange.Peer,
                                       partition=Partition.Adaptive, info=info)
        for o in outs:
            assert torch.equal(o.view(torch.int32), want.view(torch.int32))
        assert info["partition"] == "Adaptive"
        b = [s[0] for s in info["slabs"]] + [info["slabs"][-1][1]]
        assert b[0] == 0 and b[-1] == 256 and all(b[k + 1] > b[k] and b[k] % 32 == 0 for k in range(4)) and all(s[2] == 0 for s in info["slabs"])
        seen.add(tuple(b))
    assert (0, 64, 128, 192, 256) in seen                                   # the first call knows nothing: even slabs
    for n in (2, 4, 3, 1, 4):                                                # the worker pool grows and idles
        info = {}
        outs = generate_grid_sdf_multi(dv, Topology.TriangleList(di), g, SignMethod.Raycast, devices=[0] * n, exchange=Exchange.Peer, info=info)
        assert all(torch.equal(o.view(torch.int32), want.view(torch.int32)) for o in outs)
        assert info["partition"] in ("Interleaved", "Contiguous")
    errs = []

    def call():
        try:
            o = generate_grid_sdf_multi(dv, Topology.TriangleList(di), g, SignMethod.Raycast, devices=[0, 0], exchange=Exchange.Peer)
            assert all(torch.equal(x.view(torch.int32), want.view(torch.int32)) for x in o)
        except Exception as e:   # noqa: BLE001
            errs.append(e)

    th = [threading.Thread(target=call) for _ in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs


def test_multi_xchg_none_keeps_contiguous_slabs():
    """M2S_XCHG_NONE leaves buffer k with shard k's cells only, and a caller finds them through m2s_slab_bounds: AUTO must not
    interleave there."""
    import torch

    v, idx = meshes.named("blob-100k")
    lo, hi = meshes.extended_bbox(v, 0.1)
    g = Grid.from_bounding_box(lo, hi, [128, 64, 64])
    dv, di = _device_inputs(v, idx)
    want = generate_grid_sdf(dv, Topology.TriangleList(di), g, SignMethod.Raycast).view(torch.int32)
    info = {}
    outs = [torch.zeros(128 * 64 * 64, dtype=torch.float32, device="cuda:0") for _ in range(4)]
    generate_grid_sdf_multi(dv, Topology.TriangleList(di), g, SignMethod.Raycast, devices=[0, 0, 0, 0], outs=outs, exchange=Exchange.Nothing, info=info)
    assert info["partition"] == "Contiguous"
    row = 64 * 64
    for k, o in enumerate(outs):
        a, b = slab_bounds(128, 4, k)
        assert torch.equal(o.view(torch.int32)[a * row:b * row], want[a * row:b * row])


def test_multi_peer_push_large_slab_in_pieces():
    """A slab big enough for the piece pipeline (walk of piece i+1 beside the push of piece i)."""
    import torch

    v, idx = meshes.named("blob-100k")
    lo, hi = meshes.extended_bbox(v, 0.1)
    g = Grid.from_bounding_box(lo, hi, [256, 256, 256])
    dv, di = _device_inputs(v, idx)
    want = generate_grid_sdf(dv, Topology.TriangleList(di), g, SignMethod.Raycast)
    info = {}
    outs = generate_grid_sdf_multi(dv, Topology.TriangleList(di), g, SignMethod.Raycast, devices=[0, 0], exchange=Exchange.Peer, info=info,
                                   partition=Partition.Contiguous)
    assert [int(t.distance_launches) for t in info["timings"]] == [4, 4]
    for o in outs:
        assert torch.equal(o.view(torch.int32), want.view(torch.int32))


def test_multi_rccl_one_rank():
    """The ncclAllGather fallback through librccl (dlopen), as a 1-rank communicator on this box's GPU."""
    import torch

    v, idx, g = _case()
    dv, di = _device_inputs(v, idx)
    want = generate_grid_sdf(dv, Topology.TriangleList(di), g, SignMethod.Raycast)
    info = {}
    outs = generate_grid_sdf_multi(dv, Topology.TriangleList(di), g, SignMethod.Raycast, devices=[0], exchange=Exchange.Rccl, info=info)
    assert info["exchange"] == "Rccl"
    assert torch.equal(outs[0].view(torch.int32), want.view(torch.int32))


def test_multi_exchange_none_leaves_other_slabs_alone():
    import torch

    v, idx, g = _case()
    dv, di = _device_inputs(v, idx)
    want = generate_grid_sdf(dv, Topology.TriangleList(di), g, SignMethod.Raycast).view(70, -1)
    outs = [torch.full((70 * 40 * 36,), -7.0, device="cuda:0") for _ in range(2)]
    generate_grid_sdf_multi(dv, Topology.TriangleList(di), g, SignMethod.Raycast, devices=[0, 0], outs=outs, exchange=Exchange.Nothing,
                            partition=Partition.Contiguous)
    for k in range(2):
        a, b = slab_bounds(70, 2, k)
        o = outs[k].view(70, -1)
        assert torch.equal(o[a:b], want[a:b])
        mask = torch.ones(70, dtype=torch.bool, device="cuda:0")
        mask[a:b] = False
        assert bool((o[mask] == -7.0).all())


# ---- generate_sdf over several shards (m2s_generate_sdf_multi) ------------------------------------------------------
QUERY_METHODS = [AccelerationMethod.RtreeBvh, AccelerationMethod.Rtree, AccelerationMethod.Bvh(SignMethod.Normal), AccelerationMethod.None_(SignMethod.Raycast)]


@pytest.mark.parametrize("devices", [[0], [0, 0], [0, 0, 0]])
def test_queries_multi_host_result_equals_single_call(devices):
    v, idx = meshes.named("blob-6k")
    q = meshes.uniform_queries(*meshes.extended_bbox(v, 0.1), 50_001)      # does not divide
    for am in QUERY_METHODS:
        want = generate_sdf(v, Topology.TriangleList(idx), q, am)
        info = {}
        got = generate_sdf_multi(v, Topology.TriangleList(idx), q, am, devices=devices, info=info)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), am
        assert info["exchange"] == "Nothing"
        assert [int(t.n_units) for t in info["timings"]] == [b - a for a, b in (slab_bounds(len(q), len(devices), k) for k in range(len(devices)))]


@pytest.mark.parametrize("exchange", [Exchange.Peer, Exchange.Nothing])
def test_queries_multi_device_resident_three_shards_one_gpu(exchange):
    import torch

    v, idx = meshes.named("blob-6k")
    q = meshes.uniform_queries(*meshes.extended_bbox(v, 0.1), 40_003)
    dv, di = _device_inputs(v, idx)
    dq = torch.as_tensor(q, device="cuda:0")
    want = generate_sdf(dv, Topology.TriangleList(di), dq, AccelerationMethod.RtreeBvh)
    outs = [torch.full_like(want, -7.0) for _ in range(3)]
    info = {}
    got = generate_sdf_multi(dv, Topology.TriangleList(di), dq, AccelerationMethod.RtreeBvh, devices=[0, 0, 0], outs=outs, exchange=exchange, info=info)
    assert info["exchange"] == exchange.name
    for k, o in enumerate(got):
        a, b = slab_bounds(len(q), 3, k)
        if exchange == Exchange.Peer:          # EVERY buffer holds all distances
            assert torch.equal(o.view(torch.int32), want.view(torch.int32)), f"buffer {k}"
        else:                                  # buffer k holds range k only
            assert torch.equal(o[a:b].view(torch.int32), want[a:b].view(torch.int32))
            mask = torch.ones(len(q), dtype=torch.bool, device="cuda:0")
            mask[a:b] = False
            assert bool((o[mask] == -7.0).all())


def test_queries_multi_rccl_one_rank_and_edge_cases():
    import torch

    from mesh_to_sdf_amd import M2SPanic

    v, idx = meshes.named("blob-6k")
    q = meshes.uniform_queries(*meshes.extended_bbox(v, 0.1), 10_000)
    dv, di = _device_inputs(v, idx)
    dq = torch.as_tensor(q, device="cuda:0")
    want = generate_sdf(dv, Topology.TriangleList(di), dq, AccelerationMethod.RtreeBvh)
    info = {}
    outs = generate_sdf_multi(dv, Topology.TriangleList(di), dq, AccelerationMethod.RtreeBvh, devices=[0], exchange=Exchange.Rccl, info=info)
    assert info["exchange"] == "Rccl" and torch.equal(outs[0].view(torch.int32), want.view(torch.int32))
    # more shards than queries; no queries at all; RtreeBvh on a mesh without triangles returns nothing (rtree_bvh.rs:104-106)
    few = generate_sdf_multi(v, Topology.TriangleList(idx), q[:2], AccelerationMethod.RtreeBvh, devices=[0, 0, 0])
    assert np.array_equal(few.view(np.uint32), generate_sdf(v, Topology.TriangleList(idx), q[:2], AccelerationMethod.RtreeBvh).view(np.uint32))
    assert generate_sdf_multi(v, Topology.TriangleList(idx), q[:0], AccelerationMethod.RtreeBvh, devices=[0, 0]).size == 0
    assert generate_sdf_multi(v, Topology.TriangleList(idx[:0]), q[:100], AccelerationMethod.RtreeBvh, devices=[0, 0]).size == 0
    with pytest.raises(M2SPanic):
        generate_sdf_multi(v, Topology.TriangleList(idx), q, AccelerationMethod.RtreeBvh, devices=[0, 99])


def test_multi_real_peers_when_the_box_has_them():
    import torch

    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs (the driver's GPU box has one)")
    n = min(torch.cuda.device_count(), 8)
    v, idx, g = _case((96, 64, 64), "blob-100k")
    dv, di = _device_inputs(v, idx)
    want = generate_grid_sdf(dv, Topology.TriangleList(di), g, SignMethod.Raycast)
    for exchange, mode in ((Exchange.Peer, PeerMode.Push), (Exchange.Peer, PeerMode.Store), (Exchange.Rccl, PeerMode.Push)):
        outs = generate_grid_sdf_multi(dv, Topology.TriangleList(di), g, SignMethod.Raycast, devices=list(range(n)), exchange=exchange, peer_mode=mode)
        for o in outs:
            assert torch.equal(o.cpu().view(torch.int32), want.cpu().view(torch.int32)), (exchange, mode)
    host = generate_grid_sdf_multi(v, Topology.TriangleList(idx), g, SignMethod.Raycast, devices=list(range(n)))
    assert np.array_equal(host.view(np.uint32), want.cpu().numpy().view(np.uint32))
    q = meshes.uniform_queries(*meshes.extended_bbox(v, 0.1), 300_007)
    dq = torch.as_tensor(q, device="cuda:0")
    wq = generate_sdf(dv, Topology.TriangleList(di), dq, AccelerationMethod.RtreeBvh)
    for exchange in (Exchange.Peer, Exchange.Rccl):
        for o in generate_sdf_multi(dv, Topology.TriangleList(di), dq, AccelerationMethod.RtreeBvh, devices=list(range(n)), exchange=exchange):
            assert torch.equal(o.cpu().view(torch.int32), wq.cpu().view(torch.int32)), exchange


def test_multi_errors():
    from mesh_to_sdf_amd import M2SPanic

    v, idx, g = _case()
    with pytest.raises(M2SPanic):
        generate_grid_sdf_multi(v, Topology.TriangleList(idx), g, SignMethod.Raycast, devices=[0, 99])
    bad = idx.copy()
    bad[5] = 10 ** 6                                              # out-of-range index: the reference panics; every shard reports it
    with pytest.raises(M2SPanic):
        generate_grid_sdf_multi(v, Topology.TriangleList(bad), g, SignMethod.Raycast, devices=[0, 0])
    empty = Grid.from_bounding_box([0, 0, 0], [1, 1, 1], [0, 4, 4])
    assert generate_grid_sdf_multi(v, Topology.TriangleList(idx), empty, SignMethod.Raycast, devices=[0, 0]).size == 0


def test_concurrent_calls_on_two_lanes_from_two_threads():
    """Threading contract of include/m2s.h: contexts are per (device, lane); two host threads on different lanes run
    concurrently and do not disturb each other."""
    import threading

    v, idx, g = _case()
    want = {s: generate_grid_sdf(v, Topology.TriangleList(idx), g, s) for s in (SignMethod.Raycast, SignMethod.Normal)}
    errs = []

    def work(lane, sign):
        try:
            for _ in range(6):
                got = generate_grid_sdf(v, Topology.TriangleList(idx), g, sign, lane=lane)
                assert np.array_equal(got.view(np.uint32), want[sign].view(np.uint32))
        except Exception as e:   # noqa: BLE001
            errs.append(e)

    th = [threading.Thread(target=work, args=(k, s)) for k, s in enumerate((SignMethod.Raycast, SignMethod.Normal, SignMethod.Raycast))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs


def _run_worker(mode, world, extra_env=None, timeout=300):
    env = dict(os.environ)
    env.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(29500 + (os.getpid() % 400)), "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    env.update(extra_env or {})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", env["MASTER_PORT"], os.path.join(ROOT, "tests", "dist_worker.py"), mode]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-3000:] + "\n" + r.stderr[-3000:]
    return r.stdout


def test_two_processes_one_gpu_ipc_peer_exchange():
    """One process per GPU, peers mapped through m2s_ipc_*: two ranks share this box's GPU, each writes its slab into
    the other's whole-grid buffer (push and epilogue-store modes), host barrier, both compare with the single call."""
    out = _run_worker("ipc_peer", 2, {"M2S_DIST_BACKEND": "gloo"})
    assert out.count("ipc_peer ok") == 2, out


def test_nccl_backend_one_rank():
    """The torch.distributed "nccl" backend (= RCCL) on this box's GPU: 1-rank group, chunked in-place all-gathers
    issued from the piece streams (M2S_FORCE_COLLECTIVES), result equal to the single call."""
    out = _run_worker("nccl_one_rank", 1, {"M2S_FORCE_COLLECTIVES": "1"})
    assert "nccl_one_rank ok" in out, out


def _bench(args, env=None, launcher=None):
    e = dict(os.environ)
    e.update(env or {})
    cmd = (launcher or [sys.executable]) + [os.path.join(ROOT, "bench.py")] + args
    r = subprocess.run(cmd, env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, f"stdout must hold exactly one line (the JSON): {lines[:5]}"    # RCCL's banner and the like belong on stderr
    import json

    return json.loads(lines[0])


def test_bench_contract_single_line_json_in_every_mode():
    """bench.py prints ONE JSON line on stdout whatever runs underneath (RCCL prints a banner on stdout when a communicator is
    created), carries `roofline`, and the multi-shard modes verify the delivered grid against the single-GPU result."""
    small = ["--grid", "128", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    d = _bench(small)
    assert d["n_gpus"] == 1 and d["unit"] == "Mvoxels/s" and d["roofline"]["bound"] == "hbm" and d["roofline"]["frac"] > 0
    d = _bench(small + ["--gpus", "3"], {"M2S_BENCH_DEVICES": "0,0,0"})                       # one process, three shards on this GPU
    assert d["n_gpus"] == 3 and d["config"]["gather_verified"] is True and d["config"]["exchange"] == "in-process"
    assert len(d["per_rank"]) == 3 and all(r["walk_ms"] > 0 and r["accel_build_ms"] > 0 and len(r["slab"]) == 3 for r in d["per_rank"])
    assert d["config"]["exchange_ran"] == "Peer"
    launch = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
              "--master-port", str(29700 + os.getpid() % 200)]
    for exchange in ("peer", "rccl"):                                                           # two ranks share this GPU (gloo: RCCL refuses that)
        d = _bench(small + ["--gpus", "2"], {"M2S_DIST_BACKEND": "gloo", "M2S_EXCHANGE": exchange}, launch)
        assert d["n_gpus"] == 2 and d["config"]["exchange"] == exchange and d["config"]["gather_verified"] is True
        assert [r["rank"] for r in d["per_rank"]] == [0, 1] and all(r["walk_ms"] > 0 and r["step_wall_ms"] > 0 for r in d["per_rank"])
        if exchange == "peer":                                                                  # the link probe ran on both ranks (here: into the same GPU's HBM)
            assert [p["rank"] for p in d["link_probe"]] == [0, 1] and all(p["gbps_per_peer"][0] > 1.0 for p in d["link_probe"])
    d = _bench(small, {"M2S_FORCE_COLLECTIVES": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(29900 + os.getpid() % 90), "RANK": "0",
                       "WORLD_SIZE": "1", "M2S_EXCHANGE": "rccl"})                              # 1-rank nccl group: the banner case
    assert d["config"]["exchange"] == "rccl" and d["config"]["gather_verified"] is True
