"""Multi-GPU generate_grid_sdf: one process per GPU, x-slab shards, RCCL all-gather over xGMI.

Every voxel depends only on the (replicated) mesh, so the grid shards with no data-path
exchange.  The reference's output layout is x-slowest (`idx = z + y*nz + x*ny*nz`, grid.rs:122-124),
so a slab of cells along cell-axis 0 is one CONTIGUOUS range of the output and the gather is an
in-place `all_gather_into_tensor` (backend "nccl" == RCCL on ROCm).  Each rank marks the sign planes
for the whole grid (hits are slab independent: t is measured from cell 0 of the line,
generate/grid.rs:570-617) — O(T) work, no halo, no exchange.

Overlap: the grid is cut into `chunks` contiguous x-ranges; inside chunk c rank r owns the r-th
sub-slab, so the all-gather of chunk c is an in-place collective on ONE contiguous range and runs
(asynchronously, on RCCL's stream) while the ranks already compute chunk c+1.  xGMI is
point-to-point, so the gather is per-link bound; hiding it under compute is what keeps strong
scaling alive once the kernel is fast.  The mesh (LBVH + sign planes) is built once per call.

A rank's pieces are thin (512 / 8 ranks / 4 chunks = 16 layers): alone on the chip each would end in a
partly filled tail, and its small seeding kernels would leave most CUs idle.  Consecutive pieces are
therefore enqueued on two alternating side streams (the library keeps one scratch block per stream), so
piece c+1 fills the CUs that piece c's tail leaves free; the gather of chunk c is ordered after the
stream that computed it.
"""
from typing import Callable, List, Optional, Tuple

from .api import (AccelerationMethod, Grid, Mesh, PeerMode, SharedGrid, SignMethod, Topology, generate_grid_sdf,
                  generate_sdf, interleaved_slab)


import os

# Test hook: run the collectives even in a 1-rank group (exercises the RCCL calls on a 1-GPU box).
_FORCE_COLLECTIVES = os.environ.get("M2S_FORCE_COLLECTIVES", "0") == "1"


def slab_bounds(nx: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous x-slab [x0, x1) of rank `rank` out of `world`; sizes differ by at most one layer."""
    base, rem = divmod(nx, world)
    x0 = rank * base + min(rank, rem)
    return x0, x0 + base + (1 if rank < rem else 0)


def chunk_plan(nx: int, world: int, chunks: int) -> List[Tuple[int, int]]:
    """Contiguous x-ranges [c0, c1) that are gathered one after the other.  Every chunk is a multiple
    of `world` layers (so the in-place all-gather applies) except possibly the last one."""
    chunks = max(1, min(chunks, max(1, nx // max(world, 1))))
    per = (nx // chunks) // world * world if nx >= chunks * world else nx
    if per == 0:
        return [(0, nx)]
    plan, x = [], 0
    for c in range(chunks):
        x1 = nx if c == chunks - 1 else x + per
        plan.append((x, x1))
        x = x1
    return [p for p in plan if p[1] > p[0]]


def piece_bounds(chunk: Tuple[int, int], world: int, rank: int) -> Tuple[int, int]:
    c0, c1 = chunk
    a, b = slab_bounds(c1 - c0, world, rank)
    return c0 + a, c0 + b


def gather_chunk(out, chunk: Tuple[int, int], row_cells: int, group=None, async_op: bool = False):
    """Completes `out[chunk]` on every rank; rank r has filled its piece of the chunk."""
    import torch.distributed as dist

    world = dist.get_world_size(group)
    if world == 1 and not _FORCE_COLLECTIVES:
        return None
    rank = dist.get_rank(group)
    c0, c1 = chunk
    if (c1 - c0) % world == 0:
        a, b = piece_bounds(chunk, world, rank)
        return dist.all_gather_into_tensor(out[c0 * row_cells : c1 * row_cells], out[a * row_cells : b * row_cells],
                                           group=group, async_op=async_op)  # in place
    works = []  # uneven pieces: one broadcast per piece
    for r in range(world):
        a, b = piece_bounds(chunk, world, r)
        if b > a:
            src = dist.get_global_rank(group, r) if group is not None else r
            works.append(dist.broadcast(out[a * row_cells : b * row_cells], src=src, group=group, async_op=async_op))
    return works


def gather_slabs(out, nx: int, row_cells: int, group=None):
    """Single-chunk gather (every rank filled slab_bounds(nx, world, rank))."""
    gather_chunk(out, (0, nx), row_cells, group)
    return out


def _wait(work):
    if work is None:
        return
    for w in work if isinstance(work, list) else [work]:
        if w is not None:
            w.wait()


_SIDE_STREAMS = {}


def _side_streams(device):
    import torch

    key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = [torch.cuda.Stream(device=key), torch.cuda.Stream(device=key)]
    return _SIDE_STREAMS[key]


def run_pieces(mesh: Mesh, grid: Grid, sign_method: SignMethod, out, pieces: List[Tuple[int, int]],
               after_piece: Optional[Callable] = None, overlap: bool = True) -> list:
    """Enqueues the x-pieces `pieces` of one grid (device-resident `out`), calling `after_piece(i)` right
    after piece i has been enqueued — with the stream that computes piece i as torch's current stream, so a
    collective issued there is ordered after exactly that piece.  Returns what `after_piece` returned.
    On return the caller's current stream waits for every piece."""
    import torch

    def one(i):
        a, b = pieces[i]
        if b > a:
            mesh.generate_grid_sdf(grid, sign_method, x_slab=(a, b), out=out, synchronous=False)
        return after_piece(i) if after_piece is not None else None

    if not overlap or len(pieces) < 2 or os.environ.get("M2S_PIECE_STREAMS", "2") == "1":
        return [one(i) for i in range(len(pieces))]
    cur = torch.cuda.current_stream(out.device)
    streams = _side_streams(out.device)
    start = cur.record_event()
    results = []
    for i in range(len(pieces)):
        s = streams[i % 2]
        if i < 2:
            s.wait_event(start)     # inputs and `out` were produced on the caller's stream
        with torch.cuda.stream(s):
            results.append(one(i))
    for s in streams:
        cur.wait_stream(s)
        out.record_stream(s)
    return results


class PeerGrid:
    """This rank's whole-grid result buffer plus the mapped buffers of every other rank (m2s_shared_alloc +
    m2s_ipc_export / m2s_ipc_open; the 64-byte handles travel through `all_gather_object`).  With it the ranks deliver
    their x-slabs by writing them into each other's buffers over xGMI themselves (m2s_opts.peer_out) — no all-gather.
    Created once and reused for every step: mapping a peer's memory costs milliseconds."""

    def __init__(self, n_cells: int, device: int, group=None):
        import torch.distributed as dist

        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.peers, self.local, self.tensor = [], None, None
        handle, why = None, None
        try:
            self.local = SharedGrid(n_cells, device)
            handle = self.local.handle
        except Exception as e:   # noqa: BLE001  — every rank still takes part in the exchange below, or the others would hang
            why = e
        handles = [None] * self.world
        dist.all_gather_object(handles, handle, group=group)
        if any(h is None for h in handles):
            self.close(sync=False)
            raise RuntimeError(f"a rank could not allocate / export its shared grid buffer ({why})")
        # the same care for the second phase: a rank that cannot map a peer must not leave the others waiting in the step's barrier
        try:
            self.peers = [SharedGrid.open(h, n_cells, device) for r, h in enumerate(handles) if r != self.rank]
            self.tensor = self.local.tensor
        except Exception as e:   # noqa: BLE001
            why = e
        opened = [None] * self.world
        dist.all_gather_object(opened, why is None, group=group)
        if not all(opened):
            self.close(sync=False)
            raise RuntimeError(f"a rank could not map a peer's shared grid buffer ({why})")

    def close(self, sync: bool = True):
        """`sync`: barrier between unmapping the peers and freeing the own buffer (nobody frees under a peer that is still
        writing).  Pass False when the ranks did not all get this far (a collective would hang)."""
        import torch.distributed as dist

        for p in self.peers:
            p.close()
        self.peers = []
        if sync and dist.is_initialized():
            dist.barrier(self.group)
        self.tensor = None
        if self.local is not None:
            self.local.close()
            self.local = None


def generate_grid_sdf_sharded(vertices, indices: Topology, grid: Grid, sign_method: SignMethod = SignMethod.Raycast, *,
                              group=None, out=None, compute_slab: Optional[Callable] = None, chunks: int = 1,
                              mesh: Optional[Mesh] = None, gather: bool = True, return_mesh: bool = False,
                              peer_grid: Optional[PeerGrid] = None, peer_mode: PeerMode = PeerMode.Push, timings=None,
                              interleave: bool = True):
    """generate_grid_sdf over all ranks of `group` (default: the world).  `vertices`/`indices` are this
    rank's copies (CUDA tensors for the HIP path); returns the full grid on every rank.

    Two ways to deliver the slabs:
      peer_grid given   one m2s_generate_grid_sdf call per rank, writing into `peer_grid.tensor` and into every other rank's
                        buffer (peer pushes over xGMI), then a barrier: no collective moves data.  The rank's cells are the
                        chunks r and world + r of 2 world (`interleave`, m2s_opts.x_period: balanced — a contiguous middle
                        slab costs 30 % more than an outer one), or its contiguous x-slab where the grid does not allow that;
      otherwise         `chunks` contiguous x-ranges, each gathered in place by an asynchronous RCCL all-gather that
                        overlaps the next chunk's compute.

    Reuse of a PeerGrid across steps: the barrier at the end of a step orders the HOST threads.  A rank's next step starts
    writing into every rank's buffer as soon as it is called, so whatever consumes `out` on the GPU (a kernel of the caller's, a
    copy) must have finished on EVERY rank before ANY rank starts the next step — synchronise the consuming stream and put a
    barrier in front of the next call, or double-buffer with two PeerGrids (bench.py consumes nothing between steps).

    compute_slab(out, x0, x1) may replace the slab computation (the CPU tests inject the oracle there
    so the partition / overlap / gather logic runs under gloo without a GPU)."""
    import torch
    import torch.distributed as dist

    nx, ny, nz = grid.get_cell_count()
    inited = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if inited else 1
    rank = dist.get_rank(group) if inited else 0
    row = ny * nz
    if peer_grid is not None:
        a, b, period = interleaved_slab(grid, world, rank) if interleave else (*slab_bounds(nx, world, rank), 0)
        out = peer_grid.tensor
        generate_grid_sdf(vertices, indices, grid, sign_method, x_slab=(a, b), x_period=period, out=out, peer_out=peer_grid.peers,
                          peer_mode=peer_mode, timings=timings)   # synchronous: this rank's cells and their pushes are complete
        if world > 1:
            dist.barrier(group)                                   # ... and so are everybody else's into `out`
        return out
    if out is None:
        dev = vertices.device if hasattr(vertices, "device") else "cpu"
        out = torch.empty(nx * row, dtype=torch.float32, device=dev)
    plan = chunk_plan(nx, world, chunks if (world > 1 or _FORCE_COLLECTIVES) else 1)
    own_mesh = None
    if compute_slab is None and mesh is None:
        mesh = own_mesh = Mesh(vertices, indices)   # LBVH built once for all pieces of this call
    pending = []
    collective = gather and (world > 1 or (_FORCE_COLLECTIVES and inited))
    try:
        if compute_slab is not None:
            for ci, chunk in enumerate(plan):
                a, b = piece_bounds(chunk, world, rank)
                compute_slab(out, a, b)
                if collective:
                    pending.append(gather_chunk(out, chunk, row, group, async_op=True))
        else:
            # asynchronous: the kernels are only enqueued, so the collective of the previous chunk
            # (already running on RCCL's stream) overlaps them
            pending = run_pieces(mesh, grid, sign_method, out, [piece_bounds(ch, world, rank) for ch in plan],
                                 (lambda i: gather_chunk(out, plan[i], row, group, async_op=True)) if collective else None)
        for w in pending:
            _wait(w)
    finally:
        if own_mesh is not None and not return_mesh:
            if out.is_cuda:
                torch.cuda.current_stream(out.device).synchronize()
            own_mesh.close()
    return (out, mesh) if return_mesh else out


def query_bounds(n_queries: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous query range of rank `rank`; every rank gets ceil(n / world) queries except the tail ranks."""
    per = -(-n_queries // world) if world else n_queries
    q0 = min(rank * per, n_queries)
    return q0, min(q0 + per, n_queries)


def generate_sdf_sharded(vertices, indices: Topology, query_points, acceleration_method: AccelerationMethod = None, *,
                         group=None, out=None, compute_range: Optional[Callable] = None):
    """generate_sdf over all ranks of `group`: the query array is cut into `world` contiguous ranges (queries are
    independent given the replicated mesh), every rank computes its range and the results are
    all-gathered in place; returns all distances on every rank.  `query_points` is the FULL array on every rank.

    compute_range(out, q0, q1) may replace the computation (CPU tests inject the oracle)."""
    import torch
    import torch.distributed as dist

    inited = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if inited else 1
    rank = dist.get_rank(group) if inited else 0
    nq = int(query_points.shape[0])
    per = -(-nq // world) if nq else 0
    if out is None:
        dev = query_points.device if hasattr(query_points, "device") else "cpu"
        out = torch.empty(nq, dtype=torch.float32, device=dev)
    q0, q1 = query_bounds(nq, world, rank)
    padded = out
    if world > 1 and per * world != nq:
        padded = torch.empty(per * world, dtype=torch.float32, device=out.device)   # equal pieces for the in-place gather
    if q1 > q0:
        if compute_range is not None:
            compute_range(padded, q0, q1)
        else:
            padded[q0:q1] = generate_sdf(vertices, indices, query_points[q0:q1], acceleration_method)
    if world > 1 or (_FORCE_COLLECTIVES and inited):
        if per:
            dist.all_gather_into_tensor(padded[: per * world], padded[rank * per : (rank + 1) * per], group=group)
    if padded is not out:
        out.copy_(padded[:nq])
    return out
