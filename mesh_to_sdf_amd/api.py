"""Host-side mirror of the reference's public interface for the hot path, over the C ABI.

Same names, argument meaning and error behaviour as mesh_to_sdf 0.4.0
(/root/reference/mesh_to_sdf/src/lib.rs:151-311, generate/grid.rs:265-274, grid.rs:30-170), so
the parity tests read like the reference's own tests:

    sdf = generate_sdf(vertices, Topology.TriangleList(indices), query_points, AccelerationMethod.RtreeBvh)
    grid = Grid.from_bounding_box(bbox_min, bbox_max, [nx, ny, nz])
    sdf = generate_grid_sdf(vertices, Topology.TriangleList(indices), grid, SignMethod.Raycast)

numpy in -> numpy out (host pointers through the ABI: the drop-in case, H2D/D2H inside the call);
torch CUDA tensors in -> torch CUDA tensor out (device pointers, enqueued on torch's current
stream, nothing crosses PCIe).  Where the reference panics this raises M2SPanic.
"""
import ctypes as C
import enum
from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np

from . import _lib
from ._lib import M2SGrid, M2SOpts, M2STimings


class M2SError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"m2s error {code}: {msg}")
        self.code = code


class M2SPanic(M2SError):
    """The reference would have panicked here (index out of range, NaN distance, empty R-tree)."""


class SignMethod(enum.IntEnum):
    """lib.rs:204-216 (default Raycast)."""
    Raycast = 0
    Normal = 1


@dataclass(frozen=True)
class AccelerationMethod:
    """lib.rs:224-239: None(SignMethod) | Bvh(SignMethod) | Rtree | RtreeBvh (default)."""
    kind: int
    sign: SignMethod = SignMethod.Raycast

    @staticmethod
    def None_(sign: SignMethod = SignMethod.Raycast):
        return AccelerationMethod(0, SignMethod(sign))

    @staticmethod
    def Bvh(sign: SignMethod = SignMethod.Raycast):
        return AccelerationMethod(1, SignMethod(sign))


AccelerationMethod.Rtree = AccelerationMethod(2, SignMethod.Normal)
AccelerationMethod.RtreeBvh = AccelerationMethod(3, SignMethod.Raycast)


@dataclass(frozen=True)
class Topology:
    """lib.rs:151-167.  indices=None means 0..len(vertices)."""
    kind: int
    indices: Optional[object] = None

    @staticmethod
    def TriangleList(indices=None):
        return Topology(0, indices)

    @staticmethod
    def TriangleStrip(indices=None):
        return Topology(1, indices)


class Grid:
    """grid.rs:30-170.  All arithmetic is the reference's f32 arithmetic (through the C ABI helpers)."""

    def __init__(self, first_cell, cell_size, cell_count):  # Grid::new, grid.rs:43-49
        self._g = M2SGrid()
        for k in range(3):
            self._g.first_cell[k] = float(np.float32(first_cell[k]))
            self._g.cell_size[k] = float(np.float32(cell_size[k]))
            self._g.cell_count[k] = int(cell_count[k])

    new = classmethod(lambda cls, first_cell, cell_size, cell_count: cls(first_cell, cell_size, cell_count))

    @classmethod
    def from_bounding_box(cls, bbox_min, bbox_max, cell_count):  # grid.rs:59-74
        g = cls.__new__(cls)
        g._g = M2SGrid()
        mn = (C.c_float * 3)(*[float(np.float32(v)) for v in bbox_min])
        mx = (C.c_float * 3)(*[float(np.float32(v)) for v in bbox_max])
        cnt = (C.c_uint64 * 3)(*[int(v) for v in cell_count])
        _lib.lib().m2s_grid_from_bounding_box(mn, mx, cnt, C.byref(g._g))
        return g

    def __eq__(self, other):  # #[derive(PartialEq)], grid.rs:27
        return isinstance(other, Grid) and bytes(self._g) == bytes(other._g)

    def __repr__(self):
        return f"Grid(first_cell={list(self._g.first_cell)}, cell_size={list(self._g.cell_size)}, cell_count={list(self._g.cell_count)})"

    def get_first_cell(self):
        return np.array(list(self._g.first_cell), np.float32)

    def get_cell_size(self):
        return np.array(list(self._g.cell_size), np.float32)

    def get_cell_count(self):
        return [int(v) for v in self._g.cell_count]

    def get_total_cell_count(self):
        c = self.get_cell_count()
        return c[0] * c[1] * c[2]

    def get_last_cell(self):  # grid.rs:82-88 (first + count * size, as written in the reference)
        f, s, c = self.get_first_cell(), self.get_cell_size(), self.get_cell_count()
        return (f + np.array(c, np.float32) * s).astype(np.float32)

    def get_bounding_box(self):  # grid.rs:110-119
        f, s, c = self.get_first_cell(), self.get_cell_size(), self.get_cell_count()
        mn = (f - s * np.float32(0.5)).astype(np.float32)
        mx = (mn + np.array(c, np.float32) * s).astype(np.float32)
        return mn, mx

    def get_cell_idx(self, cell):  # grid.rs:122-124
        return int(_lib.lib().m2s_grid_cell_idx(C.byref(self._g), (C.c_uint64 * 3)(*[int(v) for v in cell])))

    def get_cell_integer_coordinates(self, cell_idx):  # grid.rs:127-132
        c = self.get_cell_count()
        return [cell_idx // (c[1] * c[2]), (cell_idx // c[2]) % c[1], cell_idx % c[2]]

    def get_cell_center(self, cell):  # grid.rs:135-141
        out = (C.c_float * 3)()
        _lib.lib().m2s_grid_cell_center(C.byref(self._g), (C.c_uint64 * 3)(*[int(v) for v in cell]), out)
        return np.array(list(out), np.float32)

    def snap_point_to_grid(self, point):  # grid.rs:145-170 -> ("Inside"|"Outside", [x, y, z])
        mn, _ = self.get_bounding_box()
        s, c = self.get_cell_size(), self.get_cell_count()
        with np.errstate(all="ignore"):
            cell = np.floor((np.asarray(point, np.float32) - mn) / s)
        cell = [0 if not np.isfinite(v) and np.isnan(v) else int(np.clip(v, -2.0**62, 2.0**62)) for v in cell]
        res = [min(max(cell[k], 0), c[k] - 1) for k in range(3)]
        return ("Inside" if res == cell else "Outside"), res

    def __eq__(self, other):
        return isinstance(other, Grid) and bytes(self._g) == bytes(other._g)


def _raise(rc):
    msg = _lib.last_error()
    if rc in (_lib.ERR_BAD_ARG, _lib.ERR_NAN, _lib.ERR_EMPTY_MESH):
        raise M2SPanic(rc, msg)
    raise M2SError(rc, msg)


def _is_torch(x):
    return type(x).__module__.startswith("torch")


class _Args:
    """Normalises (vertices, topology[, queries]) to raw pointers for one call."""

    def __init__(self, vertices, topology: Topology, queries=None):
        self.keep = []
        self.device = _is_torch(vertices) and vertices.is_cuda
        if self.device:
            import torch

            self.torch = torch
            self.dev = vertices.device
            v = vertices.detach().to(torch.float32).contiguous().reshape(-1, 3)
            self.n_verts = v.shape[0]
            self.p_verts = v.data_ptr() if v.numel() else None
            self.keep.append(v)
            idx = topology.indices
            self.index_bytes = 4
            if idx is None:
                self.p_idx, self.n_idx = None, 0
            else:
                if not _is_torch(idx):
                    idx = torch.as_tensor(np.ascontiguousarray(idx).astype(np.int64), device=self.dev)
                idx = idx.to(device=self.dev, dtype=torch.int32).contiguous().reshape(-1)  # bit pattern == u32
                self.n_idx = idx.numel()
                self.p_idx = idx.data_ptr() if idx.numel() else _dummy_device_ptr(torch, self.dev, self.keep)
                self.keep.append(idx)
            if queries is not None:
                q = queries if _is_torch(queries) else torch.as_tensor(np.asarray(queries, np.float32), device=self.dev)
                q = q.detach().to(device=self.dev, dtype=torch.float32).contiguous().reshape(-1, 3)
                self.n_q = q.shape[0]
                self.p_q = q.data_ptr() if q.numel() else None
                self.keep.append(q)
        else:
            v = np.ascontiguousarray(np.asarray(vertices, np.float32)).reshape(-1, 3)
            self.n_verts = v.shape[0]
            self.p_verts = v.ctypes.data if v.size else None
            self.keep.append(v)
            idx = topology.indices
            if idx is None:
                self.p_idx, self.n_idx, self.index_bytes = None, 0, 4
            else:
                idx = np.asarray(idx)
                if idx.dtype == np.uint16:
                    idx = np.ascontiguousarray(idx).reshape(-1)
                    self.index_bytes = 2
                else:
                    if idx.size and (idx.min() < 0 or idx.max() > 0xFFFFFFFF):
                        raise M2SPanic(_lib.ERR_BAD_ARG, "index does not fit u32")
                    idx = np.ascontiguousarray(idx.astype(np.uint32)).reshape(-1)
                    self.index_bytes = 4
                self.n_idx = idx.size
                self._empty = np.zeros(4, np.uint32)
                self.p_idx = idx.ctypes.data if idx.size else self._empty.ctypes.data
                self.keep.append(idx)
            if queries is not None:
                q = np.ascontiguousarray(np.asarray(queries, np.float32)).reshape(-1, 3)
                self.n_q = q.shape[0]
                self.p_q = q.ctypes.data if q.size else None
                self.keep.append(q)
        self.topology = topology.kind

    def opts(self, timings=None, algorithm=0, x_begin=0, x_end=0, synchronous=True, peer_out=None, peer_mode=0, lane=0, x_period=0):
        o = M2SOpts()
        o.struct_size = C.sizeof(M2SOpts)
        o.algorithm = int(algorithm)
        o.x_begin, o.x_end = int(x_begin), int(x_end)
        o.x_period = int(x_period)
        o.synchronous = 1 if synchronous else 0
        o.lane = int(lane)
        if peer_out:
            ptrs = [int(p.data_ptr()) if hasattr(p, "data_ptr") else int(p) for p in peer_out]
            arr = (C.c_void_p * len(ptrs))(*ptrs)
            self.keep.append(arr)
            o.n_peer_out = len(ptrs)
            o.peer_out = C.cast(arr, C.POINTER(C.c_void_p))
            o.peer_mode = int(peer_mode)
        if timings is not None:
            o.timings = C.pointer(timings)
        if self.device:
            o.device = self.dev.index if self.dev.index is not None else self.torch.cuda.current_device()
            # torch's current stream, exactly: its default stream has handle 0, which must NOT be read as
            # "pick your own stream" or work torch orders after this call (collectives!) would race with it
            o.stream = self.torch.cuda.current_stream(self.dev).cuda_stream
            o.stream_mode = 1
            o.mem_kind = _lib.MEM_DEVICE
        else:
            o.device = -1
            o.mem_kind = _lib.MEM_HOST
        return o


def _dummy_device_ptr(torch, dev, keep):
    t = torch.zeros(4, dtype=torch.int32, device=dev)
    keep.append(t)
    return t.data_ptr()


def generate_sdf(vertices, indices: Topology, query_points, acceleration_method: AccelerationMethod = None, *,
                 timings: M2STimings = None, algorithm: int = 0):
    """lib.rs:291-311."""
    am = acceleration_method if acceleration_method is not None else AccelerationMethod.RtreeBvh
    a = _Args(vertices, indices, query_points)
    n_out = C.c_size_t(0)
    if a.device:
        out = a.torch.empty(a.n_q, dtype=a.torch.float32, device=a.dev)
        p_out = out.data_ptr() if a.n_q else None
    else:
        out = np.empty(a.n_q, np.float32)
        p_out = out.ctypes.data if a.n_q else None
    o = a.opts(timings, algorithm)
    rc = _lib.lib().m2s_generate_sdf(a.p_verts, a.n_verts, a.p_idx, a.n_idx, a.index_bytes, a.topology, a.p_q, a.n_q,
                                     int(am.kind), int(am.sign), p_out, C.byref(n_out), C.byref(o))
    if rc != _lib.M2S_OK:
        _raise(rc)
    return out[: n_out.value]


class PeerMode(enum.IntEnum):
    """include/m2s.h `m2s_peer_mode`: how a slab reaches the peers' whole-grid buffers."""
    Push = 0    # one wide copy kernel per slab piece, overlapped with the next piece's walk
    Store = 1   # the walk's epilogue stores every value to every peer
    Trail = 2   # one walk; a copy kernel beside it pushes each unit of 8 x-layers as soon as the walk has finished it


class Partition(enum.IntEnum):
    """include/m2s.h `m2s_partition`."""
    Auto = 0
    Contiguous = 1
    Interleaved = 2
    Adaptive = 3


class Exchange(enum.IntEnum):
    """include/m2s.h `m2s_exchange` (device-resident results of generate_grid_sdf_multi)."""
    Auto = 0
    Peer = 1
    Rccl = 2
    Nothing = 3


def generate_grid_sdf(vertices, indices: Topology, grid: Grid, sign_method: SignMethod = SignMethod.Raycast, *,
                      timings: M2STimings = None, algorithm: int = 0, x_slab: Sequence[int] = None, out=None,
                      peer_out=None, peer_mode: PeerMode = PeerMode.Push, lane: int = 0, synchronous: bool = True,
                      x_period: int = 0):
    """generate/grid.rs:265-378.  `x_slab=(x0, x1)` computes only cells with x0 <= x < x1 (the rest
    of `out` is left untouched); used by the multi-GPU driver in distributed.py.  With `x_period` > 0 the call owns the
    interleaved chunks [x0 + j * x_period, x1 + j * x_period), j = 0, 1, ... (m2s_opts.x_period).  `peer_out`: whole-grid
    device buffers (tensors or raw pointers, usually on other GPUs) that receive the same cells (m2s_opts.peer_out)."""
    a = _Args(vertices, indices)
    total = grid.get_total_cell_count()
    if out is None:
        if a.device:
            out = a.torch.empty(total, dtype=a.torch.float32, device=a.dev)
        else:
            out = np.empty(total, np.float32)
    if a.device:
        assert out.is_cuda and out.dtype == a.torch.float32 and out.numel() == total and out.is_contiguous()
        p_out = out.data_ptr() if total else None
    else:
        assert out.dtype == np.float32 and out.size == total and out.flags["C_CONTIGUOUS"]
        p_out = out.ctypes.data if total else None
    xb, xe = (0, 0) if x_slab is None else (int(x_slab[0]), int(x_slab[1]))
    if x_slab is not None and xb == xe:
        return out
    o = a.opts(timings, algorithm, xb, xe, synchronous or not a.device, peer_out, peer_mode, lane, x_period)
    rc = _lib.lib().m2s_generate_grid_sdf(a.p_verts, a.n_verts, a.p_idx, a.n_idx, a.index_bytes, a.topology,
                                          C.byref(grid._g), int(sign_method), p_out, C.byref(o))
    if rc != _lib.M2S_OK:
        _raise(rc)
    return out


def interleaved_slab(grid: "Grid", n: int, k: int):
    """m2s_interleaved_slab: (x0, x1, x_period) of shard k of n — the chunks k and n + k of 2n where the grid allows it
    (x_period > 0), else the contiguous slab with x_period = 0."""
    a, b, p = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
    _lib.lib().m2s_interleaved_slab(C.byref(grid._g), int(n), int(k), C.byref(a), C.byref(b), C.byref(p))
    return int(a.value), int(b.value), int(p.value)


def slab_bounds(nx: int, n: int, k: int):
    """m2s_slab_bounds: contiguous x-slab [x0, x1) of shard k out of n (sizes differ by at most one layer)."""
    a, b = C.c_uint64(0), C.c_uint64(0)
    _lib.lib().m2s_slab_bounds(int(nx), int(n), int(k), C.byref(a), C.byref(b))
    return int(a.value), int(b.value)


def generate_grid_sdf_multi(vertices, indices: Topology, grid: Grid, sign_method: SignMethod = SignMethod.Raycast, *,
                            devices: Sequence[int] = None, outs=None, exchange: Exchange = Exchange.Auto,
                            peer_mode: PeerMode = PeerMode.Push, algorithm: int = 0, info: dict = None,
                            partition: Partition = Partition.Auto):
    """generate_grid_sdf over several GPUs from this one process (m2s_generate_grid_sdf_multi): one host thread per
    device inside the library, contiguous x-slabs, no data-path collective.

    numpy in  -> one numpy array out (every device streams its slab into it over its own PCIe link);
    torch CUDA tensors in (on devices[0]) -> a list of whole-grid CUDA tensors, one per entry of `devices`, each
    holding the WHOLE grid on return (peer writes over xGMI, or RCCL where peer access is unavailable).
    `devices` may repeat a device (two shards on one GPU).  `info` (optional dict) receives wall_ms, exchange, timings."""
    a = _Args(vertices, indices)
    total = grid.get_total_cell_count()
    L = _lib.lib()
    if devices is None:
        devices = list(range(L.m2s_device_count()))
    devices = [int(d) for d in devices]
    n = len(devices)
    mo = _lib.M2SMultiOpts()
    mo.struct_size = C.sizeof(_lib.M2SMultiOpts)
    mo.n_devices = n
    dev_arr = (C.c_int32 * max(n, 1))(*devices)
    mo.devices = C.cast(dev_arr, C.POINTER(C.c_int32))
    mo.exchange = int(exchange)
    mo.peer_mode = int(peer_mode)
    mo.algorithm = int(algorithm)
    mo.partition = int(partition)
    tims = (M2STimings * max(n, 1))()
    mo.timings = C.cast(tims, C.POINTER(M2STimings))
    wall, used, part_used = C.c_float(0.0), C.c_int32(-1), C.c_int32(-1)
    mo.wall_ms = C.pointer(wall)
    mo.exchange_used = C.pointer(used)
    mo.partition_used = C.pointer(part_used)
    slabs = (C.c_uint64 * (3 * max(n, 1)))()
    mo.slabs = C.cast(slabs, C.POINTER(C.c_uint64))
    if a.device:
        mo.mem_kind = _lib.MEM_DEVICE
        dev0 = a.dev.index if a.dev.index is not None else a.torch.cuda.current_device()
        assert n == 0 or devices[0] == dev0, "the mesh tensors must live on devices[0]"
        if outs is None:
            outs = [a.torch.empty(total, dtype=a.torch.float32, device=f"cuda:{d}") for d in devices]
        assert len(outs) == n and all(o.is_cuda and o.dtype == a.torch.float32 and o.numel() == total and o.is_contiguous() for o in outs)
        a.torch.cuda.synchronize(a.dev)      # the library uses streams of its own: inputs must be complete
        ptrs = (C.c_void_p * max(n, 1))(*[(o.data_ptr() if total else None) for o in outs])
        result = outs
    else:
        mo.mem_kind = _lib.MEM_HOST
        if outs is None:
            outs = np.empty(total, np.float32)
        assert outs.dtype == np.float32 and outs.size == total and outs.flags["C_CONTIGUOUS"]
        ptrs = (C.c_void_p * 1)(outs.ctypes.data if total else None)
        result = outs
    rc = L.m2s_generate_grid_sdf_multi(a.p_verts, a.n_verts, a.p_idx, a.n_idx, a.index_bytes, a.topology, C.byref(grid._g),
                                       int(sign_method), C.cast(ptrs, C.POINTER(C.c_void_p)), C.byref(mo))
    if rc != _lib.M2S_OK:
        _raise(rc)
    if info is not None:
        info["wall_ms"] = float(wall.value)
        info["exchange"] = Exchange(used.value).name if used.value >= 0 else None
        info["timings"] = [tims[k] for k in range(n)]
        info["partition"] = Partition(part_used.value).name if part_used.value >= 0 else None
        info["slabs"] = [(int(slabs[3 * k]), int(slabs[3 * k + 1]), int(slabs[3 * k + 2])) for k in range(n)]
        info["_keep"] = tims
    return result


def warmup(device: int = -1, workspace_bytes: int = 0, host_ring_bytes: int = 0):
    """m2s_warmup: pay the one-off costs of a process's first call now (runtime, code objects, context, optional workspace / pinned ring)."""
    rc = _lib.lib().m2s_warmup(int(device), int(workspace_bytes), int(host_ring_bytes))
    if rc != _lib.M2S_OK:
        _raise(rc)


def peer_bandwidth(src, peers, n_cells: int = None):
    """m2s_peer_bandwidth: GB/s of the peer-push copy kernel from the CUDA tensor `src` into each tensor / SharedGrid of `peers` (one at
    a time) and into all of them at once.  Returns (per_peer_gbps, all_together_gbps)."""
    n = len(peers)
    if n == 0:
        return [], 0.0
    n_cells = int(n_cells if n_cells is not None else src.numel())
    ptrs = (C.c_void_p * n)(*[int(p.data_ptr()) for p in peers])
    each = (C.c_float * n)()
    allg = C.c_float(0.0)
    dev = src.device.index if src.device.index is not None else -1
    rc = _lib.lib().m2s_peer_bandwidth(int(src.data_ptr()), C.cast(ptrs, C.POINTER(C.c_void_p)), n, n_cells, dev, each, C.byref(allg))
    if rc != _lib.M2S_OK:
        _raise(rc)
    return [float(x) for x in each], float(allg.value)


def balanced_slabs(nx: int, unit: int, prev_bounds: Sequence[int], cost: Sequence[float]):
    """m2s_balanced_slabs: n + 1 slab boundaries of equal cost from the boundaries and per-shard costs of a previous call."""
    n = len(cost)
    assert len(prev_bounds) == n + 1
    pb = (C.c_uint64 * (n + 1))(*[int(b) for b in prev_bounds])
    cs = (C.c_float * n)(*[float(c) for c in cost])
    nb = (C.c_uint64 * (n + 1))()
    rc = _lib.lib().m2s_balanced_slabs(int(nx), n, int(unit), pb, cs, nb)
    if rc != _lib.M2S_OK:
        _raise(rc)
    return [int(b) for b in nb]


def generate_sdf_multi(vertices, indices: Topology, query_points, acceleration_method: AccelerationMethod = None, *,
                       devices: Sequence[int] = None, outs=None, exchange: Exchange = Exchange.Auto, algorithm: int = 0,
                       info: dict = None):
    """generate_sdf over several GPUs from this one process (m2s_generate_sdf_multi): shard k computes a contiguous range of the
    queries.  numpy in -> one numpy array out; torch CUDA tensors in (on devices[0]) -> a list of CUDA tensors, one per entry of
    `devices`, each holding ALL distances on return.  `devices` may repeat a device."""
    am = acceleration_method if acceleration_method is not None else AccelerationMethod.RtreeBvh
    a = _Args(vertices, indices, query_points)
    L = _lib.lib()
    if devices is None:
        devices = list(range(L.m2s_device_count()))
    devices = [int(d) for d in devices]
    n = len(devices)
    mo = _lib.M2SMultiOpts()
    mo.struct_size = C.sizeof(_lib.M2SMultiOpts)
    mo.n_devices = n
    dev_arr = (C.c_int32 * max(n, 1))(*devices)
    mo.devices = C.cast(dev_arr, C.POINTER(C.c_int32))
    mo.exchange = int(exchange)
    mo.algorithm = int(algorithm)
    tims = (M2STimings * max(n, 1))()
    mo.timings = C.cast(tims, C.POINTER(M2STimings))
    wall, used = C.c_float(0.0), C.c_int32(-1)
    mo.wall_ms = C.pointer(wall)
    mo.exchange_used = C.pointer(used)
    if a.device:
        mo.mem_kind = _lib.MEM_DEVICE
        dev0 = a.dev.index if a.dev.index is not None else a.torch.cuda.current_device()
        assert n == 0 or devices[0] == dev0, "mesh and queries must live on devices[0]"
        if outs is None:
            outs = [a.torch.empty(a.n_q, dtype=a.torch.float32, device=f"cuda:{d}") for d in devices]
        assert len(outs) == n and all(o.is_cuda and o.dtype == a.torch.float32 and o.numel() == a.n_q and o.is_contiguous() for o in outs)
        a.torch.cuda.synchronize(a.dev)      # the library uses streams of its own: inputs must be complete
        ptrs = (C.c_void_p * max(n, 1))(*[(o.data_ptr() if a.n_q else None) for o in outs])
    else:
        mo.mem_kind = _lib.MEM_HOST
        if outs is None:
            outs = np.empty(a.n_q, np.float32)
        assert outs.dtype == np.float32 and outs.size == a.n_q and outs.flags["C_CONTIGUOUS"]
        ptrs = (C.c_void_p * 1)(outs.ctypes.data if a.n_q else None)
    n_out = C.c_size_t(0)
    rc = L.m2s_generate_sdf_multi(a.p_verts, a.n_verts, a.p_idx, a.n_idx, a.index_bytes, a.topology, a.p_q, a.n_q, int(am.kind),
                                  int(am.sign), C.cast(ptrs, C.POINTER(C.c_void_p)), C.byref(n_out), C.byref(mo))
    if rc != _lib.M2S_OK:
        _raise(rc)
    if info is not None:
        info["wall_ms"] = float(wall.value)
        info["exchange"] = Exchange(used.value).name if used.value >= 0 else None
        info["timings"] = [tims[k] for k in range(n)]
        info["_keep"] = tims
    if a.device:
        return [o[: n_out.value] for o in outs]
    return outs[: n_out.value]


class SharedGrid:
    """A whole-grid device buffer other PROCESSES can map (m2s_shared_alloc / m2s_ipc_*): the one-process-per-GPU form of
    the peer exchange.  `tensor` views it as a torch CUDA tensor; `handle` is the 64-byte token to send to the other ranks;
    `SharedGrid.open(handle, device)` maps a peer's buffer and returns its device pointer wrapper."""

    def __init__(self, n_cells: int, device: int):
        self.n, self.device, self.owner = int(n_cells), int(device), True
        p = C.c_void_p()
        rc = _lib.lib().m2s_shared_alloc(self.n * 4, self.device, C.byref(p))
        if rc != _lib.M2S_OK:
            _raise(rc)
        self.ptr = int(p.value)

    @property
    def handle(self) -> bytes:
        buf = C.create_string_buffer(_lib.IPC_HANDLE_BYTES)
        rc = _lib.lib().m2s_ipc_export(self.ptr, buf)
        if rc != _lib.M2S_OK:
            _raise(rc)
        return bytes(buf.raw)

    @classmethod
    def open(cls, handle: bytes, n_cells: int, device: int):
        g = cls.__new__(cls)
        g.n, g.device, g.owner = int(n_cells), int(device), False
        p = C.c_void_p()
        rc = _lib.lib().m2s_ipc_open(C.create_string_buffer(bytes(handle), _lib.IPC_HANDLE_BYTES), g.device, C.byref(p))
        if rc != _lib.M2S_OK:
            _raise(rc)
        g.ptr = int(p.value)
        return g

    def data_ptr(self):
        return self.ptr

    @property
    def __cuda_array_interface__(self):
        return {"shape": (self.n,), "typestr": "<f4", "data": (self.ptr, False), "version": 2, "strides": None}

    @property
    def tensor(self):
        import torch

        t = torch.as_tensor(self, device=f"cuda:{self.device}")
        assert t.data_ptr() == self.ptr, "torch copied the buffer instead of wrapping it"
        return t

    def close(self):
        if getattr(self, "ptr", 0):
            L = _lib.lib()
            (L.m2s_shared_free if self.owner else L.m2s_ipc_close)(self.ptr, self.device)
            self.ptr = 0

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Mesh:
    """Persistent mesh (include/m2s.h `m2s_mesh`): triangle records + LBVH built once and kept on the
    device; `generate_grid_sdf` / `generate_sdf` then skip the build, and the sign planes of the last
    grid are cached.  Results are identical to the one-shot functions.  The reference rebuilds its
    trees in every call (generate/grid.rs:95-111); its client regenerates on every parameter change
    (mesh_to_sdf_client/src/sdf.rs:32-137), which is what this object is for."""

    def __init__(self, vertices, indices: Topology):
        self._a = _Args(vertices, indices)
        self._h = C.c_void_p()
        o = self._a.opts()
        rc = _lib.lib().m2s_mesh_create(self._a.p_verts, self._a.n_verts, self._a.p_idx, self._a.n_idx, self._a.index_bytes,
                                        self._a.topology, C.byref(o), C.byref(self._h))
        if rc != _lib.M2S_OK:
            self._h = C.c_void_p()
            _raise(rc)

    @property
    def device(self):
        return self._a.device

    def triangle_count(self):
        return int(_lib.lib().m2s_mesh_triangle_count(self._h))

    def generate_grid_sdf(self, grid: Grid, sign_method: SignMethod = SignMethod.Raycast, *, timings: M2STimings = None,
                          algorithm: int = 0, x_slab: Sequence[int] = None, out=None, synchronous: bool = True,
                          peer_out=None, peer_mode: PeerMode = PeerMode.Push):
        a = self._a
        total = grid.get_total_cell_count()
        if out is None:
            out = a.torch.empty(total, dtype=a.torch.float32, device=a.dev) if a.device else np.empty(total, np.float32)
        p_out = (out.data_ptr() if a.device else out.ctypes.data) if total else None
        xb, xe = (0, 0) if x_slab is None else (int(x_slab[0]), int(x_slab[1]))
        if x_slab is not None and xb == xe:
            return out
        o = a.opts(timings, algorithm, xb, xe, synchronous or not a.device, peer_out, peer_mode)
        rc = _lib.lib().m2s_mesh_generate_grid_sdf(self._h, C.byref(grid._g), int(sign_method), p_out, C.byref(o))
        if rc != _lib.M2S_OK:
            _raise(rc)
        return out

    def generate_sdf(self, query_points, acceleration_method: AccelerationMethod = None, *, timings: M2STimings = None,
                     algorithm: int = 0):
        am = acceleration_method if acceleration_method is not None else AccelerationMethod.RtreeBvh
        a = self._a
        if a.device:
            q = query_points if _is_torch(query_points) else a.torch.as_tensor(np.asarray(query_points, np.float32), device=a.dev)
            q = q.detach().to(device=a.dev, dtype=a.torch.float32).contiguous().reshape(-1, 3)
            n_q, p_q = q.shape[0], (q.data_ptr() if q.numel() else None)
            out = a.torch.empty(n_q, dtype=a.torch.float32, device=a.dev)
            p_out = out.data_ptr() if n_q else None
        else:
            q = np.ascontiguousarray(np.asarray(query_points, np.float32)).reshape(-1, 3)
            n_q, p_q = q.shape[0], (q.ctypes.data if q.size else None)
            out = np.empty(n_q, np.float32)
            p_out = out.ctypes.data if n_q else None
        n_out = C.c_size_t(0)
        o = a.opts(timings, algorithm)
        rc = _lib.lib().m2s_mesh_generate_sdf(self._h, p_q, n_q, int(am.kind), int(am.sign), p_out, C.byref(n_out), C.byref(o))
        if rc != _lib.M2S_OK:
            _raise(rc)
        return out[: n_out.value]

    def debug_digest(self):
        """FNV-1a digests of the resident arrays (test hook `m2s_debug_mesh_digest`): triangle records, pre-test planes, box nodes,
        oriented bounds, centroids, slot table, scene words, triangle count."""
        L = _lib.lib()
        L.m2s_debug_mesh_digest.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        L.m2s_debug_mesh_digest.restype = C.c_int
        out = (C.c_uint64 * 8)()
        rc = L.m2s_debug_mesh_digest(self._h, out)
        if rc != _lib.M2S_OK:
            _raise(rc)
        return [int(x) for x in out]

    def drain_timings(self) -> M2STimings:
        t = M2STimings()
        rc = _lib.lib().m2s_mesh_drain_timings(self._h, C.byref(t))
        if rc != _lib.M2S_OK:
            _raise(rc)
        return t

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            _lib.lib().m2s_mesh_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
