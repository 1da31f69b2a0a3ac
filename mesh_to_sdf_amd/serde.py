"""Host-side mirror of mesh_to_sdf::serde (mesh_to_sdf/src/serde.rs:75-221) over the C ABI.

    ser = SerializeSdf.Grid(SerializeGrid(grid, distances))
    save_to_file(ser, "sdf.bin")
    de = read_from_file("sdf.bin")          # DeserializeGrid | DeserializeGeneric

The bytes are the reference's V1 container (rmp-serde compact MessagePack), byte for byte; the
payload arrays are encoded / decoded by HIP kernels (csrc/serde.hip).  numpy in -> numpy / bytes out;
torch CUDA tensors in -> the encoded container stays on the GPU (a uint8 tensor) unless it goes to a file.
"""
import ctypes as C
import os
from dataclasses import dataclass
from typing import Union

import numpy as np

from . import _lib
from ._lib import M2SOpts, M2SSdfInfo
from .api import Grid, M2SError, _is_torch


class SerdeError(M2SError):
    """serde.rs:43-51: SerializationFailed | DeserializationFailed | IoError (see .kind)."""

    def __init__(self, code, msg):
        super().__init__(code, msg)
        self.kind = "IoError" if code == _lib.ERR_IO else msg.split(":")[0] if ":" in msg else "SerdeError"


@dataclass
class SerializeGeneric:  # serde.rs:87-95
    query_points: object
    distances: object


@dataclass
class SerializeGrid:  # serde.rs:99-107
    grid: Grid
    distances: object


class SerializeSdf:  # serde.rs:75-83
    Generic = staticmethod(lambda g: g if isinstance(g, SerializeGeneric) else SerializeGeneric(*g))
    Grid = staticmethod(lambda g: g if isinstance(g, SerializeGrid) else SerializeGrid(*g))


@dataclass
class DeserializeGeneric:  # serde.rs:131-139
    query_points: object
    distances: object


@dataclass
class DeserializeGrid:  # serde.rs:143-151
    grid: Grid
    distances: object


def _raise(rc):
    raise SerdeError(rc, _lib.last_error())


def _opts(device_tensor=None, synchronous=True):
    o = M2SOpts()
    o.struct_size = C.sizeof(M2SOpts)
    o.synchronous = 1 if synchronous else 0
    if device_tensor is not None:
        import torch

        dev = device_tensor.device
        o.device = dev.index if dev.index is not None else torch.cuda.current_device()
        o.stream = torch.cuda.current_stream(dev).cuda_stream
        o.stream_mode = 1
        o.mem_kind = _lib.MEM_DEVICE
    else:
        o.device = -1
        o.mem_kind = _lib.MEM_HOST
    return o


class _F32:
    """A float array argument as (pointer, element count, keep-alive, device tensor or None)."""

    def __init__(self, x, width=1):
        if _is_torch(x) and x.is_cuda:
            import torch

            t = x.detach().to(torch.float32).contiguous()
            self.keep, self.dev, self.ptr, self.n = t, t, t.data_ptr(), t.numel() // width
        else:
            if _is_torch(x):
                x = x.detach().cpu().numpy()
            a = np.ascontiguousarray(np.asarray(x, dtype=np.float32))
            self.keep, self.dev, self.ptr, self.n = a, None, a.ctypes.data, a.size // width
        if width == 3 and (self.keep.shape[-1] if self.n else 3) != 3:
            raise ValueError("query_points must have shape (Q, 3)")


def encoded_size(sdf: Union[SerializeGeneric, SerializeGrid]) -> int:
    L = _lib.lib()
    if isinstance(sdf, SerializeGrid):
        return L.m2s_sdf_grid_encoded_size(C.byref(sdf.grid._g), _F32(sdf.distances).n)
    return L.m2s_sdf_generic_encoded_size(_F32(sdf.query_points, 3).n, _F32(sdf.distances).n)


def serialize(sdf: Union[SerializeGeneric, SerializeGrid], synchronous=True):
    """serde.rs:161-166.  Returns `bytes` for host inputs, a uint8 CUDA tensor for CUDA inputs."""
    L = _lib.lib()
    d = _F32(sdf.distances)
    is_grid = isinstance(sdf, SerializeGrid)
    q = None if is_grid else _F32(sdf.query_points, 3)
    if q is not None and (q.dev is None) != (d.dev is None):
        raise ValueError("query_points and distances must live on the same side (both host or both CUDA)")
    size = (L.m2s_sdf_grid_encoded_size(C.byref(sdf.grid._g), d.n) if is_grid
            else L.m2s_sdf_generic_encoded_size(q.n, d.n))
    if size == 0:
        raise SerdeError(_lib.ERR_BAD_ARG, "SerializationFailed: element count exceeds a MessagePack array")
    written = C.c_size_t(0)
    if d.dev is not None:
        import torch

        out = torch.empty(size, dtype=torch.uint8, device=d.dev.device)
        out_ptr = out.data_ptr()
    else:
        out = np.empty(size, np.uint8)
        out_ptr = out.ctypes.data
    o = _opts(d.dev, synchronous)
    if is_grid:
        rc = L.m2s_sdf_encode_grid(C.byref(sdf.grid._g), d.ptr, d.n, out_ptr, size, C.byref(written), C.byref(o))
    else:
        rc = L.m2s_sdf_encode_generic(q.ptr, q.n, d.ptr, d.n, out_ptr, size, C.byref(written), C.byref(o))
    if rc != 0:
        _raise(rc)
    assert written.value == size
    return out if d.dev is not None else out.tobytes()


def _info_to_grid(info):
    return Grid(list(info.grid.first_cell), list(info.grid.cell_size), list(info.grid.cell_count))


def probe(data) -> M2SSdfInfo:
    L = _lib.lib()
    info = M2SSdfInfo()
    if _is_torch(data) and data.is_cuda:
        o = _opts(data)
        rc = L.m2s_sdf_probe(data.data_ptr(), data.numel(), C.byref(info), C.byref(o))
    else:
        buf = np.frombuffer(bytes(data), np.uint8) if not isinstance(data, np.ndarray) else data
        rc = L.m2s_sdf_probe(buf.ctypes.data, buf.size, C.byref(info), None)
    if rc != 0:
        _raise(rc)
    return info


def deserialize(data):
    """serde.rs:169-176.  `bytes`/numpy -> numpy arrays; uint8 CUDA tensor -> CUDA tensors."""
    L = _lib.lib()
    info = probe(data)
    nq, nd = int(info.n_queries), int(info.n_distances)
    if _is_torch(data) and data.is_cuda:
        import torch

        dist = torch.empty(nd, dtype=torch.float32, device=data.device)
        qp = torch.empty((nq, 3), dtype=torch.float32, device=data.device)
        o = _opts(data)
        rc = L.m2s_sdf_decode(data.data_ptr(), data.numel(), qp.data_ptr(), dist.data_ptr(), C.byref(o))
    else:
        buf = np.frombuffer(bytes(data), np.uint8) if not isinstance(data, np.ndarray) else data
        dist = np.empty(nd, np.float32)
        qp = np.empty((nq, 3), np.float32)
        o = _opts(None)
        rc = L.m2s_sdf_decode(buf.ctypes.data, buf.size, qp.ctypes.data, dist.ctypes.data, C.byref(o))
    if rc != 0:
        _raise(rc)
    if info.kind == 1:
        return DeserializeGrid(_info_to_grid(info), dist)
    return DeserializeGeneric(qp, dist)


def save_to_file(sdf: Union[SerializeGeneric, SerializeGrid], path):
    """serde.rs:192-198."""
    L = _lib.lib()
    d = _F32(sdf.distances)
    o = _opts(d.dev)
    p = os.fsencode(path)
    if isinstance(sdf, SerializeGrid):
        rc = L.m2s_sdf_save_grid(p, C.byref(sdf.grid._g), d.ptr, d.n, C.byref(o))
    else:
        q = _F32(sdf.query_points, 3)
        if (q.dev is None) != (d.dev is None):
            raise ValueError("query_points and distances must live on the same side (both host or both CUDA)")
        rc = L.m2s_sdf_save_generic(p, q.ptr, q.n, d.ptr, d.n, C.byref(o))
    if rc != 0:
        _raise(rc)


def read_from_file(path, device=None):
    """serde.rs:216-220.  device=None -> numpy arrays; a torch CUDA device -> CUDA tensors."""
    L = _lib.lib()
    p = os.fsencode(path)
    info = M2SSdfInfo()
    rc = L.m2s_sdf_probe_file(p, C.byref(info))
    if rc != 0:
        _raise(rc)
    nq, nd = int(info.n_queries), int(info.n_distances)
    if device is not None:
        import torch

        dist = torch.empty(nd, dtype=torch.float32, device=device)
        qp = torch.empty((nq, 3), dtype=torch.float32, device=device)
        o = _opts(dist)
        rc = L.m2s_sdf_read_file(p, qp.data_ptr(), dist.data_ptr(), C.byref(o))
    else:
        dist = np.empty(nd, np.float32)
        qp = np.empty((nq, 3), np.float32)
        o = _opts(None)
        rc = L.m2s_sdf_read_file(p, qp.ctypes.data, dist.ctypes.data, C.byref(o))
    if rc != 0:
        _raise(rc)
    if info.kind == 1:
        return DeserializeGrid(_info_to_grid(info), dist)
    return DeserializeGeneric(qp, dist)
