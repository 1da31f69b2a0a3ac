"""Host-side mirror of the reference client's steps either side of the generator, over the C ABI.

  merge_instances          mesh_to_sdf_client/src/sdf_program.rs:597-641 (load_gltf's merge + bounding box)
  order_cells_by_distance  mesh_to_sdf_client/src/sdf.rs:62-72, :120     (voxel order + iso limits)
  Sdf.new                  mesh_to_sdf_client/src/sdf.rs:32-137          (generate, order, limits — without wgpu)

numpy in -> numpy out (host pointers, staged inside the call); torch CUDA tensors in -> CUDA tensors out.
"""
import ctypes as C
import time
from dataclasses import dataclass

import numpy as np

from . import _lib
from ._lib import M2SGltfInfo, M2SInstance, M2SOpts
from .api import Grid, M2SError, SignMethod, Topology, _is_torch, generate_grid_sdf
from .serde import _opts


def _raise(rc):
    raise M2SError(rc, _lib.last_error())


def order_cells_by_distance(distances, want_limits=True):
    """-> (ordered_indices u32[n], (min, max) or None).  Stable ascending f32::total_cmp order."""
    L = _lib.lib()
    lim = (C.c_float * 2)() if want_limits else None
    if _is_torch(distances) and distances.is_cuda:
        import torch

        d = distances.detach().reshape(-1).to(torch.float32).contiguous()
        # torch has no uint32 arithmetic; the bits are u32 indices
        out = torch.empty(d.numel(), dtype=torch.int32, device=d.device)
        o = _opts(d)
        rc = L.m2s_order_cells_by_distance(d.data_ptr(), d.numel(), out.data_ptr(), lim, C.byref(o))
    else:
        d = np.ascontiguousarray(np.asarray(distances, np.float32).reshape(-1))
        out = np.empty(d.size, np.uint32)
        o = _opts(None)
        rc = L.m2s_order_cells_by_distance(d.ctypes.data, d.size, out.ctypes.data, lim, C.byref(o))
    if rc != 0:
        _raise(rc)
    return out, ((np.float32(lim[0]), np.float32(lim[1])) if want_limits else None)


def merge_instances(instances, want_bbox=True):
    """instances: iterable of (vertices, indices, transform).  vertices (N,3) f32 — or (N,K>=3) f32 rows whose
    first three columns are the position (the client's Vertex struct); transform: 16 f32, glam Mat4 column-major
    (`to_cols_array`).  -> (vertices (sum N,3), indices u32, bbox[6] or None)."""
    L = _lib.lib()
    instances = list(instances)
    device = bool(instances) and _is_torch(instances[0][0]) and instances[0][0].is_cuda
    table = (M2SInstance * max(1, len(instances)))()
    keep, nv, ni = [], 0, 0
    for k, (v, i, m) in enumerate(instances):
        if device:
            import torch

            v = v.detach().to(torch.float32)
            if v.stride(-1) != 1:
                v = v.contiguous()
            i = i.detach().contiguous()
            assert i.dtype in (torch.int32, torch.uint32), "indices must be 32-bit"
            vp, stride, n_v, ip, n_i = v.data_ptr(), v.stride(0) * 4 if v.shape[0] > 1 else 12, v.shape[0], i.data_ptr(), i.numel()
        else:
            v = np.asarray(v, np.float32)
            if v.ndim != 2 or v.shape[1] < 3 or (v.size and v.strides[1] != 4):
                v = np.ascontiguousarray(v.reshape(-1, 3))
            i = np.ascontiguousarray(np.asarray(i, np.uint32).reshape(-1))
            vp, stride, n_v, ip, n_i = v.ctypes.data, (v.strides[0] if v.shape[0] > 1 else 12), v.shape[0], i.ctypes.data, i.size
        keep += [v, i]
        t = table[k]
        t.vertices, t.n_vertices, t.vertex_stride, t.indices, t.n_indices = vp, n_v, stride, ip, n_i
        mm = np.asarray(m, np.float32).reshape(-1)
        assert mm.size == 16
        for j in range(16):
            t.transform[j] = float(mm[j])
        nv += n_v
        ni += n_i
    bbox = (C.c_float * 6)() if want_bbox else None
    if device:
        import torch

        dev = instances[0][0].device
        vo = torch.empty((nv, 3), dtype=torch.float32, device=dev)
        io = torch.empty(ni, dtype=torch.int32, device=dev)
        o = _opts(vo)
        rc = L.m2s_merge_instances(table, len(instances), vo.data_ptr(), io.data_ptr(), bbox, C.byref(o))
    else:
        vo = np.empty((nv, 3), np.float32)
        io = np.empty(ni, np.uint32)
        o = _opts(None)
        rc = L.m2s_merge_instances(table, len(instances), vo.ctypes.data, io.ctypes.data, bbox, C.byref(o))
    if rc != 0:
        _raise(rc)
    return vo, io, (np.array(list(bbox), np.float32) if want_bbox else None)


class GltfFile:
    """The models and instances of a glTF / GLB file (mesh_to_sdf_client/src/gltf/mod.rs:56-89 `load_scene`,
    without the wgpu resources).  instances(): [(vertices (N,3) f32, indices u32, transform (16,) f32)] as numpy
    views into the native handle — keep the GltfFile alive while they are used."""

    def __init__(self, path):
        import os

        self._h = C.c_void_p()
        self.info = M2SGltfInfo()
        rc = _lib.lib().m2s_gltf_open(os.fsencode(path), C.byref(self._h), C.byref(self.info))
        if rc != 0:
            _raise(rc)

    def instances(self):
        n = int(self.info.n_instances)
        table = (M2SInstance * max(n, 1))()
        rc = _lib.lib().m2s_gltf_instances(self._h, table, n)
        if rc != 0:
            _raise(rc)
        out = []
        for k in range(n):
            t = table[k]
            v = np.ctypeslib.as_array(C.cast(t.vertices, C.POINTER(C.c_float)), (t.n_vertices, 3)) if t.n_vertices else np.zeros((0, 3), np.float32)
            i = np.ctypeslib.as_array(C.cast(t.indices, C.POINTER(C.c_uint32)), (t.n_indices,)) if t.n_indices else np.zeros(0, np.uint32)
            out.append((v, i, np.array(list(t.transform), np.float32)))
        return out

    def close(self):
        if self._h:
            _lib.lib().m2s_gltf_close(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


def load_gltf(path):
    """mesh_to_sdf_client/src/sdf_program.rs:597-641 `load_gltf`: file -> merged (vertices, indices, bbox[6])."""
    with GltfFile(path) as g:
        inst = g.instances()
        if not inst:
            raise M2SError(_lib.ERR_BAD_ARG, "Bounding box is ill-defined")   # sdf_program.rs:624-626
        return merge_instances(inst)


@dataclass
class Sdf:
    """mesh_to_sdf_client/src/sdf.rs:12-31 without the wgpu buffers."""
    data: object
    ordered_indices: object
    grid: Grid
    iso_limits: tuple
    time_taken: float

    @staticmethod
    def new(vertices, indices, start_cell, end_cell, cell_count, sign_method: SignMethod = SignMethod.Raycast):
        grid = Grid.from_bounding_box(start_cell, end_cell, [int(c) for c in cell_count])   # sdf.rs:47
        t0 = time.perf_counter()
        data = generate_grid_sdf(vertices, Topology.TriangleList(indices), grid, sign_method)   # sdf.rs:50-55
        if _is_torch(data):
            import torch

            torch.cuda.synchronize(data.device)
        dt = time.perf_counter() - t0
        ordered, limits = order_cells_by_distance(data)   # sdf.rs:62-68, :120
        return Sdf(data, ordered, grid, limits, dt)

    def get_cell_count(self):   # sdf.rs:139-141
        c = self.grid.get_cell_count()
        return c[0] * c[1] * c[2]
