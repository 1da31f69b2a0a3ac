"""ctypes loader for the CPU oracle (oracle/m2s_oracle.cpp).

TEST INFRASTRUCTURE ONLY.  Importers allowed: tests/, __graft_entry__.smoke(), and the
`cpu_baseline` leg of bench.py.  The product package `mesh_to_sdf_amd` never imports this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libm2s_oracle.so")

_f3 = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_u64p = np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS")


def build(force=False):
    src = os.path.join(_HERE, "m2s_oracle.cpp")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s", "libm2s_oracle.so"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.orc_point_triangle_distance.restype = C.c_float
        L.orc_point_triangle_distance2.restype = C.c_float
        L.orc_point_triangle_signed_distance.restype = C.c_float
        L.orc_length.restype = C.c_float
        L.orc_dist.restype = C.c_float
        L.orc_dot.restype = C.c_float
        L.orc_compare_distances.argtypes = [C.c_float, C.c_float]
        L.orc_approx_eq.argtypes = [C.c_float, C.c_float, C.c_int, C.c_float]
        L.orc_grid_cell_idx.restype = C.c_uint64
        L.orc_get_triangles.restype = C.c_int64
        L.orc_generate_sdf.restype = C.c_int64
        L.orc_generate_sdf_fast.restype = C.c_int64
        _lib = L
    return _lib


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


# ---- primitives ---------------------------------------------------------------------
def closest_point_triangle(p, a, b, c):
    out = np.zeros(3, np.float32)
    lib().orc_closest_point_triangle(_p(_f(p)), _p(_f(a)), _p(_f(b)), _p(_f(c)), _p(out))
    return out


def closest_point_segment(p, a, b):
    out = np.zeros(3, np.float32)
    lib().orc_closest_point_segment(_p(_f(p)), _p(_f(a)), _p(_f(b)), _p(out))
    return out


def point_triangle_distance(p, a, b, c):
    return np.float32(lib().orc_point_triangle_distance(_p(_f(p)), _p(_f(a)), _p(_f(b)), _p(_f(c))))


def point_triangle_distance2(p, a, b, c):
    return np.float32(lib().orc_point_triangle_distance2(_p(_f(p)), _p(_f(a)), _p(_f(b)), _p(_f(c))))


def point_triangle_signed_distance(p, a, b, c):
    return np.float32(lib().orc_point_triangle_signed_distance(_p(_f(p)), _p(_f(a)), _p(_f(b)), _p(_f(c))))


def ray_triangle_intersection_aligned(o, a, b, c, axis):
    t = C.c_float(0.0)
    hit = lib().orc_ray_triangle_intersection_aligned(_p(_f(o)), _p(_f(a)), _p(_f(b)), _p(_f(c)), int(axis), C.byref(t))
    return np.float32(t.value) if hit else None


def triangle_bounding_box(a, b, c):
    mn, mx = np.zeros(3, np.float32), np.zeros(3, np.float32)
    lib().orc_triangle_bounding_box(_p(_f(a)), _p(_f(b)), _p(_f(c)), _p(mn), _p(mx))
    return mn, mx


def compare_distances(a, b):
    return lib().orc_compare_distances(float(np.float32(a)), float(np.float32(b)))


def approx_eq(a, b, ulps=2, eps=1e-6):
    return bool(lib().orc_approx_eq(float(np.float32(a)), float(np.float32(b)), int(ulps), float(np.float32(eps))))


def length(a):
    return np.float32(lib().orc_length(_p(_f(a))))


def dist(a, b):
    return np.float32(lib().orc_dist(_p(_f(a)), _p(_f(b))))


# ---- grid ---------------------------------------------------------------------------
def _cnt(count):
    return np.ascontiguousarray(count, dtype=np.uint64)


def grid_from_bounding_box(bmin, bmax, count):
    first, size = np.zeros(3, np.float32), np.zeros(3, np.float32)
    lib().orc_grid_from_bounding_box(_p(_f(bmin)), _p(_f(bmax)), _p(_cnt(count)), _p(first), _p(size))
    return first, size, tuple(int(c) for c in count)


def grid_bounding_box(first, size, count):
    mn, mx = np.zeros(3, np.float32), np.zeros(3, np.float32)
    lib().orc_grid_bounding_box(_p(_f(first)), _p(_f(size)), _p(_cnt(count)), _p(mn), _p(mx))
    return mn, mx


def grid_cell_idx(count, cell):
    return int(lib().orc_grid_cell_idx(_p(_cnt(count)), _p(_cnt(cell))))


def grid_cell_coords(count, idx):
    out = np.zeros(3, np.uint64)
    lib().orc_grid_cell_coords(_p(_cnt(count)), C.c_uint64(idx), _p(out))
    return [int(v) for v in out]


def grid_cell_center(first, size, count, cell):
    out = np.zeros(3, np.float32)
    lib().orc_grid_cell_center(_p(_f(first)), _p(_f(size)), _p(_cnt(count)), _p(_cnt(cell)), _p(out))
    return out


def grid_snap(first, size, count, p):
    out = np.zeros(3, np.uint64)
    inside = lib().orc_grid_snap(_p(_f(first)), _p(_f(size)), _p(_cnt(count)), _p(_f(p)), _p(out))
    return bool(inside), [int(v) for v in out]


# ---- whole path ---------------------------------------------------------------------
ACCEL = {"None": 0, "Bvh": 1, "Rtree": 2, "RtreeBvh": 3}
SIGN = {"Raycast": 0, "Normal": 1}
TOPO = {"TriangleList": 0, "TriangleStrip": 1}
EXACT, PROPAGATE, EXACT_BVH = 0, 1, 2


class OracleError(RuntimeError):
    pass


def _idx(indices):
    if indices is None:
        return None, 0
    a = np.ascontiguousarray(indices, dtype=np.uint32).reshape(-1)
    return a, a.size


def get_triangles(n_verts, indices, topology=0):
    idx, ni = _idx(indices)
    n = lib().orc_get_triangles(C.c_size_t(n_verts), _p(idx), C.c_size_t(ni), int(topology), None)
    if n < 0:
        raise OracleError(f"get_triangles rc={n}")
    out = np.zeros((n, 3), np.uint32)
    lib().orc_get_triangles(C.c_size_t(n_verts), _p(idx), C.c_size_t(ni), int(topology), _p(out))
    return out


def generate_sdf(vertices, indices, queries, accel=3, sign=0, topology=0, threads=0, fast=False):
    v = _f(vertices).reshape(-1, 3)
    q = _f(queries).reshape(-1, 3)
    idx, ni = _idx(indices)
    out = np.zeros(q.shape[0], np.float32)
    threads = threads or hardware_threads()
    fn = lib().orc_generate_sdf_fast if fast else lib().orc_generate_sdf
    n = fn(_p(v), C.c_size_t(v.shape[0]), _p(idx), C.c_size_t(ni), int(topology), _p(q),
                               C.c_size_t(q.shape[0]), int(accel), int(sign), int(threads), _p(out))
    if n < 0:
        raise OracleError(f"generate_sdf rc={n}")
    return out[:n]


def generate_grid_sdf(vertices, indices, first, size, count, sign=0, semantics=EXACT, topology=0, heaps=1, threads=0,
                      return_stats=False):
    v = _f(vertices).reshape(-1, 3)
    idx, ni = _idx(indices)
    cnt = _cnt(count)
    out = np.zeros(int(cnt[0]) * int(cnt[1]) * int(cnt[2]), np.float32)
    stats = np.zeros(3, np.uint64)
    threads = threads or (1 if semantics == PROPAGATE else hardware_threads())
    rc = lib().orc_generate_grid_sdf(_p(v), C.c_size_t(v.shape[0]), _p(idx), C.c_size_t(ni), int(topology), _p(_f(first)),
                                     _p(_f(size)), _p(cnt), int(sign), int(semantics), int(heaps), int(threads), _p(out),
                                     _p(stats))
    if rc < 0:
        raise OracleError(f"generate_grid_sdf rc={rc}")
    return (out, stats) if return_stats else out


def grid_ray_parity(vertices, indices, first, size, count, topology=0, threads=0):
    v = _f(vertices).reshape(-1, 3)
    idx, ni = _idx(indices)
    cnt = _cnt(count)
    out = np.zeros((int(cnt[0]) * int(cnt[1]) * int(cnt[2]), 3), np.uint8)
    rc = lib().orc_grid_ray_parity(_p(v), C.c_size_t(v.shape[0]), _p(idx), C.c_size_t(ni), int(topology), _p(_f(first)),
                                   _p(_f(size)), _p(cnt), int(threads or hardware_threads()), _p(out))
    if rc < 0:
        raise OracleError(f"grid_ray_parity rc={rc}")
    return out


def hardware_threads():
    return int(lib().orc_hardware_threads())
