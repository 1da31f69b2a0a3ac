"""CPU restatement of the reference client's steps either side of generate_grid_sdf — TEST INFRASTRUCTURE ONLY.

  order_cells      mesh_to_sdf_client/src/sdf.rs:62-68   (0..n).sorted_by(|i,j| data[i].total_cmp(&data[j])) as u32
                   itertools::sorted_by = Vec::sort_by = stable merge sort; f32::total_cmp = IEEE totalOrder
                   (core::f32: compares the bits as i32 after `bits ^= ((bits >> 31) as u32 >> 1) as i32`).
  minmax           mesh_to_sdf_client/src/sdf.rs:120     data.iter().copied().minmax() — itertools 0.13
                   `minmax_impl` restated literally (pairwise, `<` from PartialOrd): first of equal minima,
                   last of equal maxima.
  merge_instances  mesh_to_sdf_client/src/sdf_program.rs:607-621  vertices transformed by the instance matrix
                   (glam Mat4::transform_point3, scalar path: ((x_axis*x + y_axis*y) + z_axis*z) + w_axis, no FMA),
                   indices offset by the running vertex count; then per-axis minmax for the bounding box (:624-632).
PINNING: no known-answer test exists in the reference for these client steps ("parity unpinned" against
reference outputs); the restatements are cross-checked against independent formulations in
tests/test_client_cpu.py (python `sorted` with a literal total_cmp comparator; float64 matrix product bounds).
"""
import functools

import numpy as np


def total_order_key(a):
    b = np.ascontiguousarray(a, np.float32).view(np.uint32)
    mask = np.where(b >> 31 == 1, np.uint32(0xFFFFFFFF), np.uint32(0x80000000))
    return b ^ mask


def order_cells(data):
    return np.argsort(total_order_key(data), kind="stable").astype(np.uint32)


def total_cmp(a, b):
    """core::f32::total_cmp, literally."""
    l = int(np.float32(a).view(np.int32))
    r = int(np.float32(b).view(np.int32))
    l ^= ((l >> 31) & 0xFFFFFFFF) >> 1
    r ^= ((r >> 31) & 0xFFFFFFFF) >> 1
    return (l > r) - (l < r)


def order_cells_literal(data):
    data = np.asarray(data, np.float32)
    idx = sorted(range(data.size), key=functools.cmp_to_key(lambda i, j: total_cmp(data[i], data[j])))  # stable
    return np.array(idx, np.uint32)


def minmax(data):
    """itertools minmax_impl with lt = `<`; returns None for an empty input (MinMaxResult::NoElements)."""
    it = iter(np.asarray(data, np.float32).tolist())
    try:
        first = next(it)
    except StopIteration:
        return None
    try:
        second = next(it)
    except StopIteration:
        return np.float32(first), np.float32(first)
    mn, mx = (second, first) if second < first else (first, second)
    while True:
        try:
            a = next(it)
        except StopIteration:
            break
        try:
            b = next(it)
        except StopIteration:
            if a < mn:
                mn = a
            elif not (a < mx):
                mx = a
            break
        if not (b < a):
            if a < mn:
                mn = a
            if not (b < mx):
                mx = b
        else:
            if b < mn:
                mn = b
            if not (a < mx):
                mx = a
    return np.float32(mn), np.float32(mx)


def transform_point3(m, v):
    """glam Mat4::transform_point3 (scalar): m is column-major 4x4 f32 (cols x_axis..w_axis), v (N,3) f32."""
    m = np.asarray(m, np.float32).reshape(4, 4)   # m[c] = column c
    v = np.asarray(v, np.float32).reshape(-1, 3)
    F = np.float32
    res = m[0][None, :] * v[:, 0:1].astype(F)
    res = (m[1][None, :] * v[:, 1:2]).astype(F) + res
    res = (m[2][None, :] * v[:, 2:3]).astype(F) + res
    res = m[3][None, :] + res
    return res[:, :3].astype(F)


def merge_instances(instances):
    """instances: list of (vertices (N,3) f32, indices u32, mat4 column-major) -> (vertices, indices, bbox[6])."""
    vs, ids, n = [], [], 0
    for v, i, m in instances:
        tv = transform_point3(m, v)
        vs.append(tv)
        ids.append((np.asarray(i, np.uint32) + np.uint32(n)).astype(np.uint32))
        n += tv.shape[0]
    v = np.concatenate(vs) if vs else np.zeros((0, 3), np.float32)
    i = np.concatenate(ids) if ids else np.zeros(0, np.uint32)
    bbox = None
    if v.shape[0] >= 1:
        mm = [minmax(v[:, k]) for k in range(3)]
        bbox = np.array([mm[0][0], mm[1][0], mm[2][0], mm[0][1], mm[1][1], mm[2][1]], np.float32)
    return v, i, bbox
