"""The oracle's restatement of the client steps (oracle/client_oracle.py) cross-checked against independent
formulations — there is no reference known-answer test for them — plus ABI argument checks (no GPU)."""
import ctypes as C

import numpy as np

from mesh_to_sdf_amd import _lib
from oracle import client_oracle as co

F = np.float32


def nasty(n, seed):
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32).view(F).copy()
    a[: min(n, 8)] = np.array([0.0, -0.0, np.inf, -np.inf, 1e-45, -1e-45, 0.0, -0.0], F)[: min(n, 8)]
    return a


def test_order_cells_matches_a_literal_total_cmp_stable_sort():
    for seed in range(4):
        d = nasty(700, seed)
        d[100:200] = d[300:400]          # ties: stability decides
        d[500:520] = np.float32("nan")
        d[520:540] = -np.float32("nan")
        assert np.array_equal(co.order_cells(d), co.order_cells_literal(d))
    # -NaN < -inf < ... < -0 < +0 < ... < inf < NaN  (IEEE totalOrder)
    d = np.array([np.nan, np.inf, 1.0, 0.0, -0.0, -1.0, -np.inf, -np.nan], F)
    assert list(co.order_cells(d)) == [7, 6, 5, 4, 3, 2, 1, 0]


def test_minmax_first_min_last_max():
    assert co.minmax([]) is None
    assert co.minmax([3.0]) == (3.0, 3.0)
    d = np.array([0.0, -0.0, 5.0, 5.0, -0.0, 0.0], F)
    mn, mx = co.minmax(d)
    assert not np.signbit(mn) and mn == 0 and mx == 5.0          # first of the equal minima is +0.0 (index 0)
    d = np.array([-0.0, 0.0, 2.0], F)
    assert np.signbit(co.minmax(d)[0])
    d = np.array([1.0, 7.0, -0.0, 7.0, 0.0], F)                   # odd length: the trailing single element path
    mn, mx = co.minmax(d)
    assert np.signbit(mn) and mx == 7.0
    rng = np.random.default_rng(3)
    for n in (2, 3, 10, 11, 1000, 1001):
        d = rng.standard_normal(n).astype(F)
        assert co.minmax(d) == (d.min(), d.max())


def test_transform_point3_matches_float64_within_rounding_and_is_exact_for_integers():
    rng = np.random.default_rng(11)
    m = rng.standard_normal(16).astype(F)
    m[3::4] = [0, 0, 0, 1]      # affine: bottom row (0,0,0,1) in column-major
    v = rng.standard_normal((500, 3)).astype(F)
    got = co.transform_point3(m, v)
    M = m.reshape(4, 4).astype(np.float64)   # M[c] = column c
    want = v.astype(np.float64) @ M[:3, :3] + M[3, :3]
    assert np.max(np.abs(got - want)) < 1e-5
    mi = np.array([2, 0, 0, 0, 0, 3, 0, 0, 0, 0, 4, 0, 10, 20, 30, 1], F)
    vi = np.array([[1, 2, 3], [-1, 0, 5]], F)
    assert np.array_equal(co.transform_point3(mi, vi), np.array([[12, 26, 42], [8, 20, 50]], F))


def test_merge_instances_offsets_indices_by_the_running_vertex_count():
    ident = np.eye(4, dtype=F).reshape(-1)
    shift = ident.copy(); shift[12:15] = [10, 0, 0]
    a = (np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], F), np.array([0, 1, 2], np.uint32), ident)
    b = (np.array([[0, 0, 1], [1, 0, 1], [0, 1, 1], [5, 5, 5]], F), np.array([0, 1, 2, 2, 1, 3], np.uint32), shift)
    v, i, bbox = co.merge_instances([a, b, a])
    assert v.shape == (10, 3) and list(i) == [0, 1, 2, 3, 4, 5, 5, 4, 6, 7, 8, 9]
    assert np.array_equal(v[3:7], b[0] + np.array([10, 0, 0], F))
    assert list(bbox) == [0, 0, 0, 15, 5, 5]


def test_abi_argument_errors_without_a_gpu():
    L = _lib.lib()
    d = np.zeros(4, F)
    out = np.zeros(4, np.uint32)
    assert L.m2s_order_cells_by_distance(None, 4, out.ctypes.data, None, None) == _lib.ERR_BAD_ARG
    assert L.m2s_order_cells_by_distance(d.ctypes.data, 2**32, out.ctypes.data, None, None) == _lib.ERR_BAD_ARG
    assert "u32" in _lib.last_error()
    assert L.m2s_merge_instances(None, 1, None, None, None, None) == _lib.ERR_BAD_ARG
    inst = (_lib.M2SInstance * 1)()
    inst[0].vertices, inst[0].n_vertices, inst[0].vertex_stride = d.ctypes.data, 1, 10
    assert L.m2s_merge_instances(inst, 1, d.ctypes.data, out.ctypes.data, None, None) == _lib.ERR_BAD_ARG
    assert "vertex_stride" in _lib.last_error()
    # valid arguments, no GPU here: loud failure, never a CPU fallback
    if L.m2s_device_count() == 0:
        assert L.m2s_order_cells_by_distance(d.ctypes.data, 4, out.ctypes.data, None, None) == _lib.ERR_HIP
        assert "no CPU fallback" in _lib.last_error()
