"""The device math header (mesh_to_sdf_amd/csrc/geo.hip.h) compiled for the HOST by hipcc and
compared with the oracle: catches logic slips in the select-style restatement without a GPU.
(The gfx950 code generation itself is what the `-m gpu` parity tests cover.)"""
import ctypes as C
import os

import numpy as np
import pytest

import oracle as orc
from mesh_to_sdf_amd import _lib

F = np.float32
PROBE = os.path.join(os.path.dirname(_lib.SO_PATH), "libm2s_probe.so")


@pytest.fixture(scope="module")
def probe():
    if not os.path.exists(PROBE):
        _lib.build()
    L = C.CDLL(PROBE)
    L.probe_dist2.restype = C.c_float
    L.probe_dist2_signed.restype = C.c_float
    L.probe_normal_fold_result.restype = C.c_float
    L.probe_normal_fold_result.argtypes = [C.c_float, C.c_float]
    L.probe_approx_eq_abs.argtypes = [C.c_float, C.c_float]
    return L


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _cases(n, seed):
    rng = np.random.default_rng(seed)
    for i in range(n):
        p, a, b, c = (np.ascontiguousarray(rng.uniform(-10, 10, 3).astype(F)) for _ in range(4))
        k = i % 10
        if k == 6:
            b = a.copy()
        elif k == 7:
            c = b.copy()
        elif k == 8:
            c = a.copy()
        elif k == 9:
            b = a.copy()
            c = a.copy()
        elif k == 5:  # point on an edge line: exercises the <= / >= region tests
            t = rng.random()
            p = np.ascontiguousarray((a * F(1 - t) + b * F(t)).astype(F))
        yield p, a, b, c


def test_closest_point_distance_bits(probe):
    for p, a, b, c in _cases(4000, 1):
        d2 = F(probe.probe_dist2(_p(p), _p(a), _p(b), _p(c)))
        assert d2.view(np.uint32) == orc.point_triangle_distance2(p, a, b, c).view(np.uint32)
        assert np.sqrt(d2).view(np.uint32) == orc.point_triangle_distance(p, a, b, c).view(np.uint32)
        pos = C.c_int(0)
        d2s = F(probe.probe_dist2_signed(_p(p), _p(a), _p(b), _p(c), C.byref(pos)))
        sd = orc.point_triangle_signed_distance(p, a, b, c)
        assert d2s.view(np.uint32) == d2.view(np.uint32)
        assert (pos.value == 1) == (not np.signbit(sd)), (p, a, b, c, sd)


def test_ray_bits(probe):
    for p, a, b, c in _cases(3000, 2):
        for axis in range(3):
            t = C.c_float(0)
            hit = probe.probe_ray(axis, _p(p), _p(a), _p(b), _p(c), C.byref(t))
            want = orc.ray_triangle_intersection_aligned(p, a, b, c, axis)
            assert bool(hit) == (want is not None)
            if want is not None:
                assert F(t.value).view(np.uint32) == want.view(np.uint32)


def test_triangle_box(probe):
    for p, a, b, c in _cases(500, 3):
        mn, mx = np.zeros(3, F), np.zeros(3, F)
        probe.probe_tri_box(_p(a), _p(b), _p(c), _p(mn), _p(mx))
        omn, omx = orc.triangle_bounding_box(a, b, c)
        assert np.array_equal(mn, omn) and np.array_equal(mx, omx)


def test_normal_fold_result_equals_literal_fold(probe):
    """normal_fold_result(min d2, min positive d2) == the literal compare_distances fold
    (lib.rs:242-259 applied as in generic/default.rs:52-59)."""
    rng = np.random.default_rng(4)
    fmax = np.finfo(F).max
    checked = 0
    for _ in range(4000):
        n = int(rng.integers(1, 7))
        base = F(rng.uniform(0.0, 5.0)) if rng.random() < 0.8 else F(rng.uniform(0, 2000))
        mags = []
        for _k in range(n):
            if rng.random() < 0.3:
                mags.append(np.nextafter(base, F(1e9)))
            else:
                mags.append(F(base + F(rng.choice([0, 3e-7, 9e-7, 1.5e-6, 1e-3, 0.5]))))
        d = [m if rng.random() < 0.5 else F(-m) for m in mags]
        m = fmax
        for x in d:
            if orc.compare_distances(x, m) == -1:
                m = x
        amin = min(abs(x) for x in d)
        pmin = min([x for x in d if not np.signbit(x)], default=None)
        # feed squares whose correctly rounded sqrt returns the distances themselves
        sq_all = F(np.float64(amin) ** 2)
        sq_pos = F(np.float64(pmin) ** 2) if pmin is not None else F(np.inf)
        if np.sqrt(sq_all) != amin or (pmin is not None and np.sqrt(sq_pos) != pmin):
            continue
        got = F(probe.probe_normal_fold_result(float(sq_all), float(sq_pos)))
        assert got == m, (d, got, m)
        checked += 1
    assert checked > 1000
    assert F(probe.probe_normal_fold_result(float("inf"), float("inf"))) == fmax
