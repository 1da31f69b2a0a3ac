"""The oracle's BVH-accelerated EXACT mode must give the brute-force answers bit for bit
(it is what the GPU parity tests compare against at 100k triangles).  CPU only."""
import numpy as np
import pytest

import oracle as orc
from mesh_to_sdf_amd import meshes

F = np.float32


def _bits(a):
    return np.ascontiguousarray(a, F).view(np.uint32)


def _rand_queries(v, n, seed, pad=0.3):
    rng = np.random.default_rng(seed)
    lo, hi = v.min(0), v.max(0)
    ext = hi - lo
    return (lo - pad * ext + rng.random((n, 3)) * (1 + 2 * pad) * ext).astype(F)


@pytest.mark.parametrize("accel,sign", [(0, 1), (1, 0), (1, 1), (2, 0), (3, 0)])
def test_fast_generic_equals_brute_suzanne(suzanne, accel, sign):
    v, idx = suzanne
    q = _rand_queries(v, 3000, 11)
    a = orc.generate_sdf(v, idx, q, accel=accel, sign=sign)
    b = orc.generate_sdf(v, idx, q, accel=accel, sign=sign, fast=True)
    assert np.array_equal(_bits(a), _bits(b))


@pytest.mark.parametrize("sign", [0, 1])
def test_fast_grid_equals_brute(suzanne, sign):
    v, idx = suzanne
    first, size, cnt = orc.grid_from_bounding_box(v.min(0) - 0.2, v.max(0) + 0.2, [24, 20, 28])
    a = orc.generate_grid_sdf(v, idx, first, size, cnt, sign=sign, semantics=orc.EXACT)
    b = orc.generate_grid_sdf(v, idx, first, size, cnt, sign=sign, semantics=orc.EXACT_BVH)
    assert np.array_equal(_bits(a), _bits(b))


def test_fast_equals_brute_blob_far_offset():
    # a translated mesh (coordinates ~ 1000) stresses the pruning slack
    v, idx = meshes.blob(slices=40, stacks=21)
    v = (v * F(3.0) + F(1000.0)).astype(F)
    q = _rand_queries(v, 2000, 5)
    for accel, sign in [(1, 0), (0, 1), (2, 0)]:
        a = orc.generate_sdf(v, idx, q, accel=accel, sign=sign)
        b = orc.generate_sdf(v, idx, q, accel=accel, sign=sign, fast=True)
        assert np.array_equal(_bits(a), _bits(b)), (accel, sign)


def test_normal_fold_closed_form(suzanne):
    """The compare_distances fold (lib.rs:242-259, default.rs:52-59) is order dependent in
    principle (non-transitive).  Its result nevertheless equals a closed form that the HIP
    kernels use: with dmin = min |d| and P = {d : d not sign-negative, approx_eq(|d|, dmin)},
    result = min(P) if P else -dmin.  Checked here against the literal fold, in index order
    and in shuffled orders, on points near edges/vertices where ties actually occur."""
    v, idx = suzanne
    tris = idx.reshape(-1, 3)
    rng = np.random.default_rng(3)
    # queries ON mesh edges and vertices pushed out a little: many exact/near ties
    e = tris[rng.integers(0, len(tris), 400)]
    t = rng.random((400, 1)).astype(F)
    on_edge = v[e[:, 0]] * (1 - t) + v[e[:, 1]] * t
    nrm = rng.normal(size=(400, 3)).astype(F)
    q = np.concatenate([on_edge + 0.05 * nrm, v[rng.integers(0, len(v), 200)] + 0.02 * rng.normal(size=(200, 3)).astype(F)]).astype(F)
    lit = orc.generate_sdf(v, idx, q, accel=0, sign=1)
    for i, p in enumerate(q):
        d = np.array([orc.point_triangle_signed_distance(p, v[a], v[b], v[c]) for a, b, c in tris], F)
        dmin = np.min(np.abs(d))
        P = [x for x in d if not np.signbit(x) and orc.approx_eq(abs(x), dmin)]
        closed = min(P) if P else -dmin
        assert closed == lit[i], (i, closed, lit[i])
        for _ in range(3):  # shuffled literal folds agree too
            m = np.finfo(F).max
            for x in d[rng.permutation(len(d))]:
                if orc.compare_distances(x, m) == -1:
                    m = x
            assert m == lit[i]
