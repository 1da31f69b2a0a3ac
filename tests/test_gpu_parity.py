"""Parity tests proper: the HIP path (through the C ABI) against the CPU oracle on the same inputs.

Bar (BASELINE.json north_star): f32 distances within 1e-5 absolute, sign bit-exact.  Against the
oracle's EXACT semantics the kernels are expected — and asserted — to be bit-identical; against
the reference's label-PROPAGATION semantics the documented <0.5 % of cells differ (SURVEY.md
header fact 2) and the test reports the histogram and asserts the one-sided property.
Needs a real MI355X: run with `-m gpu`.  Nothing here reads /root/reference.
"""
import os

import numpy as np
import pytest

import oracle as orc
from mesh_to_sdf_amd import _lib
from mesh_to_sdf_amd import (AccelerationMethod, Grid, M2SPanic, M2STimings, SignMethod, Topology, generate_grid_sdf,
                             generate_sdf, meshes)

pytestmark = pytest.mark.gpu
F = np.float32
TOL = 1e-5  # north_star tolerance (absolute, f32 distances); signs must match exactly


@pytest.fixture(autouse=True, params=["default", "cut lists on every grid", "lane walks", "split walks", "direct evaluations", "queued evaluations", "queued + direct evaluations",
                                      "packet groups"])
def cut_lists_mode(request):
    """The cut lists (k_cut) are used from 100 000 packets per launch upwards and the packet walk of generic queries from a
    few million queries; the second run of every test lowers the thresholds and forces the packet walks, so that the nasty
    small inputs below (degenerate, non-finite, huge, anisotropic, sliced, duplicated) meet packets and cut lists too; the third
    run forces the lane walks (k_lane, k_lane_q) on everything that is small enough for them to be quick.  In the default run
    the tiny grids (cells x triangles <= M2S_BRUTE_MAX) take the tree-less k_brute_split.  The packet walks queue their leaf
    pre-tests and their exact evaluations as (voxel, triangle) pairs and run them 64 at a time (distance.hip DeferQueue; what these
    small inputs take by default): the last three runs force the other forms — every reached triangle wave-wide at once (round 3's
    walk, and the one M2S_STATS counts), only the evaluations queued, and the mixed form of grids far finer than the mesh."""
    import os

    mode = request.param
    big = any(t in request.node.name for t in ("full_size", "10M", "512", "1024", "config", "256"))
    if mode != "default" and "full_size" in request.node.name:
        pytest.skip("large enough to use the cut lists anyway")
    if mode != "default" and ("four_million" in request.node.name or "far_field" in request.node.name):
        pytest.skip("the test switches the walks itself")
    if mode != "default" and "reference_propagation_parity" in request.node.name:
        pytest.skip("a report against minutes of CPU propagation: once is enough (the kernels' modes are covered by the other tests)")
    if mode == "lane walks" and big:
        pytest.skip("the lane walk is not meant for this size")
    if mode in ("direct evaluations", "queued evaluations", "queued + direct evaluations") and big and "512" not in request.node.name:
        pytest.skip("the evaluation forms differ per packet, not per size: the small inputs and one large grid cover them")
    if mode == "packet groups" and big and "256" not in request.node.name:
        pytest.skip("four waves per packet are for launches shallower than the chip; 256^3 stands for the large ones")
    if mode == "split walks" and big:
        pytest.skip("a budget of 24 work units is for small inputs (test_split_walk_* covers the large ones with realistic budgets)")
    # the library reads its knobs from the environment once; afterwards they are switched through m2s_tuning_set (_lib.set_knob)
    forced = {}
    if mode == "cut lists on every grid":
        # M2S_BRUTE_MAX=0: no brute-force shortcut for tiny problems (the default mode takes it): the walks must see them
        forced = {"M2S_CUT_MIN_PACKETS": 8, "M2S_QUERY_CUT_MIN": 1, "M2S_LANE_WALK": 0, "M2S_BRUTE_MAX": 0, "M2S_SPLIT": 0, "M2S_TREELETS": 0,   # (and a tree without the treelet pass)
                  "M2S_CUT_COARSE": 1}                                    # (and the lists made in two levels, as the large grids' are)
    elif mode == "lane walks":
        forced = {"M2S_LANE_WALK": 1, "M2S_BRUTE_MAX": 0, "M2S_LEAF_MAX": 3}
    elif mode == "split walks":
        # every packet that lasts longer than a few node tests hands the rest of its ranges to other waves (distance.hip, split walk)
        forced = {"M2S_SPLIT": 2, "M2S_SPLIT_BUDGET": 24, "M2S_SPLIT_MIN_RECORDS": 4, "M2S_SPLIT_MAX_RECORDS": 64, "M2S_SPLIT_ROUNDS": 3, "M2S_LANE_WALK": 0, "M2S_BRUTE_MAX": 0, "M2S_TREELETS": 1}
    elif mode == "direct evaluations":
        forced = {"M2S_DEFER": 0, "M2S_LANE_WALK": 0, "M2S_BRUTE_MAX": 0, "M2S_LEAF_MAX": 2}    # (leaves of 2: round 3's tree)
    elif mode == "queued evaluations":
        forced = {"M2S_DEFER": 1, "M2S_LANE_WALK": 0, "M2S_BRUTE_MAX": 0, "M2S_LEAF_MAX": 4}
    elif mode == "queued + direct evaluations":
        forced = {"M2S_DEFER": 2, "M2S_LANE_WALK": 0, "M2S_BRUTE_MAX": 0}
    elif mode == "packet groups":
        # every packet a workgroup of four waves that share their minima in LDS (k_packet_group; automatic for shallow launches over fine meshes)
        forced = {"M2S_GROUP": 1, "M2S_LANE_WALK": 0, "M2S_BRUTE_MAX": 0, "M2S_CUT_MIN_PACKETS": 4000000000}
    with _lib.knobs(**forced):
        yield


def bits(a):
    return np.ascontiguousarray(a, F).view(np.uint32)


def assert_bit_equal(got, want, what=""):
    got, want = np.asarray(got, F), np.asarray(want, F)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    bad = np.flatnonzero(bits(got) != bits(want))
    assert bad.size == 0, f"{what}: {bad.size}/{got.size} differ, first {bad[:5]}: got {got[bad[:5]]} want {want[bad[:5]]}"


def assert_within_tol(got, want, what=""):
    got, want = np.asarray(got, F), np.asarray(want, F)
    assert np.array_equal(np.signbit(got), np.signbit(want)), f"{what}: sign mismatch"
    assert np.max(np.abs(got - want)) <= TOL, what


def grid_of(v, count, frac=0.1):
    lo, hi = meshes.extended_bbox(v, frac)
    return Grid.from_bounding_box(lo, hi, count)


def oracle_grid(v, idx, grid, sign, semantics=orc.EXACT_BVH, topology=0, **kw):
    return orc.generate_grid_sdf(v, idx, grid.get_first_cell(), grid.get_cell_size(), grid.get_cell_count(), sign=int(sign),
                                 semantics=semantics, topology=topology, **kw)


ACCELS = [
    ("None(Raycast)", AccelerationMethod.None_(SignMethod.Raycast), 0, 0),
    ("None(Normal)", AccelerationMethod.None_(SignMethod.Normal), 0, 1),
    ("Bvh(Raycast)", AccelerationMethod.Bvh(SignMethod.Raycast), 1, 0),
    ("Bvh(Normal)", AccelerationMethod.Bvh(SignMethod.Normal), 1, 1),
    ("Rtree", AccelerationMethod.Rtree, 2, 1),
    ("RtreeBvh", AccelerationMethod.RtreeBvh, 3, 0),
]


# ---- the reference's own known answers, through the C ABI -------------------------------------
def test_doctest_known_answers():
    v = np.array([[0.5, 1.5, 0.5], [1.0, 2.0, 3.0], [1.0, 3.0, 7.0]], F)
    assert generate_sdf(v, Topology.TriangleList([0, 1, 2]), [[0.5, 0.5, 0.5]], AccelerationMethod.RtreeBvh).tolist() == [1.0]  # lib.rs:13-31
    v = np.array([[0.0, 1.0, 0.0], [1.0, 2.0, 3.0], [1.0, 3.0, 4.0]], F)
    assert generate_sdf(v, Topology.TriangleList([0, 1, 2]), [[0.0, 0.0, 0.0]]).tolist() == [1.0]  # lib.rs:269-289
    v = np.array([[0.5, 1.5, 0.5], [1.0, 2.0, 3.0], [1.0, 3.0, 4.0]], F)
    g = Grid.from_bounding_box([0, 0, 0], [10, 10, 10], [10, 10, 10])
    assert generate_grid_sdf(v, Topology.TriangleList([0, 1, 2]), g, SignMethod.Raycast)[0] == 1.0  # generate/grid.rs:207-231


@pytest.mark.parametrize("algorithm", [0, 1])
def test_generate_grid_equals_generate_sdf(algorithm):
    # generate/grid.rs:693-724, assert_eq! on all 125 cells
    v = np.array([[0.0, 1.0, 0.0], [1.0, 2.0, 3.0], [1.0, 3.0, 4.0], [2.0, 0.0, 0.0]], F)
    idx = [0, 1, 2, 1, 2, 3]
    g = Grid.from_bounding_box([0, 0, 0], [5, 5, 5], [5, 5, 5])
    q = np.array([g.get_cell_center([x, y, z]) for x in range(5) for y in range(5) for z in range(5)])
    sdf = generate_sdf(v, Topology.TriangleList(idx), q, AccelerationMethod.None_(SignMethod.Raycast))
    grid_sdf = generate_grid_sdf(v, Topology.TriangleList(idx), g, SignMethod.Raycast, algorithm=algorithm)
    assert_bit_equal(grid_sdf, sdf, "grid vs generate_sdf")
    assert_bit_equal(grid_sdf, orc.generate_sdf(v, idx, q, accel=0, sign=0), "grid vs oracle")


def test_suzanne_query_table(suzanne):
    # generic/bvh.rs:158-164 etc.; expected values from SURVEY.md appendix A, pinned in test_oracle_kat
    v, idx = suzanne
    q = np.array([[0.01, 0.01, 0.5], [1.0, 1.0, 1.0], [0.1, 0.2, 0.2], [1.1, 2.2, 5.2], [-0.1, 0.2, -0.2], [0.0, 0.0, 0.0]], F)
    for name, am, accel, sign in ACCELS:
        assert_bit_equal(generate_sdf(v, Topology.TriangleList(idx), q, am), orc.generate_sdf(v, idx, q, accel=accel, sign=sign), name)


# ---- generic path, every back-end, BVH kernels and brute-force kernels -------------------------
@pytest.mark.parametrize("name,am,accel,sign", ACCELS, ids=[a[0] for a in ACCELS])
@pytest.mark.parametrize("algorithm", [0, 1])
def test_generic_suzanne(suzanne, name, am, accel, sign, algorithm):
    v, idx = suzanne
    lo, hi = meshes.extended_bbox(v, 0.3)
    q = meshes.uniform_queries(lo, hi, 20000)
    got = generate_sdf(v, Topology.TriangleList(idx), q, am, algorithm=algorithm)
    want = orc.generate_sdf(v, idx, q, accel=accel, sign=sign, fast=(accel != 0 or sign != 0))
    assert_bit_equal(got, want, f"{name} alg={algorithm}")
    assert_within_tol(got, want, name)


def test_generic_u16_indices_and_strip(suzanne):
    v, idx = suzanne
    q = meshes.uniform_queries(*meshes.extended_bbox(v, 0.2), 4000)
    a = generate_sdf(v, Topology.TriangleList(idx.astype(np.uint16)), q)           # the asset's own index type
    b = generate_sdf(v, Topology.TriangleList(idx), q)
    assert_bit_equal(a, b, "u16 vs u32 indices")
    strip = idx[:300]
    got = generate_sdf(v, Topology.TriangleStrip(strip), q, AccelerationMethod.Bvh(SignMethod.Normal))
    assert_bit_equal(got, orc.generate_sdf(v, strip, q, accel=1, sign=1, topology=1), "strip")
    nv = v[idx]                                                                     # Topology::TriangleList(None)
    got = generate_sdf(nv, Topology.TriangleList(None), q)
    assert_bit_equal(got, b, "no indices")


def test_generic_blob100k_rtreebvh():
    # BASELINE config 3 at oracle-affordable size: 100k-tri mesh, RtreeBvh parity
    v, idx = meshes.named("blob-100k")
    q = meshes.uniform_queries(*meshes.extended_bbox(v, 0.1), 200_000)
    got = generate_sdf(v, Topology.TriangleList(idx), q, AccelerationMethod.RtreeBvh)
    want = orc.generate_sdf(v, idx, q, accel=3, fast=True)
    assert_bit_equal(got, want, "blob-100k RtreeBvh")
    got = generate_sdf(v, Topology.TriangleList(idx), q, AccelerationMethod.Rtree)
    assert_bit_equal(got, orc.generate_sdf(v, idx, q, accel=2, fast=True), "blob-100k Rtree")


def test_generic_duplicate_and_clustered_queries():
    """Packets of the generic path are prefix cells of the 30-bit Morton keys (k_qcells): more than 64 queries in one cell of the
    finest level (identical points; a tight cluster beside far outliers that stretch the query box) are cut by position."""
    v, idx = meshes.blob(60, 31)
    lo, hi = meshes.extended_bbox(v, 0.1)
    rng = np.random.default_rng(7)
    q = meshes.uniform_queries(lo, hi, 3000)
    same = np.repeat(q[:3], [700, 65, 129], axis=0)                                   # identical points
    cluster = (q[10] + rng.uniform(-1e-5, 1e-5, (5000, 3))).astype(F)                # one finest-level cell, distinct points
    far = np.array([[50.0, -40.0, 30.0], [-60.0, 55.0, -45.0]], F)                     # stretch the box: everything else shares few cells
    allq = np.concatenate([same, q, cluster, far, q[:500]]).astype(F)
    for name, am, accel, sign in ACCELS[2:]:
        got = generate_sdf(v, Topology.TriangleList(idx), allq, am)
        assert_bit_equal(got, orc.generate_sdf(v, idx, allq, accel=accel, sign=sign, fast=True), name)
    for n in (1, 2, 63, 64, 65, 129):                                                 # tiny sets: one packet, a window that does not exist
        got = generate_sdf(v, Topology.TriangleList(idx), allq[:n], AccelerationMethod.RtreeBvh)
        assert_bit_equal(got, orc.generate_sdf(v, idx, allq[:n], accel=3, fast=True), f"{n} queries")


def test_generic_nonfinite_queries_raycast():
    """A NaN / inf query coordinate: every distance is NaN and f32::min drops it (default.rs:47), no ray hits (geo.rs:203) — the
    reference returns +f32::MAX for it in the Raycast modes; the packets, their boxes and cut lists must survive such a member."""
    v, idx = meshes.blob(60, 31)
    q = meshes.uniform_queries(*meshes.extended_bbox(v, 0.1), 20_000)
    q[5] = [np.nan, 0.1, 0.2]
    q[6000] = [0.3, np.inf, 0.1]
    q[12345] = [0.2, 0.2, -np.inf]
    q[19999] = [np.nan, np.nan, np.nan]
    bad = [5, 6000, 12345, 19999]
    got = generate_sdf(v, Topology.TriangleList(idx), q, AccelerationMethod.Bvh(SignMethod.Raycast))
    assert_bit_equal(got, orc.generate_sdf(v, idx, q, accel=1, sign=0), "Bvh(Raycast)")
    assert (got[bad] == np.finfo(F).max).all()
    # RtreeBvh: the reference measures the distance to whichever triangle rstar returns for a NaN / inf point (unspecified; include/m2s.h):
    # only the finite queries are compared
    got = generate_sdf(v, Topology.TriangleList(idx), q, AccelerationMethod.RtreeBvh)
    want = orc.generate_sdf(v, idx, q, accel=3, fast=True)
    assert_bit_equal(np.delete(got, bad), np.delete(want, bad), "RtreeBvh, finite queries")


def test_generic_consecutive_packet_fallback():
    """More bucket cells than the launch has waves (M2S_QUERY_LAUNCH_TIGHT forces it): k_qtable_mode switches to 64 consecutive
    queries per packet on the device; same bits."""
    import os

    v, idx = meshes.blob(60, 31)
    q = meshes.uniform_queries(*meshes.extended_bbox(v, 0.1), 50_000)
    want = orc.generate_sdf(v, idx, q, accel=3, fast=True)
    with _lib.knobs(M2S_QUERY_LAUNCH_TIGHT=1):
        got = generate_sdf(v, Topology.TriangleList(idx), q, AccelerationMethod.RtreeBvh)
    assert_bit_equal(got, want, "consecutive-packet fallback")


def test_generic_large_coordinates():
    # mesh far from the origin: the pruning slack must scale with the coordinate magnitude
    v, idx = meshes.blob(60, 31)
    v = (v * F(2.0) + F(1000.0)).astype(F)
    q = meshes.uniform_queries(*meshes.extended_bbox(v, 0.3), 30000)
    for name, am, accel, sign in ACCELS[2:]:
        got = generate_sdf(v, Topology.TriangleList(idx), q, am)
        assert_bit_equal(got, orc.generate_sdf(v, idx, q, accel=accel, sign=sign, fast=True), name)


# ---- grid path ----------------------------------------------------------------------------------
@pytest.mark.parametrize("sign", [SignMethod.Raycast, SignMethod.Normal])
@pytest.mark.parametrize("algorithm", [0, 1])
def test_grid_suzanne_64(suzanne, sign, algorithm):
    # BASELINE config 1 shape: bundled asset, 64^3 (tight bbox as generic/bvh.rs:192-249 uses)
    v, idx = suzanne
    g = Grid.from_bounding_box(v.min(0), v.max(0), [64, 64, 64])
    got = generate_grid_sdf(v, Topology.TriangleList(idx), g, sign, algorithm=algorithm)
    want = oracle_grid(v, idx, g, sign)
    assert_bit_equal(got, want, f"suzanne 64^3 {sign.name}")


@pytest.mark.parametrize("sign", [SignMethod.Raycast, SignMethod.Normal])
def test_grid_ragged_dims(suzanne, sign):
    v, idx = suzanne
    g = grid_of(v, [37, 21, 50], 0.15)   # not multiples of the 4x4x4 brick, nz not a multiple of 32
    got = generate_grid_sdf(v, Topology.TriangleList(idx), g, sign)
    assert_bit_equal(got, oracle_grid(v, idx, g, sign), "ragged")
    g1 = grid_of(v, [1, 1, 1], 0.0)
    assert_bit_equal(generate_grid_sdf(v, Topology.TriangleList(idx), g1, sign), oracle_grid(v, idx, g1, sign), "1x1x1")


def test_grid_vs_reference_propagation(suzanne):
    """What a user switching from the reference sees: the reference's grid path propagates
    triangle labels (generate/grid.rs:495-558) and ends >= the exact minimum on a small fraction
    of cells; everywhere else the values are bit-identical."""
    v, idx = suzanne
    g = grid_of(v, [32, 32, 32], 0.2)
    got = generate_grid_sdf(v, Topology.TriangleList(idx), g, SignMethod.Raycast)
    prop = oracle_grid(v, idx, g, SignMethod.Raycast, semantics=orc.PROPAGATE)
    diff = np.abs(np.abs(got) - np.abs(prop))
    frac_within = np.mean(diff <= TOL)
    print(f"\n[propagation parity] within 1e-5: {100 * frac_within:.3f}%  identical bits: {100 * np.mean(bits(got) == bits(prop)):.3f}%  "
          f"max dev {diff.max():.3e}  sign mismatches {int(np.sum(np.signbit(got) != np.signbit(prop)))}")
    assert frac_within > 0.99
    assert np.all(np.abs(got) <= np.abs(prop))                       # exact minimum never exceeds the propagated value
    assert np.array_equal(np.signbit(got), np.signbit(prop))        # sign rule is independent of the magnitude path
    assert diff.max() < 0.01                                         # the tolerance the reference's own tests use


def test_grid_continuity_property():
    # generate/grid.rs:729-807 on a watertight synthetic mesh: |d - d_n| <= cell, sign flips only within a cell
    v, idx = meshes.blob(60, 41)
    g = grid_of(v, [32, 32, 32], 0.2)
    sdf = generate_grid_sdf(v, Topology.TriangleList(idx), g, SignMethod.Raycast).reshape(32, 32, 32)
    cs = g.get_cell_size()
    for ax in range(3):
        a = np.take(sdf, range(0, 31), axis=ax)
        b = np.take(sdf, range(1, 32), axis=ax)
        assert np.all(np.abs(np.abs(a) - np.abs(b)) <= cs[ax] * (1 + 1e-5))
        flip = np.signbit(a) != np.signbit(b)
        assert np.all(np.abs(a[flip]) <= cs[ax]) and np.all(np.abs(b[flip]) <= cs[ax])
    assert (sdf < 0).any() and (sdf > 0).any()


def test_grid_smaller_than_mesh_raycast(suzanne):
    # generate/grid.rs:811-843: grid that does not contain the mesh must not index out of bounds
    v, idx = suzanne
    g = Grid.from_bounding_box(v.min(0), v.max(0) * F(0.5), [32, 32, 32])
    got = generate_grid_sdf(v, Topology.TriangleList(idx), g, SignMethod.Raycast)
    assert_bit_equal(got, oracle_grid(v, idx, g, SignMethod.Raycast), "grid smaller than mesh")


def test_grid_negative_and_anisotropic_cell_size(suzanne):
    v, idx = suzanne
    g = Grid.new([1.2, -0.9, 0.8], [-0.05, 0.04, -0.03], [48, 44, 52])   # "cell_size can be ... even negative" (grid.rs:24)
    for sign in (SignMethod.Raycast, SignMethod.Normal):
        assert_bit_equal(generate_grid_sdf(v, Topology.TriangleList(idx), g, sign), oracle_grid(v, idx, g, sign), f"negative cells {sign.name}")


def test_topology_variants_grid():
    # generate/grid.rs:847-904
    g = Grid.from_bounding_box([0, 0, 0], [5, 5, 5], [25, 25, 25])
    v0, v1, v2, v3 = [0.0, 1.0, 0.0], [1.0, 2.0, 3.0], [1.0, 3.0, 4.0], [2.0, 0.0, 0.0]
    a = generate_grid_sdf(np.array([v0, v1, v2, v3], F), Topology.TriangleList([0, 1, 2, 1, 2, 3, 2, 3, 0]), g, SignMethod.Normal)
    b = generate_grid_sdf(np.array([v0, v1, v2, v1, v2, v3, v2, v3, v0], F), Topology.TriangleList(None), g, SignMethod.Normal)
    c = generate_grid_sdf(np.array([v0, v1, v2, v3], F), Topology.TriangleStrip([0, 1, 2, 3, 0]), g, SignMethod.Normal)
    d = generate_grid_sdf(np.array([v0, v1, v2, v3, v0], F), Topology.TriangleStrip(None), g, SignMethod.Normal)
    want = orc.generate_grid_sdf([v0, v1, v2, v3], [0, 1, 2, 1, 2, 3, 2, 3, 0], g.get_first_cell(), g.get_cell_size(), [25, 25, 25], sign=1)
    for name, x in (("list", a), ("list-none", b), ("strip", c), ("strip-none", d)):
        assert_bit_equal(x, want, name)


def test_degenerate_triangles():
    # geo.rs:73-88: point / segment triangles, mixed with a regular one
    v = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [2, 2, 2], [3, 2.5, 2]], F)
    idx = [0, 1, 2, 3, 3, 3, 3, 3, 4, 3, 4, 4, 4, 3, 4, 0, 0, 1]
    g = Grid.from_bounding_box([-1, -1, -1], [4, 4, 4], [20, 20, 20])
    got = generate_grid_sdf(v, Topology.TriangleList(idx), g, SignMethod.Raycast)
    assert_bit_equal(got, oracle_grid(v, idx, g, SignMethod.Raycast, semantics=orc.EXACT), "degenerate grid")
    q = meshes.uniform_queries([-1, -1, -1], [4, 4, 4], 5000)
    for name, am, accel, sign in ACCELS:
        assert_bit_equal(generate_sdf(v, Topology.TriangleList(idx), q, am), orc.generate_sdf(v, idx, q, accel=accel, sign=sign), name)


def test_empty_mesh_and_panics(suzanne):
    v, idx = suzanne
    g = Grid.from_bounding_box([0, 0, 0], [1, 1, 1], [8, 8, 8])
    e = np.zeros((0, 3), F)
    fmax = np.finfo(F).max
    # grid path with no triangle: nothing seeded, every cell stays f32::MAX (generate/grid.rs:474)
    assert np.all(generate_grid_sdf(e, Topology.TriangleList(None), g, SignMethod.Normal) == fmax)
    assert np.all(generate_grid_sdf(e, Topology.TriangleList(None), g, SignMethod.Raycast) == fmax)
    q = np.zeros((5, 3), F)
    assert np.all(generate_sdf(e, Topology.TriangleList(None), q, AccelerationMethod.None_(SignMethod.Raycast)) == fmax)
    assert generate_sdf(e, Topology.TriangleList(None), q, AccelerationMethod.RtreeBvh).size == 0
    with pytest.raises(M2SPanic):   # vertices[i] out of range panics in the reference
        generate_sdf(v, Topology.TriangleList([0, 1, 5000]), q)
    with pytest.raises(M2SPanic):
        generate_grid_sdf(v, Topology.TriangleList([0, 1, 5000]), g)
    # NaN distance panics (lib.rs:257 expect("NaN distance")) in the compare_distances fold
    vn = np.array([[0, 0, 0], [1, 0, 0], [np.nan, 1, 0]], F)
    with pytest.raises(M2SPanic):
        generate_sdf(vn, Topology.TriangleList([0, 1, 2]), np.array([[0.5, 0.0, 0.0]], F), AccelerationMethod.None_(SignMethod.Normal))
    with pytest.raises(orc.OracleError):
        orc.generate_sdf(vn, [0, 1, 2], np.array([[0.5, 0.0, 0.0]], F), accel=0, sign=1)


# ---- BASELINE configs at oracle-affordable sizes ------------------------------------------------
def test_grid_blob100k_128_raycast():
    # config 2 (blob-100k, Raycast) at 128^3: oracle in seconds
    v, idx = meshes.named("blob-100k")
    g = grid_of(v, [128, 128, 128])
    t = M2STimings()
    got = generate_grid_sdf(v, Topology.TriangleList(idx), g, SignMethod.Raycast, timings=t)
    want = oracle_grid(v, idx, g, SignMethod.Raycast)
    assert_bit_equal(got, want, "blob-100k 128^3 Raycast")
    assert t.n_triangles == 100000 and t.n_units == 128 ** 3 and t.distance_ms > 0
    print(f"\n[timings 128^3] build {t.accel_build_ms:.3f} ms, sign {t.sign_ms:.3f} ms, distance {t.distance_ms:.3f} ms")


def test_grid_sheet100k_normal_open_surface():
    # config 5 shape (open surface, Normal sign, sign-leak parity vs CPU) at 96^3
    v, idx = meshes.named("sheet-100k")
    g = grid_of(v, [96, 96, 96])
    got = generate_grid_sdf(v, Topology.TriangleList(idx), g, SignMethod.Normal)
    assert_bit_equal(got, oracle_grid(v, idx, g, SignMethod.Normal), "sheet-100k Normal")


def _propagation_report(name, n, sign):
    """GPU (exact minimum) against the oracle's restatement of the reference's OWN grid semantics — the label
    propagation of generate/grid.rs:495-558, deterministic 1-heap form — on a BASELINE workload."""
    from mesh_to_sdf_amd.report import reference_parity

    v, idx = meshes.named(name)
    g = grid_of(v, [n, n, n])
    got = generate_grid_sdf(v, Topology.TriangleList(idx), g, sign)
    prop = oracle_grid(v, idx, g, sign, semantics=orc.PROPAGATE, heaps=1, threads=1)
    rep = reference_parity(got, prop, normal_sign=(sign == SignMethod.Normal))
    print(f"\n[reference-propagation parity] {name} {n}^3 {sign.name}: {rep}")
    return got, prop, rep


def test_reference_propagation_parity_blob100k_128_raycast():
    # config 2's mesh and sign rule at 128^3 (cells 3x a triangle: the regime where the propagation is least exact)
    got, prop, rep = _propagation_report("blob-100k", 128, SignMethod.Raycast)
    assert rep["ours_le_ref_everywhere"]            # the exact minimum never exceeds the propagated value
    assert rep["sign_mismatches"] == 0              # Raycast sign is independent of the magnitude path (grid.rs:568-684)
    assert rep["max_dev"] < 0.01                    # the reference's own cross-method tolerance (generic/bvh.rs:237-248)
    # the 1-heap propagation is deterministic and the GPU returns the exact minimum bit for bit, so the report is a constant:
    # 80.6052 % within 1e-5, 69.9275 % bit-identical, max deviation 1.306e-3 (oracle EXACT vs oracle PROPAGATE on the CPU, round 3)
    assert abs(rep["pct_within_1e-5"] - 80.6052) < 0.05 and abs(rep["pct_bit_identical"] - 69.9275) < 0.05
    assert abs(rep["max_dev"] - 1.3058e-3) < 2e-5


def test_reference_propagation_parity_sheet100k_96_normal():
    # config 5's mesh and sign rule (open surface, Normal) at 96^3
    got, prop, rep = _propagation_report("sheet-100k", 96, SignMethod.Normal)
    assert rep["ours_le_ref_everywhere"]
    assert rep["max_dev"] < 0.01
    # Normal sign comes from the triangle that wins the fold; the propagation sometimes ends on another triangle than the
    # nearest one, so a few cells next to the sheet's ridges differ in sign between the reference's two semantics
    # (oracle EXACT vs oracle PROPAGATE on the CPU: 289 of 884 736 cells).  The GPU must reproduce the EXACT signs bit for bit:
    v, idx = meshes.named("sheet-100k")
    exact = oracle_grid(v, idx, grid_of(v, [96, 96, 96]), SignMethod.Normal)
    assert np.array_equal(np.signbit(got), np.signbit(exact))
    assert rep["sign_mismatches"] == int(np.count_nonzero(np.signbit(exact) != np.signbit(prop)))
    assert rep["sign_mismatches"] == 289            # a constant of the two CPU semantics on this input (see the test above)
    assert abs(rep["pct_within_1e-5"] - 60.4709) < 0.05 and abs(rep["max_dev"] - 2.7595e-3) < 3e-5


def test_reference_propagation_parity_blob1m_128_raycast():
    # config 4's mesh and sign rule at 128^3: cells ten times a triangle — where the propagation is furthest from the minimum
    got, prop, rep = _propagation_report("blob-1M", 128, SignMethod.Raycast)
    assert rep["ours_le_ref_everywhere"] and rep["sign_mismatches"] == 0
    assert rep["max_dev"] < 0.01                    # generic/bvh.rs:237-248
    assert abs(rep["pct_within_1e-5"] - 41.0131) < 0.05 and abs(rep["pct_bit_identical"] - 28.0273) < 0.05
    assert abs(rep["max_dev"] - 2.0049e-3) < 3e-5


def test_reference_propagation_parity_sheet100k_256_normal():
    # config 5's mesh and sign rule at the size SURVEY 8(d) names for the full check (256^3; the 1-heap propagation takes ~90 s)
    got, prop, rep = _propagation_report("sheet-100k", 256, SignMethod.Normal)
    assert rep["ours_le_ref_everywhere"]
    assert rep["max_dev"] < 0.01
    v, idx = meshes.named("sheet-100k")
    exact = oracle_grid(v, idx, grid_of(v, [256, 256, 256]), SignMethod.Normal)
    assert np.array_equal(np.signbit(got), np.signbit(exact))                    # zero sign leaks against the exact semantics
    assert rep["sign_mismatches"] == int(np.count_nonzero(np.signbit(exact) != np.signbit(prop))) == 99
    assert abs(rep["pct_within_1e-5"] - 93.1411) < 0.05 and abs(rep["pct_bit_identical"] - 80.2015) < 0.05
    assert abs(rep["max_dev"] - 6.579e-4) < 1e-5


def test_config2_256_sub_lattice():
    """BASELINE config 2 at its literal size (256^3 x blob-100k, Raycast): every 3rd cell per axis (all 64 lanes of a packet) bit for bit against
    the oracle, signs included (whole sign lattice from the oracle's grid-line parity)."""
    import torch

    v, idx = meshes.named("blob-100k")
    n = 256
    g = grid_of(v, [n, n, n])
    sdf = generate_grid_sdf(torch.as_tensor(v, device="cuda"), Topology.TriangleList(torch.as_tensor(idx.astype(np.int64), device="cuda")),
                            g, SignMethod.Raycast).view(n, n, n)
    assert _lane_coverage(n, 3, FULL_START) == (64, 8)
    sub = _sub_cells(sdf, 3, FULL_START)
    mag = orc.generate_sdf(v, idx, _sub_lattice(g, n, 3, FULL_START), accel=1, sign=0, fast=True)
    assert_bit_equal(np.abs(sub), np.abs(mag), "256^3 sub-lattice magnitudes")
    par = orc.grid_ray_parity(v, idx, g.get_first_cell(), g.get_cell_size(), [n, n, n]).reshape(n, n, n, 3)
    inside = (par.sum(-1) >= 2)
    assert np.array_equal(np.signbit(sdf.cpu().numpy()), inside), "256^3 signs (all cells)"


def test_tiny_grids_brute_force_matches_walks_and_oracle(suzanne):
    """Tiny problems (cells x triangles <= 1e8 + 3000 T) skip the LBVH and run all voxels against all triangles (k_brute_split, 2-D
    decomposition + atomic minima): must equal the walks bit for bit — device and host results, slabs, persistent mesh, both sign
    rules, anisotropic cells, a ragged grid — and the oracle.  (The reference's criterion bench has this shape: 16^3 over 11 k triangles.)"""
    import os

    import torch

    from mesh_to_sdf_amd import Mesh

    if float(_lib.describe_knobs()["M2S_BRUTE_MAX"]) == 0.0:
        pytest.skip("this run forces the walks")
    v, idx = suzanne
    for counts, lo_hi in (([16, 16, 16], None), ([13, 7, 21], None), ([24, 5, 9], ((-2.0, -0.5, -1.0), (2.0, 0.5, 1.5)))):
        g = grid_of(v, counts) if lo_hi is None else Grid.from_bounding_box(lo_hi[0], lo_hi[1], counts)
        for sign in (SignMethod.Raycast, SignMethod.Normal):
            with _lib.knobs(M2S_BRUTE_MAX=0):
                walk = generate_grid_sdf(v, Topology.TriangleList(idx), g, sign)
            t = M2STimings()
            brute = generate_grid_sdf(v, Topology.TriangleList(idx), g, sign, timings=t)
            assert_bit_equal(brute, walk, f"brute force vs walk {counts} {sign.name}")
            assert_bit_equal(brute, oracle_grid(v, idx, g, sign), f"brute force vs oracle {counts} {sign.name}")
            assert t.n_units == g.get_total_cell_count() and t.accel_build_ms < 0.1          # no tree was built
            dv = torch.as_tensor(v, device="cuda")
            di = torch.as_tensor(idx.astype(np.int64), device="cuda")
            dev = generate_grid_sdf(dv, Topology.TriangleList(di), g, sign)
            assert_bit_equal(dev.cpu().numpy(), walk, "device-resident")
            parts = torch.full_like(dev, float("nan"))
            cut = counts[0] // 3
            for xs in ((0, cut), (cut, counts[0])):
                generate_grid_sdf(dv, Topology.TriangleList(di), g, sign, x_slab=xs, out=parts)
            assert_bit_equal(parts.cpu().numpy(), walk, "two slabs")
            with Mesh(dv, Topology.TriangleList(di)) as m:
                assert_bit_equal(m.generate_grid_sdf(g, sign).cpu().numpy(), walk, "persistent mesh")


def test_four_million_triangles_every_walk_equals_brute_force():
    """A mesh of 4 M triangles (23-bit record indices: the cut lists' length codes keep 4 mantissa bits; the build sorts 4 M keys):
    the lane walk (what such a coarse grid takes by default), the packet walk from the root and the packet walk from cut lists must all
    equal the tree-less all-pairs kernel (algorithm = 1, itself pinned on the oracle by the suzanne tests) bit for bit, both sign rules."""
    import os

    import torch

    from mesh_to_sdf_amd import meshes

    v, idx = meshes.blob(2000, 1001, detail=True)
    assert len(idx) // 3 == 4_000_000
    lo, hi = meshes.extended_bbox(v, 0.1)
    g = Grid.from_bounding_box(lo, hi, [40, 36, 44])
    dv = torch.as_tensor(v, device="cuda")
    topo = Topology.TriangleList(torch.as_tensor(idx.astype(np.int64), device="cuda").to(torch.int32))
    for sign in (SignMethod.Raycast, SignMethod.Normal):
        want = generate_grid_sdf(dv, topo, g, sign, algorithm=1).cpu().numpy()
        assert np.isfinite(want).all() and (want < 0).any() and (want > 0).any()
        for name, env in (("default", {}), ("packet walk", {"M2S_LANE_WALK": 0}),
                          ("packet walk + cut lists", {"M2S_LANE_WALK": 0, "M2S_CUT_MIN_PACKETS": 1}),
                          ("packet walk, split", {"M2S_LANE_WALK": 0, "M2S_SPLIT": 2, "M2S_SPLIT_BUDGET": 200}), ("lane walk", {"M2S_LANE_WALK": 1})):
            with _lib.knobs(**env):
                got = generate_grid_sdf(dv, topo, g, sign).cpu().numpy()
            assert_bit_equal(got, want, f"4 M triangles, {sign.name}, {name}")


def test_x_slabs_concatenate(suzanne):
    # multi-GPU sharding unit: x-slabs are contiguous ranges of the reference layout (grid.rs:122-124)
    v, idx = suzanne
    g = grid_of(v, [40, 24, 36])
    whole = generate_grid_sdf(v, Topology.TriangleList(idx), g, SignMethod.Raycast)
    out = np.full(g.get_total_cell_count(), np.nan, F)
    for x0, x1 in ((0, 13), (13, 14), (14, 40)):
        generate_grid_sdf(v, Topology.TriangleList(idx), g, SignMethod.Raycast, x_slab=(x0, x1), out=out)
    assert_bit_equal(out, whole, "slabs")


def test_device_resident_path(suzanne):
    import torch

    v, idx = suzanne
    g = grid_of(v, [48, 48, 48])
    dv = torch.as_tensor(v, device="cuda")
    di = torch.as_tensor(idx.astype(np.int64), device="cuda")
    got = generate_grid_sdf(dv, Topology.TriangleList(di), g, SignMethod.Raycast)
    assert got.is_cuda
    assert_bit_equal(got.cpu().numpy(), generate_grid_sdf(v, Topology.TriangleList(idx), g, SignMethod.Raycast), "device vs host path")
    q = meshes.uniform_queries(*meshes.extended_bbox(v, 0.2), 10000)
    gq = generate_sdf(dv, Topology.TriangleList(di), torch.as_tensor(q, device="cuda"))
    assert_bit_equal(gq.cpu().numpy(), generate_sdf(v, Topology.TriangleList(idx), q), "device generic")


# ---- size-independent properties at BASELINE's full size ----------------------------------------
def test_full_size_512_properties():
    """512^3 x blob-100k Raycast (the north-star workload): too big for the CPU oracle, so check
    (1) a lattice of 73^3 sub-sampled cells (stride 7: all 64 lanes of a packet) and 3 full x-planes bit-exactly against the oracle's
    generic path at the same cell centres, (2) 1-Lipschitz continuity along z on the whole grid,
    (3) the checksum is reproducible run to run."""
    import torch

    v, idx = meshes.named("blob-100k")
    n = 512
    g = grid_of(v, [n, n, n])
    dv, di = torch.as_tensor(v, device="cuda"), torch.as_tensor(idx.astype(np.int64), device="cuda")
    t = M2STimings()
    sdf = generate_grid_sdf(dv, Topology.TriangleList(di), g, SignMethod.Raycast, timings=t)
    print(f"\n[512^3] build {t.accel_build_ms:.3f} ms, sign {t.sign_ms:.3f} ms, distance {t.distance_ms:.3f} ms "
          f"-> {n ** 3 / (t.total_ms * 1e-3) / 1e6:.1f} Mvoxels/s (kernels only)")
    s3 = sdf.view(n, n, n)
    cs = float(g.get_cell_size()[2])
    dz = (s3[:, :, 1:].abs() - s3[:, :, :-1].abs()).abs().max().item()
    assert dz <= cs + 2e-6   # 1-Lipschitz up to f32 rounding of the two distances
    assert _lane_coverage(n, 7, FULL_START) == (64, 8)
    sub = _sub_cells(s3, 7, FULL_START)                      # 73^3 cells, every lane of a packet
    first, size = g.get_first_cell(), g.get_cell_size()
    mag = orc.generate_sdf(v, idx, _sub_lattice(g, n, 7, FULL_START), accel=1, sign=0, fast=True)
    assert_bit_equal(np.abs(sub), np.abs(mag), "512^3 sub-lattice magnitudes")
    # signs: the grid-line rule on the full grid is checked on whole x-planes against the oracle's parity planes
    par = orc.grid_ray_parity(v, idx, first, size, [n, n, n]).reshape(n, n, n, 3)
    inside = (par.sum(-1) >= 2)
    for x in (5, 256, 400):
        assert np.array_equal(np.signbit(s3[x].cpu().numpy()), inside[x]), f"sign plane x={x}"
    again = generate_grid_sdf(dv, Topology.TriangleList(di), g, SignMethod.Raycast)
    assert torch.equal(sdf, again)


def _sub_lattice(g, n, step, start=(0, 0, 0)):
    """Cell centres of the sub-lattice start[k] + step * j (j = 0, 1, ...) of an n^3 grid, as the grid computes them (grid.rs:135-141)."""
    first, size = g.get_first_cell(), g.get_cell_size()
    ax = [np.arange(start[k], n, step, dtype=np.float32) for k in range(3)]
    X, Y, Z = np.meshgrid(first[0] + ax[0] * size[0], first[1] + ax[1] * size[1], first[2] + ax[2] * size[2], indexing="ij")
    return np.stack([X, Y, Z], -1).reshape(-1, 3).astype(F)


def _sub_cells(s3, step, start=(0, 0, 0)):
    return s3[start[0]::step, start[1]::step, start[2]::step].contiguous().cpu().numpy().reshape(-1)


def _lane_coverage(n, step, start, brick=4):
    """How many of the brick^3 positions inside a packet brick (and of the 2^3 halves of an 8^3 super-brick) the sub-lattice visits."""
    ax = [np.arange(start[k], n, step) for k in range(3)]
    local = {(int(x) % brick, int(y) % brick, int(z) % brick) for x in set(ax[0] % 8) for y in set(ax[1] % 8) for z in set(ax[2] % 8)}
    halves = {(int(x) // 4, int(y) // 4, int(z) // 4) for x in set(ax[0] % 8) for y in set(ax[1] % 8) for z in set(ax[2] % 8)}
    return len(local), len(halves)


# Full-size oracle checks sample with a stride COPRIME to the 4 x 4 x 4 packet brick (and to the 8^3 super-brick) plus different
# start offsets per axis: every one of the 64 lanes of a packet, in both halves of a super-brick along every axis, meets the oracle on
# the default 512^3 / 1024^3 configuration (cut lists + queued leaf work + leaf size by density + XCD order).  Rounds 1-5 sampled every
# 8th (4th) cell from 0: the corner lane of every packet only.
FULL_START = (1, 2, 3)


def test_full_size_config4_blob1M_512_raycast():
    """BASELINE config 4 at full size on one GPU: blob-1M (triangles a third of a voxel wide), 512^3, Raycast, computed
    as the eight 64-layer x-slabs the 8-GPU run would use.  Every 7th cell per axis (all 64 lanes of a packet) bit-exactly against the oracle's
    generic path, three whole sign planes against the oracle's grid-line parity, slabs == one call."""
    import torch

    from mesh_to_sdf_amd import Mesh

    v, idx = meshes.named("blob-1M")
    n = 512
    g = grid_of(v, [n, n, n])
    dv, di = torch.as_tensor(v, device="cuda"), torch.as_tensor(idx.astype(np.int64), device="cuda")
    sdf = torch.empty(n ** 3, dtype=torch.float32, device="cuda")
    with Mesh(dv, Topology.TriangleList(di)) as m:
        for r in range(8):
            m.generate_grid_sdf(g, SignMethod.Raycast, x_slab=(64 * r, 64 * r + 64), out=sdf)
    s3 = sdf.view(n, n, n)
    assert _lane_coverage(n, 7, FULL_START) == (64, 8)
    sub = _sub_cells(s3, 7, FULL_START)
    mag = orc.generate_sdf(v, idx, _sub_lattice(g, n, 7, FULL_START), accel=1, sign=0, fast=True)
    assert_bit_equal(np.abs(sub), np.abs(mag), "blob-1M 512^3 sub-lattice magnitudes")
    par = orc.grid_ray_parity(v, idx, g.get_first_cell(), g.get_cell_size(), [n, n, n]).reshape(n, n, n, 3)
    inside = par.sum(-1) >= 2
    for x in (3, 255, 256, 470):
        assert np.array_equal(np.signbit(s3[x].cpu().numpy()), inside[x]), f"sign plane x={x}"
    whole = generate_grid_sdf(dv, Topology.TriangleList(di), g, SignMethod.Raycast)
    assert torch.equal(sdf, whole)


def test_full_size_config5_sheet100k_1024_normal():
    """BASELINE config 5 at full size on one GPU: open surface, 1024^3 (4 GiB of f32), Normal sign, computed as the
    eight 128-layer x-slabs of the 8-GPU run.  Every 9th cell per axis (114^3 cells, all 64 lanes of a packet) bit-exactly — magnitude AND sign,
    i.e. zero sign leaks — against the oracle's compare_distances fold over all triangles at the same points."""
    import torch

    from mesh_to_sdf_amd import Mesh

    v, idx = meshes.named("sheet-100k")
    n = 1024
    g = grid_of(v, [n, n, n])
    dv, di = torch.as_tensor(v, device="cuda"), torch.as_tensor(idx.astype(np.int64), device="cuda")
    sdf = torch.empty(n ** 3, dtype=torch.float32, device="cuda")
    with Mesh(dv, Topology.TriangleList(di)) as m:
        for r in range(8):
            m.generate_grid_sdf(g, SignMethod.Normal, x_slab=(128 * r, 128 * r + 128), out=sdf)
    s3 = sdf.view(n, n, n)
    assert _lane_coverage(n, 9, FULL_START) == (64, 8)
    sub = _sub_cells(s3, 9, FULL_START)                      # 114^3 cells, every lane of a packet
    want = orc.generate_sdf(v, idx, _sub_lattice(g, n, 9, FULL_START), accel=1, sign=1, fast=True)
    assert_bit_equal(sub, want, "sheet-100k 1024^3 sub-lattice, Normal sign")
    cs = float(g.get_cell_size()[2])
    dz = 0.0
    for x0 in range(0, n, 128):   # 1-Lipschitz along z on the whole grid, slab by slab (keeps the temporaries small)
        a = s3[x0 : x0 + 128].abs()
        dz = max(dz, float((a[:, :, 1:] - a[:, :, :-1]).abs().max()))
    assert dz <= cs + 2e-6
    # the single 1024^3 call makes its cut lists in two levels (262 144 fine k_cut waves; a 128-layer slab's 32 768 stay below the automatic
    # threshold): it must reproduce the slabs — which have just met the oracle on every lane of a packet — bit for bit
    del s3
    whole = generate_grid_sdf(dv, Topology.TriangleList(di), g, SignMethod.Normal)
    assert torch.equal(whole.view(torch.int32), sdf.view(torch.int32)), "1024^3 in one call (two-level cut lists) differs from its slabs"
    del sdf, whole


@pytest.mark.parametrize("sign", [SignMethod.Raycast, SignMethod.Normal])
def test_host_pointer_result_in_pieces(suzanne, sign):
    """A host-pointer result of 16 MiB or more leaves the device in x-pieces (capi.hip run_grid_distance_to_host): seeds
    and cut lists are prepared once for the slab, every piece is a walk with its brick offset into them.  172 x 160 x 160
    cells = 4 pieces (the last one ragged), also as an x-slab that does not start at 0."""
    v, idx = suzanne
    g = grid_of(v, [172, 160, 160])
    want = oracle_grid(v, idx, g, sign)
    got = generate_grid_sdf(v, Topology.TriangleList(idx), g, sign)
    assert_bit_equal(got, want, f"host pieces {sign.name}")
    out = np.full(g.get_total_cell_count(), np.nan, F)
    generate_grid_sdf(v, Topology.TriangleList(idx), g, sign, x_slab=(3, 171), out=out)
    row = 160 * 160
    assert_bit_equal(out[3 * row : 171 * row], want[3 * row : 171 * row], f"host pieces of a slab {sign.name}")
    assert np.isnan(out[: 3 * row]).all() and np.isnan(out[171 * row :]).all()


# ---- persistent mesh (include/m2s.h m2s_mesh) -----------------------------------------------------
def test_persistent_mesh_matches_one_shot(suzanne):
    import torch

    from mesh_to_sdf_amd import Mesh

    v, idx = suzanne
    q = meshes.uniform_queries(*meshes.extended_bbox(v, 0.2), 8000)
    g1, g2 = grid_of(v, [40, 28, 36]), grid_of(v, [33, 33, 33], 0.3)
    with Mesh(v, Topology.TriangleList(idx)) as m:                      # host pointers
        assert m.triangle_count() == 968
        for g in (g1, g2, g1):                                           # sign-plane cache: miss, miss, miss(g1 again after g2)
            for sign in (SignMethod.Raycast, SignMethod.Normal):
                assert_bit_equal(m.generate_grid_sdf(g, sign), generate_grid_sdf(v, Topology.TriangleList(idx), g, sign), "mesh grid")
        assert_bit_equal(m.generate_grid_sdf(g1, SignMethod.Raycast), generate_grid_sdf(v, Topology.TriangleList(idx), g1), "cached planes")
        for name, am, accel, sign in ACCELS:
            assert_bit_equal(m.generate_sdf(q, am), generate_sdf(v, Topology.TriangleList(idx), q, am), name)
    dv, di = torch.as_tensor(v, device="cuda"), torch.as_tensor(idx.astype(np.int64), device="cuda")
    with Mesh(dv, Topology.TriangleList(di)) as m:                       # device pointers, asynchronous slab calls
        out = torch.full((g1.get_total_cell_count(),), float("nan"), device="cuda")
        for x0, x1 in ((0, 7), (7, 8), (8, 40)):
            m.generate_grid_sdf(g1, SignMethod.Raycast, x_slab=(x0, x1), out=out, synchronous=False)
        t = m.drain_timings()
        assert t.distance_launches == 3 and t.n_units == g1.get_total_cell_count() and t.distance_ms > 0
        assert_bit_equal(out.cpu().numpy(), generate_grid_sdf(v, Topology.TriangleList(idx), g1), "async slabs")
    with pytest.raises(M2SPanic):
        Mesh(v, Topology.TriangleList([0, 1, 99999]))


def test_persistent_mesh_leaf_size_follows_the_grid():
    """A resident tree is re-marked with the leaf size each grid wants (grid_leaf_max: 2 / 4 / 8 / 16 triangles by triangles per brick,
    capi.hip set_leaf_size); grids of every class in a row, queries and asynchronous slab calls in between, against one-shot calls."""
    import torch

    from mesh_to_sdf_amd import Mesh

    v, idx = meshes.named("blob-6k")
    topo = Topology.TriangleList(idx)
    q = meshes.uniform_queries(*meshes.extended_bbox(v, 0.2), 3000)
    dv, di = torch.as_tensor(v, device="cuda"), torch.as_tensor(idx.astype(np.int64), device="cuda")
    with Mesh(dv, Topology.TriangleList(di)) as m:
        for n in (96, 24, 12, 48, 64, 12, 96):                      # 0.4, 27, 220, 3.4, 1.4, 220, 0.4 triangles per brick: leaves of 2, 8, 16, 8, 4, 16, 2
            g = grid_of(v, [n, n, n])
            for sign in (SignMethod.Raycast, SignMethod.Normal):
                got = m.generate_grid_sdf(g, sign)
                assert_bit_equal(got.cpu().numpy(), generate_grid_sdf(v, topo, g, sign), f"mesh grid {n}^3 {sign.name}")
            name, am, accel, sign = ACCELS[n % len(ACCELS)]
            assert_bit_equal(m.generate_sdf(q, am).cpu().numpy() if hasattr(m.generate_sdf(q, am), "cpu") else m.generate_sdf(q, am),
                             generate_sdf(v, topo, q, am), f"queries after {n}^3: {name}")
            out = torch.full((g.get_total_cell_count(),), float("nan"), device="cuda")
            for x0, x1 in ((0, n // 3), (n // 3, n)):
                m.generate_grid_sdf(g, SignMethod.Raycast, x_slab=(x0, x1), out=out, synchronous=False)
            m.drain_timings()
            assert_bit_equal(out.cpu().numpy(), generate_grid_sdf(v, topo, g), f"async slabs {n}^3")


def test_far_field_every_form_equals_all_pairs():
    """tools/soak_far.py, 80 cases: a small mesh in a box 3 - 100 mesh sizes wide (off-centre: large coordinates), where a voxel sees many triangles
    at almost the same distance and the pruning margins decide what is evaluated; five walk forms against the all-pairs kernel, bit for bit."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "soak_far.py"), "--seeds", "80", "--seconds", "120"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "80 cases" in r.stdout and ", 0 differences" in r.stdout, r.stdout[-500:]


def test_sharded_driver_single_process(suzanne):
    # the multi-GPU driver with world size 1 (no process group): chunk plan + persistent mesh + async calls
    import torch

    from mesh_to_sdf_amd.distributed import generate_grid_sdf_sharded

    v, idx = suzanne
    g = grid_of(v, [36, 20, 28])
    dv, di = torch.as_tensor(v, device="cuda"), torch.as_tensor(idx.astype(np.int64), device="cuda")
    out = generate_grid_sdf_sharded(dv, Topology.TriangleList(di), g, SignMethod.Raycast, chunks=4)
    assert_bit_equal(out.cpu().numpy(), generate_grid_sdf(v, Topology.TriangleList(idx), g), "sharded driver")
    # the query driver (split the query array, all-gather), world size 1
    from mesh_to_sdf_amd.distributed import generate_sdf_sharded

    lo, hi = meshes.extended_bbox(v, 0.2)
    q = meshes.uniform_queries(lo, hi, 4099)
    got = generate_sdf_sharded(dv, Topology.TriangleList(di), torch.as_tensor(q, device="cuda"), AccelerationMethod.RtreeBvh)
    assert_bit_equal(got.cpu().numpy(), orc.generate_sdf(v, idx, q, accel=3, fast=True), "sharded queries")


# ---- nastier inputs ---------------------------------------------------------------------------------
def test_huge_and_tiny_triangles_mixed():
    # two triangles spanning far beyond the grid (k_ray_mark's wave-cooperative window path) + a small blob
    v, idx = meshes.blob(20, 11)
    big = np.array([[-50, -50, 0.3], [50, -50, 0.31], [0, 60, 0.29], [-40, 0.2, -45], [45, 0.21, -44], [0.5, 0.19, 70]], F)
    vv = np.concatenate([v, big]).astype(F)
    ii = np.concatenate([idx, len(v) + np.arange(6, dtype=np.uint32)])
    g = Grid.from_bounding_box([-2, -2, -2], [2, 2, 2], [44, 40, 36])
    for sign in (SignMethod.Raycast, SignMethod.Normal):
        got = generate_grid_sdf(vv, Topology.TriangleList(ii), g, sign)
        assert_bit_equal(got, oracle_grid(vv, ii, g, sign, semantics=orc.EXACT), f"huge triangles {sign.name}")
    q = meshes.uniform_queries([-3, -3, -3], [3, 3, 3], 6000)
    for name, am, accel, sign in ACCELS:
        assert_bit_equal(generate_sdf(vv, Topology.TriangleList(ii), q, am), orc.generate_sdf(vv, ii, q, accel=accel, sign=sign), name)


@pytest.mark.parametrize("ntri", [1, 2, 3, 5])
def test_tiny_meshes(ntri):
    rng = np.random.default_rng(ntri)
    v = rng.uniform(-1, 1, (3 * ntri, 3)).astype(F)
    g = Grid.from_bounding_box([-1.5, -1.5, -1.5], [1.5, 1.5, 1.5], [17, 19, 23])
    for sign in (SignMethod.Raycast, SignMethod.Normal):
        got = generate_grid_sdf(v, Topology.TriangleList(None), g, sign)
        assert_bit_equal(got, oracle_grid(v, None, g, sign, semantics=orc.EXACT), f"{ntri} tris {sign.name}")
    q = meshes.uniform_queries([-2, -2, -2], [2, 2, 2], 3000)
    for name, am, accel, sign in ACCELS:
        assert_bit_equal(generate_sdf(v, Topology.TriangleList(None), q, am), orc.generate_sdf(v, None, q, accel=accel, sign=sign), name)


def test_grid_large_coordinates():
    v, idx = meshes.blob(50, 31)
    v = (v * F(3.0) + np.array([1000.0, -2000.0, 500.0], F)).astype(F)
    g = grid_of(v, [40, 40, 40])
    for sign in (SignMethod.Raycast, SignMethod.Normal):
        assert_bit_equal(generate_grid_sdf(v, Topology.TriangleList(idx), g, sign), oracle_grid(v, idx, g, sign), f"offset grid {sign.name}")


def test_nonfinite_vertices_raycast():
    # f32::min drops NaN distances (default.rs:47) and NaN never passes the strict ray test (geo.rs:203):
    # in Raycast mode a triangle with a NaN / inf vertex must simply not matter, also through the BVH.
    v, idx = meshes.blob(24, 13)
    bad = np.array([[np.nan, 0.1, 0.2], [0.3, np.inf, 0.1], [0.2, 0.2, -np.inf], [0.1, 0.0, 0.3]], F)
    vv = np.concatenate([v, bad]).astype(F)
    n = len(v)
    ii = np.concatenate([idx, np.array([n, 0, 1, n + 1, 2, 3, 4, n + 2, 5, n + 3, n, 6], np.uint32)])
    g = grid_of(v, [24, 24, 24], 0.2)
    got = generate_grid_sdf(vv, Topology.TriangleList(ii), g, SignMethod.Raycast)
    assert_bit_equal(got, oracle_grid(vv, ii, g, SignMethod.Raycast, semantics=orc.EXACT), "non-finite grid")
    q = meshes.uniform_queries(*meshes.extended_bbox(v, 0.3), 4000)
    for name, am, accel, sign in (ACCELS[0], ACCELS[2], ACCELS[5]):
        assert_bit_equal(generate_sdf(vv, Topology.TriangleList(ii), q, am), orc.generate_sdf(vv, ii, q, accel=accel, sign=sign), name)


def _two_rank_worker(rank, world, port, q):
    import os

    import torch
    import torch.distributed as dist

    from mesh_to_sdf_amd.distributed import generate_grid_sdf_sharded

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)   # RCCL refuses two ranks on one GPU; gloo moves CUDA tensors
    try:
        v, idx = meshes.blob(40, 21)
        g = grid_of(v, [48, 20, 28])
        dv, di = torch.as_tensor(v, device="cuda:0"), torch.as_tensor(idx.astype(np.int64), device="cuda:0")
        out = generate_grid_sdf_sharded(dv, Topology.TriangleList(di), g, SignMethod.Raycast, chunks=3)
        from mesh_to_sdf_amd.distributed import generate_sdf_sharded

        lo, hi = meshes.extended_bbox(v, 0.2)
        pts = torch.as_tensor(meshes.uniform_queries(lo, hi, 3001), device="cuda:0")   # uneven split: padded gather
        dq = generate_sdf_sharded(dv, Topology.TriangleList(di), pts, AccelerationMethod.RtreeBvh)
        q.put((rank, (out.cpu().numpy(), dq.cpu().numpy())))
    finally:
        dist.destroy_process_group()


def test_two_ranks_share_one_gpu_gloo():
    """The complete N=2 driver flow (piece partition, asynchronous slab calls through the persistent mesh,
    chunked in-place gathers) with both ranks on cuda:0: every rank must end with the full, exact grid."""
    import socket

    import torch.multiprocessing as mp

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_two_rank_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
    codes = [p.exitcode for p in procs]
    assert codes == [0, 0], f"worker exit codes {codes}"
    v, idx = meshes.blob(40, 21)
    g = grid_of(v, [48, 20, 28])
    want = oracle_grid(v, idx, g, SignMethod.Raycast)
    lo, hi = meshes.extended_bbox(v, 0.2)
    want_q = orc.generate_sdf(v, idx, meshes.uniform_queries(lo, hi, 3001), accel=3, fast=True)
    for r in range(2):
        assert_bit_equal(res[r][0], want, f"rank {r}")
        assert_bit_equal(res[r][1], want_q, f"rank {r} queries")


def test_generic_10M_queries_subsample():
    """BASELINE config 3 at full size: 10 M splitmix queries x blob-100k, RtreeBvh, resident on the GPU.
    The oracle checks a strided subsample bit for bit; the rest is covered by a permutation property:
    the same queries in reversed order must give the reversed result (the kernels Morton-sort internally)."""
    import torch

    v, idx = meshes.named("blob-100k")
    lo, hi = meshes.extended_bbox(v, 0.1)
    nq = 10_000_000
    q = meshes.uniform_queries(lo, hi, nq)
    dv, di = torch.as_tensor(v, device="cuda"), torch.as_tensor(idx.astype(np.int64), device="cuda")
    dq = torch.as_tensor(q, device="cuda")
    t = M2STimings()
    got = generate_sdf(dv, Topology.TriangleList(di), dq, AccelerationMethod.RtreeBvh, timings=t)
    print(f"\n[10M queries RtreeBvh] build {t.accel_build_ms:.2f} ms, distance {t.distance_ms:.2f} ms -> {nq / t.total_ms / 1e3:.0f} Mqueries/s")
    sub = slice(0, nq, 251)
    want = orc.generate_sdf(v, idx, q[sub], accel=3, fast=True)
    assert_bit_equal(got[sub].cpu().numpy(), want, "10M RtreeBvh subsample")
    rev = generate_sdf(dv, Topology.TriangleList(di), torch.flip(dq, dims=[0]), AccelerationMethod.RtreeBvh)
    assert torch.equal(torch.flip(rev, dims=[0]), got)


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("M2S_FUZZ_SEEDS", "12"))))
def test_fuzz_tree_walk_equals_brute_force(seed):
    """Random triangle soups at random scales / offsets / anisotropic (also negative) cell sizes: the pruned LBVH walk
    (algorithm 0) must equal the on-device brute force over all triangles (algorithm 1) bit for bit, in both
    sign modes and for random queries — pruning slack, oriented bounds, leaf pre-test and seeds are all in play.
    (Brute force itself is pinned to the oracle by the algorithm=1 variants of the tests above.)"""
    rng = np.random.default_rng(1000 + seed)
    nt = int(rng.integers(1, 4000))
    scale = float(10.0 ** rng.uniform(-3, 3))
    offset = rng.uniform(-1, 1, 3) * scale * float(10.0 ** rng.uniform(0, 2.5)) * (seed % 3 == 0)
    centers = rng.uniform(-1, 1, (nt, 1, 3)) * scale
    size = scale * 10.0 ** rng.uniform(-3, -0.3, (nt, 1, 1))
    tri = (centers + rng.standard_normal((nt, 3, 3)) * size + offset).astype(F)
    if seed % 4 == 1:                       # some exactly degenerate and duplicated triangles
        tri[::7, 2] = tri[::7, 1]
        tri[5::11] = tri[4::11][: tri[5::11].shape[0]]
    v = tri.reshape(-1, 3)
    idx = np.arange(v.shape[0], dtype=np.uint32)
    lo, hi = v.min(0), v.max(0)
    ext = np.maximum(hi - lo, 1e-3 * scale)
    lo, hi = lo - rng.uniform(0.0, 0.5) * ext, hi + rng.uniform(0.0, 0.5) * ext
    counts = [int(c) for c in rng.integers(5, 70, 3)]
    if seed % 5 == 2:
        lo[0], hi[0] = hi[0], lo[0]         # negative cell size along x
    grid = Grid.from_bounding_box(lo, hi, counts)
    if True:
        for sign in (SignMethod.Raycast, SignMethod.Normal):
            a = generate_grid_sdf(v, Topology.TriangleList(idx), grid, sign, algorithm=0)
            b = generate_grid_sdf(v, Topology.TriangleList(idx), grid, sign, algorithm=1)
            assert_bit_equal(a, b, f"grid seed {seed} {sign.name} nt {nt} scale {scale:.3g}")
            if seed % 2 == 0:                   # the same grid in x-slab pieces: per-piece seeds and cut lists
                xs = sorted({0, counts[0]} | {int(x) for x in rng.integers(0, counts[0] + 1, 2)})
                c = np.full(grid.get_total_cell_count(), np.nan, F)
                for x0, x1 in zip(xs[:-1], xs[1:]):
                    generate_grid_sdf(v, Topology.TriangleList(idx), grid, sign, x_slab=(x0, x1), out=c)
                assert_bit_equal(c, b, f"grid pieces seed {seed} {sign.name}")
        q = (lo + rng.uniform(-0.3, 1.3, (20000, 3)) * (hi - lo)).astype(F)
        for am in (AccelerationMethod.RtreeBvh, AccelerationMethod.Rtree, AccelerationMethod.Bvh(SignMethod.Normal)):
            a = generate_sdf(v, Topology.TriangleList(idx), q, am, algorithm=0)
            b = generate_sdf(v, Topology.TriangleList(idx), q, am, algorithm=1)
            assert_bit_equal(a, b, f"queries seed {seed} accel {am.kind}")


def _uv_sphere(nu, nv, radius=1.0, centre=(0.0, 0.0, 0.0)):
    th = np.linspace(0.0, np.pi, nv + 1)[:, None]
    ph = np.linspace(0.0, 2 * np.pi, nu, endpoint=False)[None, :]
    pts = np.stack([np.sin(th) * np.cos(ph), np.sin(th) * np.sin(ph), np.cos(th) * np.ones_like(ph)], -1).reshape(-1, 3)
    tris = []
    for i in range(nv):
        for j in range(nu):
            a, b = i * nu + j, i * nu + (j + 1) % nu
            c, d = a + nu, b + nu
            tris += [[a, c, b], [b, c, d]]
    return (pts * radius + np.asarray(centre)).astype(F), np.asarray(tris, np.uint32).reshape(-1)


def _box_surface(n):
    """Six faces of [-1, 1]^3, each an n x n grid of quads (two triangles each)."""
    u = np.linspace(-1.0, 1.0, n + 1)
    verts, tris = [], []
    for axis in range(3):
        for side in (-1.0, 1.0):
            base = len(verts)
            for a in u:
                for b in u:
                    p = [0.0, 0.0, 0.0]
                    p[axis], p[(axis + 1) % 3], p[(axis + 2) % 3] = side, a, b
                    verts.append(p)
            for i in range(n):
                for j in range(n):
                    q0 = base + i * (n + 1) + j
                    tris += [[q0, q0 + 1, q0 + n + 1], [q0 + 1, q0 + n + 2, q0 + n + 1]]
    return np.asarray(verts, F), np.asarray(tris, np.uint32).reshape(-1)


@pytest.mark.parametrize("shape", ["sphere from inside", "sphere from outside", "box", "two sheets", "shell far from the origin"])
@pytest.mark.parametrize("sign", [SignMethod.Raycast, SignMethod.Normal])
def test_cut_lists_where_many_triangles_are_equidistant(shape, sign):
    """The brick-level tests of k_cut (sphere and gradient test, distance.hip) on inputs built to hurt them: voxels for which
    hundreds of triangles are (nearly) equidistant — the centre of a sphere, the medial planes of a box, the mid-plane between
    two sheets — and far-field bricks at many cells from the surface, where the gradient test does the pruning.  The lists are
    forced onto these small grids (fixture) and the result must equal the on-device brute force bit for bit."""
    import os

    with _lib.knobs(M2S_CUT_MIN_PACKETS=8):
        if shape.startswith("sphere"):
            v, idx = _uv_sphere(96, 48)
            lo, hi = (np.array([-0.6] * 3, F), np.array([0.6] * 3, F)) if "inside" in shape else (np.array([-4.0, -3.0, -2.5], F), np.array([3.0, 4.0, 5.0], F))
        elif shape == "box":
            v, idx = _box_surface(24)
            lo, hi = np.array([-0.97] * 3, F), np.array([0.97] * 3, F)
        elif shape == "two sheets":
            a, ia = meshes.sheet(40, 40)
            b = a.copy()
            a[:, 2] = 0.5
            b[:, 2] = -0.5
            v, idx = np.concatenate([a, b]), np.concatenate([ia, ia + a.shape[0]]).astype(np.uint32)
            lo, hi = np.array([-0.8, -0.8, -0.45], F), np.array([0.8, 0.8, 0.45], F)
        else:
            v, idx = _uv_sphere(64, 32, radius=3.0, centre=(900.0, -450.0, 120.0))
            lo, hi = v.min(0) - 8.0, v.max(0) + 8.0
        g = Grid.from_bounding_box(lo, hi, [72, 64, 80])
        a = generate_grid_sdf(v, Topology.TriangleList(idx), g, sign, algorithm=0)
        b = generate_grid_sdf(v, Topology.TriangleList(idx), g, sign, algorithm=1)
        assert_bit_equal(a, b, f"{shape} {sign.name}")


@pytest.mark.parametrize("counts", [[260, 9, 33], [7, 300, 5], [3, 6, 500], [64, 1, 64]])
def test_anisotropic_grids_use_non_cubic_bricks(suzanne, counts):
    """Strongly anisotropic cell sizes make the kernels pick a packet brick that is not 4x4x4 (64x1x1 ... 1x1x64):
    exact result, also when the grid is computed in x-slab pieces."""
    v, idx = suzanne
    g = grid_of(v, counts)
    for sign in (SignMethod.Raycast, SignMethod.Normal):
        want = oracle_grid(v, idx, g, sign)
        assert_bit_equal(generate_grid_sdf(v, Topology.TriangleList(idx), g, sign), want, f"{counts} {sign.name}")
        out = np.full(g.get_total_cell_count(), np.nan, F)
        nx = counts[0]
        cuts = sorted({0, nx // 3, (2 * nx) // 3 + 1, nx} & set(range(nx + 1)))
        for a, b in zip(cuts[:-1], cuts[1:]):
            generate_grid_sdf(v, Topology.TriangleList(idx), g, sign, x_slab=(a, b), out=out)
        assert_bit_equal(out, want, f"{counts} {sign.name} in slabs {cuts}")


@pytest.mark.parametrize("sign", [SignMethod.Raycast, SignMethod.Normal])
def test_many_small_triangles_per_voxel_lane_walk(sign):
    """100k triangles into a ~24^3 grid: hundreds of triangles per brick, the regime in which the library switches
    from the packet walk to independent per-lane walks (k_lane).  Same exact result; also in x-slab pieces."""
    v, idx = meshes.named("blob-100k")
    g = grid_of(v, [20, 24, 28])
    want = oracle_grid(v, idx, g, sign)
    assert_bit_equal(generate_grid_sdf(v, Topology.TriangleList(idx), g, sign), want, f"lane walk {sign.name}")
    out = np.full(g.get_total_cell_count(), np.nan, F)
    for a, b in ((0, 7), (7, 8), (8, 20)):
        generate_grid_sdf(v, Topology.TriangleList(idx), g, sign, x_slab=(a, b), out=out)
    assert_bit_equal(out, want, f"lane walk {sign.name} in slabs")


@pytest.mark.parametrize("n,mesh", [(96, "blob-100k"), (128, "blob-100k"), (72, "blob-11k")])
def test_split_walk_stragglers_only(n, mesh):
    """The split walk as it runs by default on mid-size grids — stragglers only, chosen by the launch's own clock, so WHICH packets
    are suspended differs from run to run — against the same walk without it: bit-identical, both sign rules, whole grid and an x-slab
    (whose one-shot call marks the sign planes of its own layers only), several repetitions."""
    import torch

    if float(_lib.describe_knobs()["M2S_BRUTE_MAX"]) == 0.0:
        pytest.skip("once is enough (the forced modes cover the mechanics)")
    v, idx = meshes.blob(80, 71) if mesh == "blob-11k" else meshes.named(mesh)
    lo, hi = meshes.extended_bbox(v, 0.1)
    g = Grid.from_bounding_box(lo, hi, [n, n - 8, n + 4])
    dv = torch.as_tensor(v, device="cuda")
    topo = Topology.TriangleList(torch.as_tensor(idx.astype(np.int64), device="cuda").to(torch.int32))
    for sign in (SignMethod.Raycast, SignMethod.Normal):
        for slab in (None, (16, 48)):
            with _lib.knobs(M2S_SPLIT=0, M2S_LANE_WALK=0, M2S_BRUTE_MAX=0):
                want = generate_grid_sdf(dv, topo, g, sign, x_slab=slab).cpu().numpy()
            for patience in (1.5, 0.25, 0.0):
                with _lib.knobs(M2S_SPLIT=1, M2S_SPLIT_PATIENCE=patience, M2S_LANE_WALK=0, M2S_BRUTE_MAX=0):
                    for rep in range(2):
                        got = generate_grid_sdf(dv, topo, g, sign, x_slab=slab).cpu().numpy()
                        sel = slice(None) if slab is None else slice(slab[0] * g.get_cell_count()[1] * g.get_cell_count()[2], slab[1] * g.get_cell_count()[1] * g.get_cell_count()[2])
                        assert_bit_equal(got[sel], want[sel], f"{mesh} {n}^3 {sign.name} slab {slab} patience {patience} rep {rep}")
