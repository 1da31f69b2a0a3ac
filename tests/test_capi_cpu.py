"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/m2s.h declares,
its host-side grid/topology helpers reproduce the reference arithmetic, and without a GPU the
compute entry points fail loudly (there is no CPU fallback).  No GPU needed."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import oracle as orc
from mesh_to_sdf_amd import _lib
from mesh_to_sdf_amd import AccelerationMethod, Grid, M2SError, SignMethod, Topology, generate_grid_sdf, generate_sdf

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_lib.SO_PATH):
        _lib.build()
    return _lib.lib()


def test_exports_match_header(lib):
    hdr = open(os.path.join(ROOT, "include", "m2s.h")).read()
    declared = set(re.findall(r"\b(m2s_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(_lib.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.m2s_version() == 5


def test_struct_layout_matches_header(tmp_path):
    # the ctypes mirrors must have the layout a C compiler gives the structs of m2s.h
    import subprocess

    src = tmp_path / "probe.c"
    src.write_text(
        '#include <stdio.h>\n#include <stddef.h>\n#include "m2s.h"\n'
        'int main(void){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %d %zu %zu %d %d\\n", sizeof(m2s_grid), sizeof(m2s_timings), sizeof(m2s_opts),'
        ' offsetof(m2s_opts, x_begin), offsetof(m2s_opts, timings), offsetof(m2s_timings, n_units),'
        ' offsetof(m2s_opts, lane), offsetof(m2s_opts, peer_out), sizeof(m2s_multi_opts), offsetof(m2s_multi_opts, timings),'
        ' M2S_OPTS_V1_SIZE, offsetof(m2s_multi_opts, partition), offsetof(m2s_multi_opts, partition_used), M2S_MULTI_OPTS_V1_SIZE,'
        ' M2S_MULTI_OPTS_V2_SIZE);return 0;}\n'
    )
    exe = tmp_path / "probe"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    want = [C.sizeof(_lib.M2SGrid), C.sizeof(_lib.M2STimings), C.sizeof(_lib.M2SOpts), _lib.M2SOpts.x_begin.offset,
            _lib.M2SOpts.timings.offset, _lib.M2STimings.n_units.offset, _lib.M2SOpts.lane.offset, _lib.M2SOpts.peer_out.offset,
            C.sizeof(_lib.M2SMultiOpts), _lib.M2SMultiOpts.timings.offset, _lib.OPTS_V1_SIZE, _lib.M2SMultiOpts.partition.offset,
            _lib.M2SMultiOpts.partition_used.offset, _lib.MULTI_OPTS_V1_SIZE, _lib.MULTI_OPTS_V2_SIZE]
    assert _lib.M2SOpts.lane.offset == _lib.OPTS_V1_SIZE   # version-0.1 callers stop exactly where the new fields begin
    assert _lib.M2SMultiOpts.partition.offset == _lib.MULTI_OPTS_V1_SIZE and _lib.M2SMultiOpts.partition_used.offset == _lib.MULTI_OPTS_V2_SIZE
    assert got == want


def test_grid_helpers_match_reference_arithmetic(lib):
    rng = np.random.default_rng(0)
    for _ in range(200):
        mn = rng.uniform(-5, 5, 3).astype(np.float32)
        mx = (mn + rng.uniform(0.1, 7, 3)).astype(np.float32)
        cnt = rng.integers(1, 600, 3)
        g = Grid.from_bounding_box(mn, mx, cnt)
        first, size, _ = orc.grid_from_bounding_box(mn, mx, cnt)
        assert np.array_equal(g.get_first_cell(), first) and np.array_equal(g.get_cell_size(), size)
        cell = [int(rng.integers(0, c)) for c in cnt]
        assert np.array_equal(g.get_cell_center(cell), orc.grid_cell_center(first, size, cnt, cell))
        assert g.get_cell_idx(cell) == orc.grid_cell_idx(cnt, cell)
        assert g.get_cell_integer_coordinates(g.get_cell_idx(cell)) == cell
        p = rng.uniform(-6, 8, 3).astype(np.float32)
        inside, c2 = orc.grid_snap(first, size, cnt, p)
        assert g.snap_point_to_grid(p) == ("Inside" if inside else "Outside", c2)
    # grid.rs:201-211, 282-297
    g = Grid.from_bounding_box([-1.0, 0.0, 1.0], [0.0, 2.0, 5.0], [2, 2, 2])
    assert g.get_first_cell().tolist() == [-0.75, 0.5, 2.0] and g.get_cell_size().tolist() == [0.5, 1.0, 2.0]
    mn, mx = g.get_bounding_box()
    assert mn.tolist() == [-1.0, 0.0, 1.0] and mx.tolist() == [0.0, 2.0, 5.0]
    g = Grid.new([0.0, 1.0, 2.0], [1.0, 2.0, 3.0], [10, 20, 30])   # grid.rs:190-198
    assert g.get_last_cell().tolist() == [10.0, 41.0, 92.0]


def test_triangle_count(lib):
    # lib.rs:175-193
    for n in range(0, 12):
        assert lib.m2s_triangle_count(99, n, 1, 0) == len(orc.get_triangles(99, list(range(n)), 0))
        assert lib.m2s_triangle_count(99, n, 1, 1) == len(orc.get_triangles(99, list(range(n)), 1))
        assert lib.m2s_triangle_count(n, 0, 0, 0) == len(orc.get_triangles(n, None, 0))
        assert lib.m2s_triangle_count(n, 0, 0, 1) == len(orc.get_triangles(n, None, 1))


def test_interleaved_slab_needs_whole_cut_list_waves_per_chunk():
    """m2s_interleaved_slab (host-only): a chunk is a power of two and a multiple of 4 packet bricks along x, else the shard falls
    back to its contiguous slab.  (Round 2 accepted a chunk of ONE brick — cell_size (0.125, 1, 1) has 16-layer bricks, nx = 128 on
    4 shards gives 16-layer chunks — and the peer push then walked two chunks as one contiguous range.)"""
    from mesh_to_sdf_amd import interleaved_slab

    cubic = Grid.new([0, 0, 0], [1, 1, 1], [256, 64, 64])
    assert interleaved_slab(cubic, 4, 1) == (32, 64, 128)            # chunks of 32 layers = 8 bricks of 4
    assert interleaved_slab(cubic, 8, 3) == (48, 64, 128)            # 16 layers = exactly 4 bricks
    assert interleaved_slab(Grid.new([0, 0, 0], [1, 1, 1], [128, 64, 64]), 8, 3)[2] == 0   # 8-layer chunks: 2 bricks -> contiguous
    thin = Grid.new([0, 0, 0], [0.125, 1, 1], [128, 64, 64])         # bricks of 16 x 2 x 2 cells
    assert interleaved_slab(thin, 4, 1) == (32, 64, 0)               # 16-layer chunks would be one brick: contiguous slab instead
    assert interleaved_slab(Grid.new([0, 0, 0], [0.125, 1, 1], [1024, 64, 64]), 4, 1) == (128, 256, 512)   # 128 layers = 8 bricks


def test_balanced_slabs_equalise_a_known_cost_profile():
    """m2s_balanced_slabs (host-only): boundaries of equal cost under a piecewise-constant density; whole units; every shard keeps
    at least one unit; degenerate inputs fall back to even slabs; bad inputs are M2S_ERR_BAD_ARG."""
    from mesh_to_sdf_amd import balanced_slabs, slab_bounds

    nx, n = 512, 8
    even = [slab_bounds(nx, n, k)[0] for k in range(n)] + [nx]
    assert balanced_slabs(nx, 4, even, [1.0] * n) == even                       # uniform cost: nothing moves
    # cost density 1 on the outer quarters, 3 on the middle half (what a round body in its box looks like)
    dens = lambda x: 3.0 if nx // 4 <= x < 3 * nx // 4 else 1.0
    cost = [sum(dens(x) for x in range(even[k], even[k + 1])) for k in range(n)]
    new = balanced_slabs(nx, 4, even, cost)
    assert new[0] == 0 and new[-1] == nx and all(b % 4 == 0 for b in new) and all(new[k + 1] > new[k] for k in range(n))
    share = [sum(dens(x) for x in range(new[k], new[k + 1])) for k in range(n)]
    assert max(share) / min(share) < 1.12 < max(cost) / min(cost)               # 3.0 before, within a unit of layers after
    again = balanced_slabs(nx, 4, new, share)                                   # a fixed point up to rounding
    assert max(abs(a - b) for a, b in zip(again, new)) <= 4
    # everything measured on one shard: the others still get a unit each
    lop = balanced_slabs(64, 4, [0, 16, 32, 48, 64], [0.0, 0.0, 5.0, 0.0])
    assert lop[0] == 0 and lop[-1] == 64 and all(lop[k + 1] - lop[k] >= 4 for k in range(4))
    assert balanced_slabs(64, 4, [0, 16, 32, 48, 64], [0.0] * 4) == [0, 16, 32, 48, 64]      # nothing measured
    assert balanced_slabs(10, 4, [0, 3, 6, 8, 10], [1.0, 2.0, 3.0, 4.0]) == [0, 3, 6, 8, 10]  # 3 units for 4 shards: even slabs
    assert balanced_slabs(7, 0, [0, 7], [2.0]) == [0, 7]
    for bad in (lambda: balanced_slabs(64, 4, [1, 16, 32, 48, 64], [1.0] * 4), lambda: balanced_slabs(64, 4, [0, 32, 16, 48, 64], [1.0] * 4),
                lambda: balanced_slabs(64, 4, [0, 16, 32, 48, 64], [1.0, float("nan"), 1.0, 1.0])):
        with pytest.raises(M2SError) as e:
            bad()
        assert e.value.code == _lib.ERR_BAD_ARG


def test_leaf_size_rules(lib):
    """The leaf size a call asks of its tree (distance.hip grid_leaf_max / query_leaf_max; DESIGN.md section 4 "Leaf size by grid", section 9): host
    arithmetic of the library itself through the test hook m2s_debug_leaf_sizes."""
    from mesh_to_sdf_amd import Grid

    fn = lib.m2s_debug_leaf_sizes
    fn.restype = C.c_int
    out = (C.c_uint32 * 3)()

    def ask(n, n_tris, n_q=0):
        g = Grid.from_bounding_box([0.0, 0.0, 0.0], [1.0, 1.0, 1.0], [n, n, n])
        assert fn(C.byref(g._g), C.c_size_t(n_tris), C.c_size_t(n_q), out) == 0
        return out[0], out[1], out[2]

    # grids: triangles per 4^3 brick of the whole grid < 0.6 -> 2, < 3 -> 4, < 40 -> 8, else 16
    assert ask(512, 100000)[0] == 2          # 0.048 (headline)
    assert ask(256, 100000)[0] == 2          # 0.38
    assert ask(192, 100000)[0] == 4          # 0.90
    assert ask(128, 100000)[0] == 8          # 3.05
    assert ask(64, 100000)[0] == 8           # 24.4
    assert ask(32, 100000)[0] == 16          # 195
    assert ask(512, 1000000)[0] == 2         # 0.48
    # queries: lane walk (leaves of 2) below 2.5 queries per triangle; packets < 50 per triangle -> 8, < 500 -> 4, else 2
    assert ask(64, 100000, 100000)[1:] == (2, 1)
    assert ask(64, 100000, 300000)[1:] == (8, 0)
    assert ask(64, 100000, 10000000)[1:] == (4, 0)
    assert ask(64, 11200, 10000000)[1:] == (2, 0)


def test_cut_list_words_are_supersets_for_every_tree_size(lib):
    """distance.hip CutList: a list word carries a range's start exactly and its length as a small float rounded UP — the walk may take a
    superset of a subtree range, never less.  Host arithmetic of the library itself (test hook m2s_debug_cut_code), every width of the
    start field from 1 to 26 bits (1 … 2^25 triangles), lengths around every power of two and random ones."""
    fn = lib.m2s_debug_cut_code
    fn.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]
    fn.restype = C.c_int
    out = (C.c_uint32 * 3)()
    rng = np.random.default_rng(5)
    sizes = [1, 2, 3, 5, 1935, 22399, 199999, 1999999, (1 << 24) + 77, (1 << 26) - 1] + [int(rng.integers(1, 1 << 26)) for _ in range(20)]
    worst = {}
    for n in sizes:
        S = max(1, (n - 1).bit_length())
        M = 27 - S
        lens = {1, n, max(1, n // 2), max(1, n // 3)}
        for b in range(0, S + 1):
            for d in (-1, 0, 1, 2):
                v = (1 << b) + d
                if 1 <= v <= n:
                    lens.add(v)
        lens |= {int(x) for x in rng.integers(1, n + 1, 200)}
        for ln in sorted(lens):
            for start in {0, n - ln, int(rng.integers(0, n - ln + 1))}:
                assert fn(n, start, ln, out) == 0
                word, first, end = out[0], out[1], out[2]
                assert first == start
                assert start + ln <= end <= n, (n, start, ln, end)
                if start + ln < end < n and M >= 2:            # the excess stays below 2^-(M-1) of the length (lengths < 2^M are exact)
                    assert ln >= (1 << M) and (end - start - ln) * (1 << (M - 1)) <= ln, (n, start, ln, end, M)
                    worst[M] = max(worst.get(M, 0.0), (end - start - ln) / ln)
    assert fn(100, 90, 11, out) != 0 and fn(0, 0, 1, out) != 0 and fn(100, 0, 0, out) != 0
    assert worst                                               # some long ranges were rounded


def test_argument_errors_before_any_device_work(lib):
    v = np.zeros((3, 3), np.float32)
    with pytest.raises(M2SError) as e:
        generate_sdf(v, Topology(7, None), v, AccelerationMethod.RtreeBvh)
    assert e.value.code == _lib.ERR_BAD_ARG
    with pytest.raises(M2SError) as e:
        generate_sdf(v, Topology.TriangleList(None), v, AccelerationMethod(9))
    assert e.value.code == _lib.ERR_BAD_ARG
    # RtreeBvh on an empty mesh returns an empty vec (rtree_bvh.rs:104-106), Rtree panics (rtree.rs:117)
    assert generate_sdf(np.zeros((0, 3), np.float32), Topology.TriangleList(None), v, AccelerationMethod.RtreeBvh).size == 0
    with pytest.raises(M2SError) as e:
        generate_sdf(np.zeros((0, 3), np.float32), Topology.TriangleList(None), v, AccelerationMethod.Rtree)
    assert e.value.code == _lib.ERR_EMPTY_MESH
    assert generate_sdf(v, Topology.TriangleList(None), np.zeros((0, 3), np.float32)).size == 0
    assert generate_grid_sdf(v, Topology.TriangleList(None), Grid.new([0, 0, 0], [1, 1, 1], [0, 4, 4])).size == 0


def test_no_gpu_means_loud_failure(lib):
    if lib.m2s_device_count() > 0:
        pytest.skip("a GPU is present")
    v = np.array([[0.0, 1.0, 0.0], [1.0, 2.0, 3.0], [1.0, 3.0, 4.0]], np.float32)
    with pytest.raises(M2SError) as e:
        generate_sdf(v, Topology.TriangleList([0, 1, 2]), [[0.0, 0.0, 0.0]])
    assert e.value.code == _lib.ERR_HIP and "no CPU fallback" in str(e.value)
    with pytest.raises(M2SError):
        generate_grid_sdf(v, Topology.TriangleList([0, 1, 2]), Grid.from_bounding_box([0, 0, 0], [1, 1, 1], [4, 4, 4]))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "mesh_to_sdf_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", "Makefile")):
                txt = open(os.path.join(dp, f), errors="replace").read()
                assert "import oracle" not in txt and "from oracle" not in txt and "m2s_oracle" not in txt, f


def test_tuning_table_set_describe_and_errors(lib):
    """The knob table (csrc/tuning.h): every knob is listed, set and reset through the ABI, nested settings come back, an unknown name
    or an unparsable value is M2S_ERR_BAD_ARG with a message; the table and the DESIGN.md §9 table name the same knobs."""
    knobs = _lib.describe_knobs()
    hdr = open(os.path.join(ROOT, "mesh_to_sdf_amd", "csrc", "tuning.h")).read()
    documented = set(re.findall(r"//\s+(M2S_[A-Z_0-9]+)\b", hdr))
    assert set(knobs) == documented, (sorted(set(knobs) ^ documented))
    design = open(os.path.join(ROOT, "DESIGN.md")).read()
    section = design[design.index("## 9. Run-time knobs"):]
    for name in knobs:
        assert f"`{name}`" in section, f"{name} missing from DESIGN.md §9"
    assert float(knobs["M2S_SPLIT_PATIENCE"]) == 1.5 and int(knobs["M2S_LANE_WALK"]) == -1
    with _lib.knobs(M2S_LANE_WALK=1, M2S_BRUTE_MAX=0):
        inner = _lib.describe_knobs()
        assert int(inner["M2S_LANE_WALK"]) == 1 and float(inner["M2S_BRUTE_MAX"]) == 0.0
        with _lib.knobs(M2S_LANE_WALK=0):
            assert int(_lib.describe_knobs()["M2S_LANE_WALK"]) == 0
        assert int(_lib.describe_knobs()["M2S_LANE_WALK"]) == 1
    assert _lib.describe_knobs() == knobs
    assert lib.m2s_tuning_set(b"M2S_NO_SUCH_KNOB", b"1") == -1
    assert b"M2S_NO_SUCH_KNOB" in lib.m2s_last_error()
    assert lib.m2s_tuning_set(b"M2S_CUT_FAR", b"zero") == -1
    assert _lib.describe_knobs() == knobs
    # out-of-range values are clamped, not taken literally
    with _lib.knobs(M2S_SPLIT_ROUNDS=99):
        assert int(_lib.describe_knobs()["M2S_SPLIT_ROUNDS"]) == 6
    # one look at the environment in the whole library
    csrc = os.path.join(ROOT, "mesh_to_sdf_amd", "csrc")
    sites = [(f, n + 1) for f in sorted(os.listdir(csrc)) if f.endswith((".hip", ".cpp", ".h"))
             for n, line in enumerate(open(os.path.join(csrc, f))) if re.search(r"\bgetenv\s*\(", line)]
    assert sites == [("tuning.cpp", sites[0][1])], sites
