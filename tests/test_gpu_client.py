"""Client steps either side of the generator (sdf.rs:62-72,120; sdf_program.rs:607-632) on the GPU against the
oracle: integer/index work is bit-exact, the merged positions are bit-exact f32.  Run with `-m gpu`."""
import numpy as np
import pytest

from mesh_to_sdf_amd import Grid, SignMethod, Topology, generate_grid_sdf, meshes
from mesh_to_sdf_amd.client import Sdf, merge_instances, order_cells_by_distance
from oracle import client_oracle as co

pytestmark = pytest.mark.gpu
F = np.float32


def nasty(n, seed):
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32).view(F).copy()
    a[: min(n, 8)] = np.array([0.0, -0.0, np.inf, -np.inf, 1e-45, -1e-45, 0.0, -0.0], F)[: min(n, 8)]
    return a


@pytest.mark.parametrize("n", [0, 1, 2, 3, 63, 64, 65, 1000, 4097, 100_003, 3_000_001])
def test_order_matches_the_oracle_bit_for_bit(n):
    d = nasty(n, n + 1)
    if n > 200:
        d[50:150] = d[150:250][: min(100, n - 150)] if n >= 250 else d[50:150]   # ties
    got, lim = order_cells_by_distance(d)
    assert got.dtype == np.uint32 and np.array_equal(got, co.order_cells(d))
    clean = d[~np.isnan(d)]
    if clean.size:
        want = co.minmax(clean) if n <= 5000 else (clean.min(), clean.max())
        assert lim[0] == want[0] and lim[1] == want[1]
    else:
        assert np.isnan(lim[0]) and np.isnan(lim[1])


def test_minmax_signed_zero_tie_rule():
    for d, neg_min in (([0.0, -0.0, 5.0, 5.0, -0.0], False), ([-0.0, 0.0, 2.0], True), ([3.0, 0.0, -0.0, 0.0], False)):
        d = np.array(d, F)
        _, lim = order_cells_by_distance(d)
        mn, mx = co.minmax(d)
        assert np.signbit(lim[0]) == np.signbit(mn) == neg_min and lim[1] == mx


def test_order_of_a_generated_grid_and_sdf_new(suzanne):
    """The client flow (sdf.rs:32-137): generate -> order -> limits, on device tensors."""
    import torch

    v, i = suzanne
    lo, hi = meshes.extended_bbox(v, 0.1)
    s = Sdf.new(torch.from_numpy(v).cuda(), torch.from_numpy(i.astype(np.int32)).cuda(), lo, hi, [48, 40, 56], SignMethod.Raycast)
    data = s.data.cpu().numpy()
    assert s.get_cell_count() == 48 * 40 * 56 == data.size
    want = co.order_cells(data)
    assert np.array_equal(s.ordered_indices.cpu().numpy().view(np.uint32), want)
    assert s.iso_limits == (data.min(), data.max())
    sd = data[want]
    assert (np.diff(sd) >= 0).all()
    # host-pointer flavour gives the same thing
    got, lim = order_cells_by_distance(data)
    assert np.array_equal(got, want) and lim == s.iso_limits


def test_order_512_cubed_properties():
    """BASELINE size (134 M cells, device resident): sortedness, permutation and stability as size-independent
    properties; exact equality with the oracle on a 16 M prefix-independent subsample of positions."""
    import torch

    n = 512**3
    gen = torch.Generator(device="cuda").manual_seed(5)
    # few distinct values per bucket -> many ties, so stability is really exercised
    d = (torch.randint(-2000, 2000, (n,), device="cuda", generator=gen, dtype=torch.int32).to(torch.float32) * 0.125)
    order, lim = order_cells_by_distance(d)
    idx = order.to(torch.int64) & 0xFFFFFFFF
    sd = d[idx]
    assert bool((sd[1:] >= sd[:-1]).all())
    same = sd[1:] == sd[:-1]
    assert bool((idx[1:][same] > idx[:-1][same]).all())           # stable: equal keys keep ascending index
    assert int(torch.bincount(idx, minlength=n).max()) == 1 and idx.numel() == n   # a permutation
    assert lim == (float(d.min()), float(d.max()))
    del sd, same, idx
    m = 1 << 22
    sub = d[:m].cpu().numpy()
    got, _ = order_cells_by_distance(d[:m])
    assert np.array_equal(got.cpu().numpy().view(np.uint32), co.order_cells(sub))


def rand_instances(seed, n_inst, stride_cols=3):
    rng = np.random.default_rng(seed)
    out = []
    for k in range(n_inst):
        nv = int(rng.integers(0, 400)) if k % 5 else int(rng.integers(0, 3))
        rows = rng.standard_normal((nv, stride_cols)).astype(F) * 3
        ni = int(rng.integers(0, 300)) * 3 if nv else 0
        idx = rng.integers(0, max(nv, 1), ni).astype(np.uint32)
        m = np.eye(4, dtype=F)
        m[:3, :3] = rng.standard_normal((3, 3)).astype(F)
        m[3, :3] = rng.standard_normal(3).astype(F) * 10     # column-major: row 3 here is w_axis
        out.append((rows, idx, m.reshape(-1).copy()))
    return out


@pytest.mark.parametrize("n_inst,cols", [(1, 3), (2, 3), (7, 3), (40, 3), (9, 8)])
def test_merge_instances_matches_the_oracle(n_inst, cols):
    inst = rand_instances(n_inst * 13 + cols, n_inst, cols)
    v, i, bbox = merge_instances(inst)
    wv, wi, wb = co.merge_instances([(r[:, :3], idx, m) for r, idx, m in inst])
    assert np.array_equal(v.view(np.uint32), wv.view(np.uint32)) and np.array_equal(i, wi)
    if wb is not None:
        assert np.array_equal(bbox.view(np.uint32), wb.view(np.uint32))


def test_merge_instances_device_tensors_and_generator_input(suzanne):
    """Two suzannes, one translated: the merged buffers feed generate_grid_sdf directly (device resident)."""
    import torch

    v, i = suzanne
    ident = np.eye(4, dtype=F).reshape(-1)
    shift = ident.copy(); shift[12:15] = [3.0, 0.5, -0.25]
    tv, ti = torch.from_numpy(v).cuda(), torch.from_numpy(i.astype(np.int32)).cuda()
    mv, mi, bbox = merge_instances([(tv, ti, ident), (tv, ti, shift)])
    wv, wi, wb = co.merge_instances([(v, i, ident), (v, i, shift)])
    assert mv.is_cuda and np.array_equal(mv.cpu().numpy().view(np.uint32), wv.view(np.uint32))
    assert np.array_equal(mi.cpu().numpy().view(np.uint32), wi) and np.array_equal(bbox, wb)
    grid = Grid.from_bounding_box(bbox[:3], bbox[3:], [40, 16, 16])
    a = generate_grid_sdf(mv, Topology.TriangleList(mi), grid, SignMethod.Normal).cpu().numpy()
    b = generate_grid_sdf(wv, Topology.TriangleList(wi), grid, SignMethod.Normal)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_merge_nothing():
    v, i, bbox = merge_instances([])
    assert v.shape == (0, 3) and i.size == 0 and np.isnan(bbox).all()


def test_load_gltf_end_to_end(tmp_path):
    """File -> instances (host) -> merged buffers + bbox (GPU) -> grid SDF, against the oracle chain; the client's
    `load_gltf` + `Sdf::new` flow (sdf_program.rs:597-641, sdf.rs:32-72)."""
    import os

    import oracle as orc
    from gltf_synth import zoo
    from mesh_to_sdf_amd.client import load_gltf
    from oracle import gltf_oracle as go

    gold = os.path.join(os.path.dirname(__file__), "golden", "gltf")
    p = tmp_path / "zoo.glb"
    p.write_bytes(zoo(4).glb())
    for path in (os.path.join(gold, "cube.glb"), os.path.join(gold, "suzanne.glb"), str(p)):
        _, models, inst = go.load(path)
        wv, wi, wb = co.merge_instances([(models[m][0], models[m][1], t.reshape(-1)) for m, t in inst])
        v, i, bbox = load_gltf(path)
        assert np.array_equal(v.view(np.uint32), wv.view(np.uint32)) and np.array_equal(i, wi)
        assert np.array_equal(bbox.view(np.uint32), wb.view(np.uint32))
        grid = Grid.from_bounding_box(bbox[:3], bbox[3:], [20, 24, 16])
        got = generate_grid_sdf(v, Topology.TriangleList(i), grid, SignMethod.Raycast)
        want = orc.generate_grid_sdf(wv, wi, grid.get_first_cell(), grid.get_cell_size(), grid.get_cell_count(),
                                     sign=int(SignMethod.Raycast), semantics=orc.EXACT)
        assert np.array_equal(got.view(np.uint32), np.asarray(want, np.float32).view(np.uint32))
