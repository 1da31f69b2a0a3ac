"""The lean LBVH build (round 3: own radix sort, fused setup / segment-tree kernels, bvh.hip) against the round-2 kernel
sequence (M2S_BUILD=0): the resident arrays of a persistent mesh — triangle records, pre-test planes, box nodes, oriented
bounds, centroids, slot table — must be the same bytes, i.e. the same tree.  Replaces, as behaviour, the reference's per-call
`Bvh::build_par` (generate/grid.rs:95-111); the distances never depended on the tree (the minimum is exact), its shape only
decides the walk's cost.  Needs a real MI355X: run with `-m gpu`."""
import os

import numpy as np
import pytest

from mesh_to_sdf_amd import Mesh, Topology, meshes

pytestmark = pytest.mark.gpu


def _digest(v, idx, build, topo=None):
    os.environ["M2S_BUILD"] = str(build)
    try:
        import torch

        dv = torch.as_tensor(np.ascontiguousarray(v, np.float32), device="cuda")
        if topo is None:
            topo = Topology.TriangleList(torch.as_tensor(idx.astype(np.int64), device="cuda").to(torch.int32))
        with Mesh(dv, topo) as m:
            return m.debug_digest()
    finally:
        os.environ.pop("M2S_BUILD", None)


NAMES = ["triangle records", "pre-test planes", "box nodes", "oriented bounds", "centroids", "slot table", "scene words", "count"]


def _same_tree(v, idx, what):
    old, new = _digest(v, idx, 0), _digest(v, idx, 1)
    bad = [NAMES[k] for k in range(8) if old[k] != new[k]]
    assert not bad, f"{what}: lean build differs from the round-2 build in {bad}"
    again = _digest(v, idx, 1)
    assert again == new, f"{what}: the lean build is not deterministic"


@pytest.mark.parametrize("slices,stacks", [(3, 2), (4, 3), (8, 5), (33, 17), (48, 25), (128, 65), (250, 201)])
def test_lean_build_same_tree_blobs(slices, stacks):
    v, idx = meshes.blob(slices, stacks)
    _same_tree(v, idx, f"blob {slices}x{stacks}")


def test_lean_build_same_tree_sheet_and_suzanne():
    v, idx = meshes.sheet(101, 77)
    _same_tree(v, idx, "sheet")
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "suzanne.npz"))
    _same_tree(z["vertices"].reshape(-1, 3), z["indices"].astype(np.uint32).reshape(-1), "suzanne")


def test_lean_build_same_tree_sizes_around_the_sort_tile():
    # one tile of the sort is 4096 pairs, one block of the segment tree 512 leaves: sizes on both sides of their multiples
    rng = np.random.default_rng(7)
    for n in (1, 2, 3, 5, 64, 65, 511, 512, 513, 1023, 4095, 4096, 4097, 8191, 8193, 12289, 40000):
        v = rng.uniform(-1, 1, (3 * n, 3)).astype(np.float32) * np.float32(0.05) + np.repeat(rng.uniform(-1, 1, (n, 3)).astype(np.float32), 3, axis=0)
        idx = np.arange(3 * n, dtype=np.uint32)
        _same_tree(v, idx, f"{n} random triangles")


def test_lean_build_same_tree_duplicates_and_degenerates():
    # many identical keys (the tie-break by position), zero-area and non-finite triangles, everything in one cell
    rng = np.random.default_rng(11)
    tri = rng.uniform(-1, 1, (1, 3, 3)).astype(np.float32)
    v = np.concatenate([np.repeat(tri, 700, axis=0).reshape(-1, 3),                       # 700 copies of one triangle
                        np.zeros((300, 3), np.float32),                                  # 100 point triangles at the origin
                        rng.uniform(-1, 1, (900, 3)).astype(np.float32)])
    v[2105] = np.float32(np.nan)
    v[2200] = np.float32(np.inf)
    idx = np.arange(v.shape[0], dtype=np.uint32)
    _same_tree(v, idx, "duplicates / degenerate / non-finite")
    v2 = (rng.uniform(-1, 1, (6000, 3)) * 1e-7).astype(np.float32) + np.float32(1000.0)   # 2000 triangles in a cube of 2e-7 at 1000
    _same_tree(v2, np.arange(6000, dtype=np.uint32), "all in one cell")


def test_lean_build_same_tree_adversarial_keys():
    """What the treelet roots from the sorted keys (k_roots_from_keys) must get right without the hierarchy: runs of EQUAL keys on both
    sides of the treelet size (the tie-break by position), clusters that fill a node exactly to 64 / 65, keys that differ only in their
    last bits, a few far outliers that stretch the Morton cells, a regular lattice of identical triangles (many equal prefixes)."""
    rng = np.random.default_rng(23)
    tri = rng.uniform(-1, 1, (1, 3, 3)).astype(np.float32) * np.float32(0.01)
    for copies in (2, 3, 63, 64, 65, 66, 127, 128, 129, 200):
        v = np.concatenate([np.repeat(tri, copies, axis=0).reshape(-1, 3), rng.uniform(-1, 1, (3 * 37, 3)).astype(np.float32)])
        _same_tree(v, np.arange(v.shape[0], dtype=np.uint32), f"{copies} copies of one triangle + 37 others")
    # clusters of exactly 64 / 65 near-identical centres, far apart
    for per in (64, 65):
        parts = []
        for c in range(9):
            centre = rng.uniform(-100, 100, (1, 1, 3)).astype(np.float32)
            parts.append(centre + rng.uniform(-1e-4, 1e-4, (per, 3, 3)).astype(np.float32))
        v = np.concatenate(parts).reshape(-1, 3)
        _same_tree(v, np.arange(v.shape[0], dtype=np.uint32), f"clusters of {per}")
    # two far outliers squeeze everything else into a corner of the Morton cube
    v = np.concatenate([rng.uniform(0, 1e-3, (3 * 500, 3)).astype(np.float32), np.float32([[1e6, 1e6, 1e6]] * 3), np.float32([[-1e6, 3.0, 2.0]] * 3)])
    _same_tree(v, np.arange(v.shape[0], dtype=np.uint32), "outliers")
    # a regular lattice of identical small triangles: long runs of equal key prefixes
    gx, gy, gz = np.meshgrid(np.arange(12), np.arange(11), np.arange(10), indexing="ij")
    base = np.stack([gx, gy, gz], -1).reshape(-1, 1, 3).astype(np.float32)
    v = (base + np.float32([[0, 0, 0], [0.3, 0, 0], [0, 0.3, 0]])).reshape(-1, 3)
    _same_tree(v, np.arange(v.shape[0], dtype=np.uint32), "lattice")


def test_lean_build_same_tree_blob_1m():
    v, idx = meshes.named("blob-1M")
    _same_tree(v, idx, "blob-1M")
