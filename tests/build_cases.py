"""Meshes whose LBVH the build tests pin (tests/test_gpu_build.py, tools/make_build_golden.py): name -> (vertices, indices).
Deterministic: numpy Generators with fixed seeds, the synthetic meshes of mesh_to_sdf_amd.meshes, the bundled suzanne fixture."""
import os

import numpy as np

from mesh_to_sdf_amd import meshes


def _soup(v):
    v = np.ascontiguousarray(v, np.float32)
    return v, np.arange(v.shape[0], dtype=np.uint32)


def cases(big=True):
    for slices, stacks in [(3, 2), (4, 3), (8, 5), (33, 17), (48, 25), (128, 65), (250, 201)]:
        yield f"blob {slices}x{stacks}", *meshes.blob(slices, stacks)
    yield "sheet 101x77", *meshes.sheet(101, 77)
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "suzanne.npz"))
    yield "suzanne", z["vertices"].reshape(-1, 3), z["indices"].astype(np.uint32).reshape(-1)
    # sizes on both sides of the multiples of a segment-tree block (512 leaves) and of a sort tile
    rng = np.random.default_rng(7)
    for n in (1, 2, 3, 5, 64, 65, 511, 512, 513, 1023, 4095, 4096, 4097, 8191, 8193, 12289, 40000):
        v = rng.uniform(-1, 1, (3 * n, 3)).astype(np.float32) * np.float32(0.05) + np.repeat(rng.uniform(-1, 1, (n, 3)).astype(np.float32), 3, axis=0)
        yield f"{n} random triangles", *_soup(v)
    # many identical keys (the tie-break by position), zero-area and non-finite triangles, everything in one cell
    rng = np.random.default_rng(11)
    tri = rng.uniform(-1, 1, (1, 3, 3)).astype(np.float32)
    v = np.concatenate([np.repeat(tri, 700, axis=0).reshape(-1, 3), np.zeros((300, 3), np.float32), rng.uniform(-1, 1, (900, 3)).astype(np.float32)])
    v[2105] = np.float32(np.nan)
    v[2200] = np.float32(np.inf)
    yield "duplicates / degenerate / non-finite", *_soup(v)
    yield "all in one cell", *_soup((rng.uniform(-1, 1, (6000, 3)) * 1e-7).astype(np.float32) + np.float32(1000.0))
    # what the treelet roots from the sorted keys must get right: runs of EQUAL keys on both sides of the treelet size, clusters that
    # fill a node exactly to 64 / 65, far outliers that stretch the Morton cells, a regular lattice (many equal prefixes)
    rng = np.random.default_rng(23)
    tri = rng.uniform(-1, 1, (1, 3, 3)).astype(np.float32) * np.float32(0.01)
    for copies in (2, 3, 63, 64, 65, 66, 127, 128, 129, 200):
        v = np.concatenate([np.repeat(tri, copies, axis=0).reshape(-1, 3), rng.uniform(-1, 1, (3 * 37, 3)).astype(np.float32)])
        yield f"{copies} copies of one triangle + 37 others", *_soup(v)
    for per in (64, 65):
        parts = []
        for c in range(9):
            centre = rng.uniform(-100, 100, (1, 1, 3)).astype(np.float32)
            parts.append(centre + rng.uniform(-1e-4, 1e-4, (per, 3, 3)).astype(np.float32))
        yield f"clusters of {per}", *_soup(np.concatenate(parts).reshape(-1, 3))
    v = np.concatenate([rng.uniform(0, 1e-3, (3 * 500, 3)).astype(np.float32), np.float32([[1e6, 1e6, 1e6]] * 3), np.float32([[-1e6, 3.0, 2.0]] * 3)])
    yield "outliers", *_soup(v)
    gx, gy, gz = np.meshgrid(np.arange(12), np.arange(11), np.arange(10), indexing="ij")
    base = np.stack([gx, gy, gz], -1).reshape(-1, 1, 3).astype(np.float32)
    yield "lattice", *_soup((base + np.float32([[0, 0, 0], [0.3, 0, 0], [0, 0.3, 0]])).reshape(-1, 3))
    if big:
        yield "blob-1M", *meshes.named("blob-1M")
