"""Deterministic synthetic meshes / grids / query sets for the parity tests and bench.py.

Definitions follow SURVEY.md §8(d) (no RNG, evaluated in float64, stored as float32):
  blob-100k  : UV sphere, slices=250, stacks=201 -> 100 000 tris, 50 002 verts, watertight,
               r(theta,phi) = 1 + 0.20 sin(3 theta) sin(2 phi) + 0.10 cos(5 phi) sin^2(theta)
  blob-1M    : slices=1000, stacks=501 -> 1 000 000 tris, plus 0.03 sin(17 theta) sin(13 phi)
  sheet-100k : 251x201 height field z = 0.3 sin(3x) cos(2y) over [-1,1]^2 -> 100 000 tris (open)
All are rotated by Rz(0.7) Ry(0.5) Rx(0.3) so no grid line runs exactly through a mesh
vertex or edge (the reference's ray test is strict, geo.rs:203).
"""
import numpy as np


def _rotation():
    ax, ay, az = 0.3, 0.5, 0.7
    rx = np.array([[1, 0, 0], [0, np.cos(ax), -np.sin(ax)], [0, np.sin(ax), np.cos(ax)]])
    ry = np.array([[np.cos(ay), 0, np.sin(ay)], [0, 1, 0], [-np.sin(ay), 0, np.cos(ay)]])
    rz = np.array([[np.cos(az), -np.sin(az), 0], [np.sin(az), np.cos(az), 0], [0, 0, 1]])
    return rz @ ry @ rx


def blob(slices=250, stacks=201, detail=False, rotate=True):
    """Closed star-shaped blob; returns (vertices f32 [V,3], indices u32 [3T]), outward CCW."""

    def radius(theta, phi):
        r = 1.0 + 0.20 * np.sin(3 * theta) * np.sin(2 * phi) + 0.10 * np.cos(5 * phi) * np.sin(theta) ** 2
        if detail:
            r = r + 0.03 * np.sin(17 * theta) * np.sin(13 * phi)
        return r

    i = np.arange(1, stacks, dtype=np.float64)
    j = np.arange(slices, dtype=np.float64)
    theta = (np.pi * i / stacks)[:, None]
    phi = (2 * np.pi * j / slices)[None, :]
    r = radius(theta, phi)
    ring = np.stack([r * np.sin(theta) * np.cos(phi), r * np.sin(theta) * np.sin(phi), r * np.cos(theta) + 0 * phi], -1)
    top = np.array([[0.0, 0.0, radius(0.0, 0.0)]])
    bot = np.array([[0.0, 0.0, -radius(np.pi, 0.0)]])
    verts = np.concatenate([top, ring.reshape(-1, 3), bot])
    if rotate:
        verts = verts @ _rotation().T

    def rid(ii, jj):  # ring ii in [0, stacks-2], column jj
        return 1 + ii * slices + (jj % slices)

    jj = np.arange(slices)
    last = 1 + (stacks - 1) * slices
    tris = [np.stack([np.zeros_like(jj), rid(0, jj), rid(0, jj + 1)], -1)]
    for ii in range(stacks - 2):
        u0, u1, l0, l1 = rid(ii, jj), rid(ii, jj + 1), rid(ii + 1, jj), rid(ii + 1, jj + 1)
        tris.append(np.stack([u0, l0, l1], -1))
        tris.append(np.stack([u0, l1, u1], -1))
    tris.append(np.stack([np.full_like(jj, last), rid(stacks - 2, jj + 1), rid(stacks - 2, jj)], -1))
    idx = np.concatenate(tris).astype(np.uint32).reshape(-1)
    return verts.astype(np.float32), idx


def sheet(nx=251, ny=201, rotate=True):
    """Open height-field surface (non-watertight)."""
    x = np.linspace(-1.0, 1.0, nx)[:, None] + 0 * np.zeros((1, ny))
    y = np.linspace(-1.0, 1.0, ny)[None, :] + 0 * np.zeros((nx, 1))
    z = 0.3 * np.sin(3 * x) * np.cos(2 * y)
    verts = np.stack([x, y, z], -1).reshape(-1, 3)
    if rotate:
        verts = verts @ _rotation().T
    ii, jj = np.meshgrid(np.arange(nx - 1), np.arange(ny - 1), indexing="ij")
    a = (ii * ny + jj).reshape(-1)
    b = ((ii + 1) * ny + jj).reshape(-1)
    c = ((ii + 1) * ny + jj + 1).reshape(-1)
    d = (ii * ny + jj + 1).reshape(-1)
    idx = np.concatenate([np.stack([a, b, c], -1), np.stack([a, c, d], -1)]).astype(np.uint32).reshape(-1)
    return verts.astype(np.float32), idx


def named(name):
    if name == "blob-100k":
        return blob(250, 201)
    if name == "blob-1M":
        return blob(1000, 501, detail=True)
    if name == "sheet-100k":
        return sheet(251, 201)
    if name == "blob-6k":
        return blob(60, 51)
    if name == "blob-11k":      # the size of the reference's criterion mesh (assets/knight.glb, 11 184 triangles)
        return blob(80, 71)
    raise KeyError(name)


def extended_bbox(vertices, frac=0.1):
    """min - frac*ext, max + frac*ext in float32 (mirrors generate/grid.rs:751-756)."""
    v = np.asarray(vertices, np.float32)
    lo, hi = v.min(0), v.max(0)
    ext = (hi - lo) * np.float32(frac)
    return (lo - ext).astype(np.float32), (hi + ext).astype(np.float32)


def grid_from_bounding_box(bmin, bmax, count):
    """Grid::from_bounding_box (grid.rs:59-74) in float32: returns (first_cell, cell_size, count)."""
    bmin = np.asarray(bmin, np.float32)
    bmax = np.asarray(bmax, np.float32)
    fc = np.asarray(count, np.float32)
    cell = ((bmax - bmin) / fc).astype(np.float32)
    first = (bmin + cell * np.float32(0.5)).astype(np.float32)
    return first, cell, tuple(int(c) for c in count)


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)).astype(np.uint64)
    z = x
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def uniform_queries(bmin, bmax, n, seed=0x5DF0_0000_0000_0000):
    """q[i][k] = min[k] + ext[k] * u(3i+k), u(j) = (splitmix64(seed + j) >> 40) * 2^-24."""
    with np.errstate(over="ignore"):
        j = np.arange(3 * n, dtype=np.uint64) + np.uint64(seed)
        u = (_splitmix64(j) >> np.uint64(40)).astype(np.float64) * 2.0 ** -24
    bmin = np.asarray(bmin, np.float64)
    ext = np.asarray(bmax, np.float64) - bmin
    return (bmin[None, :] + u.reshape(n, 3) * ext[None, :]).astype(np.float32)
