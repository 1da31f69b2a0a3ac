"""How much is a better tree worth?  Builds a full-sweep top-down tree over the triangles of the benchmark mesh on
the CPU (numpy), hands it to the library as prefix-free path codes (M2S_KEYS_FILE: the radix tree over those codes is
the same tree) and times the unchanged kernels against the Morton-key LBVH.

  python tools/exp_tree.py [mesh] [n] [cost ...]      cost in {sa, width, vol}
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def morton63(v, idx):
    """The builder's own keys (bvh.hip k_morton): 21 bits per axis of the triangle-box centre over the scene of centres."""
    tri = v[idx.reshape(-1, 3)].astype(np.float32)
    c = (0.5 * (tri.min(1) + tri.max(1))).astype(np.float32)
    lo, hi = c.min(0), c.max(0)
    u = np.clip((c - lo) / np.where(hi > lo, hi - lo, 1), 0, 1)
    q = np.minimum((u * 2097152.0).astype(np.uint64), 2097151)
    def expand(x):
        x = x & np.uint64(0x1fffff)
        x = (x | x << np.uint64(32)) & np.uint64(0x1f00000000ffff)
        x = (x | x << np.uint64(16)) & np.uint64(0x1f0000ff0000ff)
        x = (x | x << np.uint64(8)) & np.uint64(0x100f00f00f00f00f)
        x = (x | x << np.uint64(4)) & np.uint64(0x10c30c30c30c30c3)
        x = (x | x << np.uint64(2)) & np.uint64(0x1249249249249249)
        return x
    return (expand(q[:, 0]) << np.uint64(2)) | (expand(q[:, 1]) << np.uint64(1)) | expand(q[:, 2])


def sweep_tree(v, idx, cost="sa", max_depth=62, top_levels=None, bottom_size=None):
    """top_levels = K: sweep splits for the first K levels only, Morton order (the LBVH) below them.
    bottom_size = G: the LBVH's own splits (highest differing Morton bit) down to nodes of at most G triangles, sweep
    splits inside those."""
    tri = v[idx.reshape(-1, 3)]
    lo, hi = tri.min(1).astype(np.float64), tri.max(1).astype(np.float64)
    cen = 0.5 * (lo + hi)
    n = len(tri)
    keys = np.zeros(n, dtype=np.uint64)

    def measure(ext):
        if cost == "sa":
            return ext[:, 0] * ext[:, 1] + ext[:, 1] * ext[:, 2] + ext[:, 2] * ext[:, 0]
        if cost in ("width", "width1"):
            return ext.sum(1)
        if cost == "diag1":
            return np.sqrt((ext * ext).sum(1))
        if cost == "max1":
            return ext.max(1)
        if cost == "sa1":
            return ext[:, 0] * ext[:, 1] + ext[:, 1] * ext[:, 2] + ext[:, 2] * ext[:, 0]
        if cost == "sawidth":   # box grown by a margin: what a query ball of that radius sees
            m = margin
            e = ext + 2 * m
            return e[:, 0] * e[:, 1] + e[:, 1] * e[:, 2] + e[:, 2] * e[:, 0]
        raise ValueError(cost)

    margin = 0.02 * float((hi.max(0) - lo.min(0)).max())
    stack = [(np.arange(n), 0, 0)]
    depth_max = 0
    mort = morton63(v, idx) if (top_levels is not None or bottom_size is not None) else None
    if bottom_size is not None and bottom_size < 0:
        # fixed chunks of |bottom_size| consecutive triangles of the Morton order: the chunk number is the top of the key
        # (the radix tree over chunk numbers splits at index midpoints), sweep splits inside a chunk
        G = -bottom_size
        order = np.argsort(mort, kind="stable")
        nchunks = (n + G - 1) // G
        cbits = max(1, int(nchunks - 1).bit_length())
        stack = [(order[c * G:(c + 1) * G], c, cbits) for c in range(nchunks)]
        bottom_size = None
    elif bottom_size is not None:
        order = np.argsort(mort, kind="stable")
        stack = [(order, 0, 0)]
    while stack:
        ids, path, depth = stack.pop()
        m = len(ids)
        if bottom_size is not None and m > bottom_size:
            # the radix-tree split of a Morton-sorted range: where the highest differing bit flips (ties: the middle)
            a, b = int(mort[ids[0]]), int(mort[ids[-1]])
            if a == b:
                cut = m // 2
            else:
                bit = (a ^ b).bit_length() - 1
                cut = int(np.searchsorted((mort[ids] >> np.uint64(bit)) & np.uint64(1), 1))
            stack.append((ids[cut:], (path << 1) | 1, depth + 1))
            stack.append((ids[:cut], path << 1, depth + 1))
            continue
        if top_levels is not None and depth >= top_levels and m > 1:
            keys[ids] = (np.uint64(path) << np.uint64(64 - depth)) | (mort[ids] >> np.uint64(depth + 1)) if depth else mort[ids]
            continue
        if m == 1:
            keys[ids[0]] = np.uint64(path << (64 - depth)) if depth else np.uint64(0)
            depth_max = max(depth_max, depth)
            continue
        best = None
        if depth < max_depth - 20 or m <= 2:
            axes = range(3)
            if cost.endswith("1"):   # sweep along the widest axis of the centroids only
                ce = cen[ids]; axes = [int(np.argmax(ce.max(0) - ce.min(0)))]
            for ax in axes:
                o = ids[np.argsort(cen[ids, ax], kind="stable")]
                llo = np.minimum.accumulate(lo[o], 0); lhi = np.maximum.accumulate(hi[o], 0)
                rlo = np.minimum.accumulate(lo[o][::-1], 0)[::-1]; rhi = np.maximum.accumulate(hi[o][::-1], 0)[::-1]
                k = np.arange(1, m)
                c = measure(lhi[:-1] - llo[:-1]) * k + measure(rhi[1:] - rlo[1:]) * (m - k)
                j = int(np.argmin(c))
                if best is None or c[j] < best[0]:
                    best = (c[j], o, j + 1)
            _, o, cut = best
        else:   # depth guard: median split along the widest axis
            ext = hi[ids].max(0) - lo[ids].min(0)
            o = ids[np.argsort(cen[ids, int(np.argmax(ext))], kind="stable")]
            cut = m // 2
        stack.append((o[cut:], (path << 1) | 1, depth + 1))
        stack.append((o[:cut], path << 1, depth + 1))
    return keys, depth_max


if __name__ == "__main__":
    import torch
    from mesh_to_sdf_amd import Grid, Topology, SignMethod, M2STimings, generate_grid_sdf, meshes
    mesh = sys.argv[1] if len(sys.argv) > 1 else "blob-100k"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    costs = sys.argv[3:] or ["sa", "width", "sawidth"]
    v, idx = meshes.named(mesh); lo, hi = meshes.extended_bbox(v, 0.1)
    g = Grid.from_bounding_box(lo, hi, [n] * 3)
    dv = torch.as_tensor(v, device="cuda"); di = torch.as_tensor(idx.astype(np.int64), device="cuda").to(torch.int32)

    def run(tag):
        out = torch.empty(n ** 3, device="cuda"); best = None
        for r in range(3):
            t = M2STimings()
            generate_grid_sdf(dv, Topology.TriangleList(di), g, SignMethod.Raycast, timings=t, out=out)
            if best is None or t.distance_ms < best.distance_ms: best = t
        print(f"{tag}: build {best.accel_build_ms:.2f} seed {best.seed_ms:.2f} distance {best.distance_ms:.2f} ms", flush=True)
        return out

    ref = run("morton LBVH")
    for c in costs:
        top = bottom = None
        if ":" in c:
            c, top = c.split(":"); top = int(top)
        elif "/" in c:
            c, bottom = c.split("/"); bottom = int(bottom)
        elif "@" in c:
            c, bottom = c.split("@"); bottom = -int(bottom)
        t0 = time.time(); keys, d = sweep_tree(v, idx, c, top_levels=top, bottom_size=bottom); t1 = time.time()
        c = f"{c}, sweep for the top {top} levels, Morton below" if top is not None else c
        c = (f"{c}, Morton splits down to {bottom} triangles, sweep inside" if bottom > 0 else f"{c}, chunks of {-bottom} consecutive triangles of the Morton order, sweep inside") if bottom is not None else c
        path = f"/tmp/keys_{abs(hash(c))}.bin"; keys.tofile(path)
        os.environ["M2S_KEYS_FILE"] = path
        out = run(f"sweep tree cost={c} (depth {d}, built in {t1 - t0:.0f} s on the CPU)")
        print("   identical distances:", bool(torch.equal(out.abs(), ref.abs())), " identical signs:", bool(torch.equal(out < 0, ref < 0)), flush=True)
        del os.environ["M2S_KEYS_FILE"]
