"""How much is a better tree worth?  Builds a full-sweep top-down tree over the triangles of the benchmark mesh on
the CPU (numpy), hands it to the library as prefix-free path codes (M2S_KEYS_FILE: the radix tree over those codes is
the same tree) and times the unchanged kernels against the Morton-key LBVH.

  python tools/exp_tree.py [mesh] [n] [cost ...]      cost in {sa, width, vol}
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def sweep_tree(v, idx, cost="sa", max_depth=62):
    tri = v[idx.reshape(-1, 3)]
    lo, hi = tri.min(1).astype(np.float64), tri.max(1).astype(np.float64)
    cen = 0.5 * (lo + hi)
    n = len(tri)
    keys = np.zeros(n, dtype=np.uint64)

    def measure(ext):
        if cost == "sa":
            return ext[:, 0] * ext[:, 1] + ext[:, 1] * ext[:, 2] + ext[:, 2] * ext[:, 0]
        if cost == "width":
            return ext.sum(1)
        if cost == "sawidth":   # box grown by a margin: what a query ball of that radius sees
            m = margin
            e = ext + 2 * m
            return e[:, 0] * e[:, 1] + e[:, 1] * e[:, 2] + e[:, 2] * e[:, 0]
        raise ValueError(cost)

    margin = 0.02 * float((hi.max(0) - lo.min(0)).max())
    stack = [(np.arange(n), 0, 0)]
    depth_max = 0
    while stack:
        ids, path, depth = stack.pop()
        m = len(ids)
        if m == 1:
            keys[ids[0]] = np.uint64(path << (64 - depth)) if depth else np.uint64(0)
            depth_max = max(depth_max, depth)
            continue
        best = None
        if depth < max_depth - 20 or m <= 2:
            for ax in range(3):
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
        t0 = time.time(); keys, d = sweep_tree(v, idx, c); t1 = time.time()
        path = f"/tmp/keys_{c}.bin"; keys.tofile(path)
        os.environ["M2S_KEYS_FILE"] = path
        out = run(f"sweep tree cost={c} (depth {d}, built in {t1 - t0:.0f} s on the CPU)")
        print("   identical distances:", bool(torch.equal(out.abs(), ref.abs())), " identical signs:", bool(torch.equal(out < 0, ref < 0)), flush=True)
        del os.environ["M2S_KEYS_FILE"]
