"""The build's sample sort (mesh_to_sdf_amd/csrc/lbvh_sort.hip.h) as a numpy model: the index rules of its three kernels — the bitonic
network with four elements per thread and round, the staggered regular samples, splitters by sample rank, the G - 1 element window that
decides a bucket's boundary in a tile, and the bucket bound G q + tiles (G - 1) that sizes the LDS — restated and checked on the CPU.
The kernels themselves are pinned by the golden trees (tests/test_gpu_build.py, -m gpu): this model is what their comments claim, in a
form that runs here.  Replaces, as behaviour, the sort inside the reference's per-call BVH build (generate/grid.rs:95-111)."""
import numpy as np
import pytest

G, CAP, PER = 32, 4096, 4          # SS_G, SS_CAP, SS_PER_TILE


def less(ka, ia, kb, ib):
    return (ka < kb) | ((ka == kb) & (ia < ib))


def cx(k, v, a, b, asc):
    gt = less(k[b], v[b], k[a], v[a])
    sw = gt == asc
    ia, ib = a[sw], b[sw]
    k[ia], k[ib] = k[ib].copy(), k[ia].copy()
    v[ia], v[ib] = v[ib].copy(), v[ia].copy()


def bitonic_rounds(k, v, n, k_first):
    """bitonic_rounds<THREADS>: phases k_first .. n, two sub-stages (j, j / 2) per round on quads, a single stage when j ends at 1."""
    kk = k_first
    while kk <= n:
        j = kk >> 1
        while j >= 2:
            h = j >> 1
            q = np.arange(n >> 2)
            low = q & (h - 1)
            i0 = ((q - low) << 2) | low
            i1, i2 = i0 + h, i0 + j
            i3 = i2 + h
            asc = (i0 & kk) == 0
            cx(k, v, i0, i2, asc); cx(k, v, i1, i3, asc); cx(k, v, i0, i1, asc); cx(k, v, i2, i3, asc)
            j >>= 2
        if j == 1:
            i0 = np.arange(n >> 1) << 1
            cx(k, v, i0, i0 + 1, (i0 & kk) == 0)
        kk <<= 1


def sample_offset(t, p):
    return (t * G) // p


def sample_sort(keys, W):
    n = len(keys)
    p = (n + W - 1) // W
    SP, Q, B = W // G, W // G // PER, p * PER
    assert p <= CAP // G - Q, "ss_max_tiles"
    tk = np.full(p * W, np.uint64(2 ** 64 - 1), dtype=np.uint64)
    tv = np.zeros(p * W, dtype=np.uint64)
    tk[:n] = keys
    tv[:n] = np.arange(n)
    tv[n:] = 0x80000000 | (np.arange(n, p * W) % W)              # padding of the last tile: behind every pair, still distinct
    for t in range(p):                                            # k_sort_tiles: quads presorted in registers, then phases 8 .. W
        k, v = tk[t * W:(t + 1) * W], tv[t * W:(t + 1) * W]
        for q4 in range(0, W, 4):
            o = np.lexsort((v[q4:q4 + 4], k[q4:q4 + 4]))
            if q4 & 4:
                o = o[::-1]
            k[q4:q4 + 4], v[q4:q4 + 4] = k[q4:q4 + 4][o], v[q4:q4 + 4][o]
        bitonic_rounds(k, v, W, 8)
        assert (np.lexsort((v, k)) == np.arange(W)).all()
    if p == 1:
        return tk[:n], tv[:n], 0
    sk = np.concatenate([tk[t * W + np.arange(SP) * G + sample_offset(t, p)] for t in range(p)])
    si = np.concatenate([tv[t * W + np.arange(SP) * G + sample_offset(t, p)] for t in range(p)])
    order = np.lexsort((si, sk))                                 # k_sort_rank: the rank of every sample among all samples
    rank = np.empty(p * SP, dtype=np.int64)
    rank[order] = np.arange(p * SP)
    split = [(sk[order[(b + 1) * Q - 1]], si[order[(b + 1) * Q - 1]]) for b in range(B - 1)]
    cmat = np.array([[int((rank[t * SP:(t + 1) * SP] <= (b + 1) * Q - 1).sum()) for b in range(B - 1)] for t in range(p)])
    outk, outv, largest = np.zeros(n, dtype=np.uint64), np.zeros(n, dtype=np.uint64), 0
    total = 0
    for b in range(B):                                           # k_sort_buckets
        lo, hi = np.zeros(p, dtype=np.int64), np.zeros(p, dtype=np.int64)
        for t in range(p):
            n_t = min(W, n - t * W)
            for side in (0, 1):
                bb = b - 1 + side
                if bb < 0:
                    cnt = 0
                elif bb >= B - 1:
                    cnt = n_t
                else:
                    c = cmat[t, bb]
                    ws = 0 if c == 0 else (c - 1) * G + sample_offset(t, p) + 1
                    we = min(W, ws + G - 1)                      # one half-wave: G - 1 elements decide
                    cnt = ws + int((~less(split[bb][0], split[bb][1], tk[t * W + ws:t * W + we], tv[t * W + ws:t * W + we])).sum())
                    assert cnt == int((~less(split[bb][0], split[bb][1], tk[t * W:(t + 1) * W], tv[t * W:(t + 1) * W])).sum())
                    cnt = min(cnt, n_t)
                (hi if side else lo)[t] = cnt
        m, goff = int((hi - lo).sum()), int(lo.sum())
        assert m <= G * Q + p * (G - 1) <= CAP                   # the bound the LDS is sized by
        largest = max(largest, m)
        k = np.concatenate([tk[t * W + lo[t]:t * W + hi[t]] for t in range(p)])
        v = np.concatenate([tv[t * W + lo[t]:t * W + hi[t]] for t in range(p)])
        if m > 1:
            N = 1 << int(m - 1).bit_length()
            kk, vv = np.full(N, np.uint64(2 ** 64 - 1), dtype=np.uint64), np.full(N, 0xffffffff, dtype=np.uint64)
            kk[:m], vv[:m] = k, v
            bitonic_rounds(kk, vv, N, 2)
            k, v = kk[:m], vv[:m]
        outk[goff:goff + m], outv[goff:goff + m] = k, v
        total += m
    assert total == n
    return outk, outv, largest


CASES = [(5000, 1024, "random"), (4097, 1024, "few distinct keys"), (3000, 1024, "one key"), (2049, 2048, "few distinct keys"),
         (20000, 2048, "clustered"), (1000, 1024, "random"), (33, 1024, "random")]


@pytest.mark.parametrize("n,W,kind", CASES)
def test_sample_sort_model_is_the_stable_sort(n, W, kind):
    rng = np.random.default_rng(n)
    if kind == "random":
        keys = rng.integers(0, 2 ** 63, n, dtype=np.uint64)
    elif kind == "few distinct keys":
        keys = rng.integers(0, 50, n, dtype=np.uint64)
    elif kind == "one key":
        keys = np.full(n, 7, dtype=np.uint64)
    else:
        keys = (rng.integers(0, 2 ** 20, n, dtype=np.uint64) ** 2) // np.uint64(3)
    k, v, _ = sample_sort(keys, W)
    want = np.lexsort((np.arange(n), keys))                       # = a stable sort of the keys
    assert (v == want).all() and (k == keys[want]).all()


def test_staggered_samples_keep_random_order_buckets_even():
    """Triangles in random order give every tile the same key distribution: with samples at the END of every gap all tiles' first
    G - 1 elements fell to the first bucket (2 420 of 100 000 where the mean is 255); staggered offsets keep the buckets even."""
    rng = np.random.default_rng(3)
    keys = rng.integers(0, 2 ** 63, 30000, dtype=np.uint64)
    _, _, largest = sample_sort(keys, 1024)
    assert largest <= 3 * 256
