"""Two rules of the build's hierarchy kernels (mesh_to_sdf_amd/csrc/bvh.hip) as plain-Python models, checked against the definitions
they replace:
  * k_emit places a node in pre-order at 2 * first + (left turns on the way down from the root) and takes the left turns as the number
    of prefix-minimum records of the adjacent prefixes d[last], d[last + 1], ... (struct Recs) instead of walking parent links;
  * k_roots_from_keys finds the treelet roots — the maximal radix-tree nodes of at most 64 triangles — per SPLIT from the nearest smaller
    adjacent prefixes on either side, instead of growing a node around every position.
The kernels themselves are pinned by the golden trees (tests/test_gpu_build.py, -m gpu).  Replaces, as behaviour, the hierarchy of the
reference's per-call BVH build (generate/grid.rs:95-111)."""
import numpy as np
import pytest

TREELET_MAX = 64


def adjacent_prefixes(keys):
    """d[j] = common prefix of keys j and j + 1 (64-bit keys; equal keys are told apart by position), -1 past the end."""
    n = len(keys)
    d = np.full(n, -1, dtype=np.int64)
    for j in range(n - 1):
        a, b = int(keys[j]), int(keys[j + 1])
        d[j] = 64 - (a ^ b).bit_length() if a != b else 64 + 32 - (j ^ (j + 1)).bit_length()
    return d


def radix_tree(d):
    """The binary radix tree over the positions, by its definition: a node [l, r] splits where the adjacent prefix is smallest."""
    n = len(d)
    nodes = []                                                   # (first, last, slot, lefts) in pre-order

    def rec(l, r, lefts):
        slot = len(nodes)
        nodes.append((l, r, slot, lefts))
        if l == r:
            return
        g = l + int(np.argmin(d[l:r]))                           # the split: the smallest prefix inside is unique (see bvh.hip)
        rec(l, g, lefts + 1)
        rec(g + 1, r, lefts)

    import sys
    sys.setrecursionlimit(10000)
    rec(0, n - 1, 0)
    return nodes


def make_keys(kind, n, rng):
    if kind == "random":
        k = rng.integers(0, 2 ** 63, n, dtype=np.uint64)
    elif kind == "few distinct":
        k = rng.integers(0, 9, n, dtype=np.uint64)
    elif kind == "all equal":
        k = np.full(n, 5, dtype=np.uint64)
    else:                                                        # clustered: long common prefixes, a few outliers
        k = (rng.integers(0, 2 ** 12, n, dtype=np.uint64) << np.uint64(20)) | np.uint64(1 << 50)
        k[:: max(n // 7, 1)] = rng.integers(0, 2 ** 63, len(k[:: max(n // 7, 1)]), dtype=np.uint64)
    return np.sort(k)


@pytest.mark.parametrize("kind", ["random", "few distinct", "all equal", "clustered"])
@pytest.mark.parametrize("n", [1, 2, 3, 64, 65, 300, 1500])
def test_preorder_slot_from_prefix_minimum_records(kind, n):
    keys = make_keys(kind, n, np.random.default_rng(n))
    d = adjacent_prefixes(keys)
    # records of d[r ..]: the values that are smaller than everything before them, scanning right from r (the last position has none)
    lefts_of_last = np.zeros(n, dtype=np.int64)
    for r in range(n):
        cur, cnt = None, 0
        for j in range(r, n - 1):
            if cur is None or d[j] < cur:
                cur, cnt = d[j], cnt + 1
        lefts_of_last[r] = cnt
    for first, last, slot, lefts in radix_tree(d):
        assert lefts_of_last[last] == lefts
        assert 2 * first + lefts_of_last[last] == slot


def recs_join(a, b):
    """records(I J) = records(I) | (records(J) & bits below the smallest of I) on 96-bit sets (bvh.hip recs_join)."""
    below = (a & -a) - 1 if a else (1 << 96) - 1
    return a | (b & below)


@pytest.mark.parametrize("n", [5, 700, 1300])
def test_record_sets_scan_like_the_kernel(n):
    """Block-local suffix scan (512 positions) joined with the records of everything right of the block = the definition."""
    keys = make_keys("random", n, np.random.default_rng(11 + n))
    d = adjacent_prefixes(keys)
    leaf = [(1 << int(x)) if x >= 0 else 0 for x in d]
    BLOCK = 512
    nb = (n + BLOCK - 1) // BLOCK
    local, total = [0] * n, [0] * nb
    for b in range(nb):
        acc = 0
        for j in range(min(n, (b + 1) * BLOCK) - 1, b * BLOCK - 1, -1):
            acc = recs_join(leaf[j], acc)
            local[j] = acc
        total[b] = acc
    carry, acc = [0] * nb, 0
    for b in range(nb - 1, -1, -1):
        carry[b] = acc
        acc = recs_join(total[b], acc)
    for r in range(n):
        cur, cnt = None, 0
        for j in range(r, n - 1):
            if cur is None or d[j] < cur:
                cur, cnt = d[j], cnt + 1
        assert bin(recs_join(local[r], carry[r // BLOCK])).count("1") == cnt


@pytest.mark.parametrize("kind", ["random", "few distinct", "all equal", "clustered"])
@pytest.mark.parametrize("n", [2, 3, 64, 65, 130, 900])
def test_treelet_roots_by_split(kind, n):
    keys = make_keys(kind, n, np.random.default_rng(100 + n))
    d = adjacent_prefixes(keys)
    nodes = radix_tree(d)
    size = {(f, l): l - f + 1 for f, l, _, _ in nodes}
    # definition: the nodes of 3 .. 64 triangles whose parent holds more than 64 (or that are the whole tree)
    parent = {}
    stack = []
    for f, l, _, _ in nodes:                                     # pre-order: the parent is the nearest enclosing node on the stack
        while stack and not (stack[-1][0] <= f and l <= stack[-1][1]):
            stack.pop()
        parent[(f, l)] = stack[-1] if stack else None
        stack.append((f, l))
    want = sorted((f, l - f + 1) for (f, l) in size if 3 <= size[(f, l)] <= TREELET_MAX and (parent[(f, l)] is None or size[parent[(f, l)]] > TREELET_MAX))
    # the kernel's rule, per split j: the node reaches from the nearest smaller prefix on the left (exclusive) to the one on the right
    dd = lambda i: d[i] if 0 <= i < n else -1                    # noqa: E731
    def left_of(i, v):
        p = i - 1
        while p >= 0 and dd(p) >= v:
            p -= 1
        return p
    def right_of(i, v):
        p = i + 1
        while p < n - 1 and dd(p) >= v:
            p += 1
        return p
    got = []
    for j in range(n - 1):
        L, R = left_of(j, d[j]), right_of(j, d[j])
        s = R - L
        if not (3 <= s <= TREELET_MAX):
            continue
        dl, dr = dd(L), dd(R)
        root = True
        if dl >= 0 or dr >= 0:
            q, vq = (L, dl) if dl > dr else (R, dr)
            root = right_of(q, vq) - left_of(q, vq) > TREELET_MAX
        if root:
            got.append((L + 1, s))
    assert sorted(got) == want


def test_treelet_rank_by_sorting_composites_equals_counting_predecessors():
    """k_treelet_lanes moves every item of an open segment to the lane of its rank by (key, position): for long segments by sorting the
    composites (segment start, key, lane) over the whole wave and pulling, for short ones by counting predecessors and pushing.  Both
    must give the same arrangement; items of closed segments stay."""
    rng = np.random.default_rng(5)
    for _ in range(200):
        cuts = sorted(set(rng.integers(1, 64, rng.integers(0, 12)).tolist()))
        starts = [0] + cuts
        ends = cuts + [64]
        s = np.zeros(64, dtype=np.int64); e = np.zeros(64, dtype=np.int64); is_open = np.zeros(64, dtype=bool)
        for a, b in zip(starts, ends):
            s[a:b], e[a:b] = a, b
            is_open[a:b] = (b - a > 1) and rng.random() < 0.8
        key = rng.integers(-5, 5, 64)                                  # many ties: the position decides
        item = np.arange(64)
        # counting form (push)
        pushed = item.copy()
        for lane in range(64):
            rank = lane
            if is_open[lane]:
                rank = s[lane] + sum(1 for j in range(s[lane], e[lane]) if key[j] < key[lane] or (key[j] == key[lane] and j < lane))
            pushed[rank] = item[lane]
        # sorting form (pull)
        ukey = np.where(is_open, key + 2 ** 31, 0)
        comp = (s << 38) | (ukey << 6) | np.arange(64)
        pulled = item[np.sort(comp) & 63]
        assert (pushed == pulled).all()
