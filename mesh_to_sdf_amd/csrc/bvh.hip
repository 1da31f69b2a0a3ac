// bvh.hip — topology flatten + triangle records + LBVH in stackless pre-order layout (gfx950).
//
// Replaces, as behaviour, Topology::get_triangles (lib.rs:175-193) and the acceleration
// structures the reference builds inside every call (bvh 0.10 Bvh::build_par at
// generate/grid.rs:95-111, generic/bvh.rs:74; rstar bulk_load at generic/rtree.rs:111): they
// only select candidates, the kernels in distance.hip take the exact minimum.
//
// Pipeline (all on the call's stream, no host round trip):
//   k_tri_setup   : indices -> (a,b,c), degeneracy class, padded box (geo.rs:4-22), scene bounds
//   k_morton      : 63-bit Morton key of the box centre
//   rocprim radix sort (key,value)            — library sort, not on the parity path
//   k_karras      : Karras 2012 hierarchy over the sorted keys (ranges, children, parents)
//   k_seg_level   : segment tree of leaf boxes, one launch per level (fence-free refit)
//   k_emit        : node boxes by range query, pre-order index = 2*first + #left-turns, skip links
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <cstdio>
#include <functional>
#include <vector>

#include <rocprim/rocprim.hpp>

#include "../../include/m2s.h"
#include "common.h"
#include "geo.hip.h"

namespace m2s {

namespace {

struct Box {
  float mnx, mny, mnz, mxx, mxy, mxz;
};

__device__ __forceinline__ int ord(float f) {
  int i = __float_as_int(f);
  return i >= 0 ? i : i ^ 0x7fffffff;
}
__host__ __device__ __forceinline__ float unord(int i) {
  int b = i >= 0 ? i : i ^ 0x7fffffff;
#ifdef __HIP_DEVICE_COMPILE__
  return __int_as_float(b);
#else
  float f;
  memcpy(&f, &b, 4);
  return f;
#endif
}

__device__ __forceinline__ uint32_t load_index(const void* idx, int index_bytes, size_t i) {
  if (idx == nullptr) return (uint32_t)i;
  return index_bytes == 2 ? (uint32_t)((const uint16_t*)idx)[i] : ((const uint32_t*)idx)[i];
}

// One thread per triangle of Topology::get_triangles (lib.rs:175-193).
__global__ __launch_bounds__(256) void k_tri_setup(const float* __restrict__ verts, uint32_t n_verts,
                                                   const void* __restrict__ indices, int index_bytes, int topology,
                                                   uint32_t n_tris, TriRec* __restrict__ raw, Box* __restrict__ boxes,
                                                   float4* __restrict__ cen_raw,
                                                   int* __restrict__ scene /*6 final + 6 per block*/, int* __restrict__ err) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  int lo[3] = {INT32_MAX, INT32_MAX, INT32_MAX}, hi[3] = {INT32_MIN, INT32_MIN, INT32_MIN};
  if (t < n_tris) {
    const size_t base = topology == 0 ? (size_t)t * 3 : (size_t)t;  // list: tuples(); strip: tuple_windows()
    uint32_t i0 = load_index(indices, index_bytes, base), i1 = load_index(indices, index_bytes, base + 1),
             i2 = load_index(indices, index_bytes, base + 2);
    if (i0 >= n_verts || i1 >= n_verts || i2 >= n_verts) {
      atomicOr(err, ERRF_INDEX_OOB);  // the reference panics on vertices[i]
      i0 = i1 = i2 = 0;
    }
    f3 a = {0, 0, 0}, b = {0, 0, 0}, c = {0, 0, 0};
    if (n_verts) {
      a = mk3(verts[3 * (size_t)i0], verts[3 * (size_t)i0 + 1], verts[3 * (size_t)i0 + 2]);
      b = mk3(verts[3 * (size_t)i1], verts[3 * (size_t)i1 + 1], verts[3 * (size_t)i1 + 2]);
      c = mk3(verts[3 * (size_t)i2], verts[3 * (size_t)i2 + 1], verts[3 * (size_t)i2 + 2]);
    }
    f3 mn, mx;
    triangle_bounding_box(a, b, c, &mn, &mx);
    TriRec r;
    r.ax = a.x; r.ay = a.y; r.az = a.z; r.cls = tri_class(a, b, c);
    r.bx = b.x; r.by = b.y; r.bz = b.z; r.index = t;
    r.cx = c.x; r.cy = c.y; r.cz = c.z; r.pad0 = 0.0f;
    r.abx = b.x - a.x; r.aby = b.y - a.y; r.abz = b.z - a.z;
    r.acx = c.x - a.x; r.acy = c.y - a.y; r.acz = c.z - a.z;
    r.bcx = c.x - b.x; r.bcy = c.y - b.y; r.bcz = c.z - b.z;
    const f3 nr = cross3(mk3(r.abx, r.aby, r.abz), mk3(r.acx, r.acy, r.acz));   // triangle_normal, not normalised
    r.nrx = nr.x; r.nry = nr.y; r.nrz = nr.z;
    const float sx = 0.5f * (mn.x + mx.x), sy = 0.5f * (mn.y + mx.y), sz = 0.5f * (mn.z + mx.z);
    raw[t] = r;
    boxes[t] = {mn.x, mn.y, mn.z, mx.x, mx.y, mx.z};
    // centroid in INPUT order (k_emit writes the same values in sorted order): lets the seed passes start before the sort
    cen_raw[t] = make_float4((r.ax + r.bx + r.cx) * (1.0f / 3.0f), (r.ay + r.by + r.cy) * (1.0f / 3.0f),
                             (r.az + r.bz + r.cz) * (1.0f / 3.0f), 0.0f);
    const float cen[3] = {sx, sy, sz};
    for (int k = 0; k < 3; ++k)
      if (cen[k] == cen[k] && fabsf(cen[k]) < 3.0e38f) { lo[k] = ord(cen[k]); hi[k] = lo[k]; }
  }
  // block reduce (wave shuffles, then LDS); one partial per block, folded by k_scene_reduce:
  // no atomics, no serialisation on six hot addresses
  __shared__ int part[6][4];
  const int wv = threadIdx.x >> 6;
  for (int k = 0; k < 3; ++k) {
    int l = lo[k], h = hi[k];
    for (int off = 32; off > 0; off >>= 1) {
      l = min(l, __shfl_xor(l, off));
      h = max(h, __shfl_xor(h, off));
    }
    if ((threadIdx.x & 63) == 0) { part[k][wv] = l; part[3 + k][wv] = h; }
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    const int k = threadIdx.x;
    int v = part[k][0];
    for (int w = 1; w < 4; ++w) v = k < 3 ? min(v, part[k][w]) : max(v, part[k][w]);
    scene[6 + blockIdx.x * 6 + k] = v;   // partials live behind the 6 final values
  }
}

// Folds the per-block partials into scene[0..5] (order-encoded min xyz / max xyz of the box centres).
__global__ __launch_bounds__(256) void k_scene_reduce(int* __restrict__ scene, uint32_t n_blocks) {
  __shared__ int part[6][4];
  int lo[3] = {INT32_MAX, INT32_MAX, INT32_MAX}, hi[3] = {INT32_MIN, INT32_MIN, INT32_MIN};
  for (uint32_t b = threadIdx.x; b < n_blocks; b += 256)
    for (int k = 0; k < 3; ++k) {
      lo[k] = min(lo[k], scene[6 + b * 6 + k]);
      hi[k] = max(hi[k], scene[6 + b * 6 + 3 + k]);
    }
  const int wv = threadIdx.x >> 6;
  for (int k = 0; k < 3; ++k) {
    int l = lo[k], h = hi[k];
    for (int off = 32; off > 0; off >>= 1) {
      l = min(l, __shfl_xor(l, off));
      h = max(h, __shfl_xor(h, off));
    }
    if ((threadIdx.x & 63) == 0) { part[k][wv] = l; part[3 + k][wv] = h; }
  }
  __syncthreads();
  __shared__ int fin[6];
  if (threadIdx.x < 6) {
    const int k = threadIdx.x;
    int v = part[k][0];
    for (int w = 1; w < 4; ++w) v = k < 3 ? min(v, part[k][w]) : max(v, part[k][w]);
    scene[k] = v;
    fin[k] = v;
  }
  __syncthreads();
  if (threadIdx.x == 0) {   // scene[6]: largest finite |coordinate| as float bits (mesh_scale); the partials are dead now
    float s = 0.0f;
    for (int k = 0; k < 6; ++k) {
      const int i = fin[k];
      const float f = __int_as_float(i >= 0 ? i : i ^ 0x7fffffff);
      if (f == f && fabsf(f) < 3.0e38f) s = fmaxf(s, fabsf(f));
    }
    scene[6] = __float_as_int(s);
    scene[7] = 0;                          // number of treelet roots (k_treelet_roots counts into it)
  }
}

__device__ __forceinline__ uint64_t expand21(uint32_t v) {
  uint64_t x = v & 0x1fffffu;
  x = (x | x << 32) & 0x1f00000000ffffull;
  x = (x | x << 16) & 0x1f0000ff0000ffull;
  x = (x | x << 8) & 0x100f00f00f00f00full;
  x = (x | x << 4) & 0x10c30c30c30c30c3ull;
  x = (x | x << 2) & 0x1249249249249249ull;
  return x;
}

__global__ __launch_bounds__(256) void k_morton(const Box* __restrict__ boxes, uint32_t n_tris,
                                                const int* __restrict__ scene, uint64_t* __restrict__ keys,
                                                uint32_t* __restrict__ vals) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_tris) return;
  const Box bx = boxes[t];
  const float c[3] = {0.5f * (bx.mnx + bx.mxx), 0.5f * (bx.mny + bx.mxy), 0.5f * (bx.mnz + bx.mxz)};
  uint32_t q[3];
  for (int k = 0; k < 3; ++k) {
    const float lo = unord(scene[k]), hi = unord(scene[3 + k]);
    float u = (c[k] - lo) / (hi - lo);
    u = (u == u) ? fminf(fmaxf(u, 0.0f), 1.0f) : 0.0f;
    q[k] = min((uint32_t)(u * 2097152.0f), 2097151u);
  }
  keys[t] = (expand21(q[0]) << 2) | (expand21(q[1]) << 1) | expand21(q[2]);
  vals[t] = t;
}

// Karras 2012.  Keys may repeat; ties are broken by position, which keeps the tree well formed.
__device__ __forceinline__ int delta(const uint64_t* __restrict__ keys, int n, int i, int j) {
  if (j < 0 || j >= n) return -1;
  const uint64_t a = keys[i], b = keys[j];
  if (a == b) return 64 + __clz((uint32_t)i ^ (uint32_t)j);
  return __clzll((long long)(a ^ b));
}

// Node ids: internal i in [0, n-2], leaf k -> (n-1) + k.
__global__ __launch_bounds__(256) void k_karras(const uint64_t* __restrict__ keys, int n, int2* __restrict__ range,
                                                int2* __restrict__ child, int* __restrict__ parent) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n - 1) return;
  const int d = (delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1)) >= 0 ? 1 : -1;
  const int dmin = delta(keys, n, i, i - d);
  int lmax = 2;
  while (delta(keys, n, i, i + lmax * d) > dmin) lmax <<= 1;
  int l = 0;
  for (int t = lmax >> 1; t >= 1; t >>= 1)
    if (delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
  const int j = i + l * d;
  const int dnode = delta(keys, n, i, j);
  int s = 0;
  for (int t = (l + 1) >> 1;; t = (t + 1) >> 1) {
    if (delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
    if (t <= 1) break;
  }
  const int gamma = i + s * d + min(d, 0);
  const int first = min(i, j), last = max(i, j);
  const int left = (first == gamma) ? (n - 1) + gamma : gamma;
  const int right = (last == gamma + 1) ? (n - 1) + gamma + 1 : gamma + 1;
  range[i] = make_int2(first, last);
  child[i] = make_int2(left, right);
  parent[left] = i;
  parent[right] = -i - 2;  // negative: "I am a right child of i"  (root keeps its initial marker)
}

__device__ __forceinline__ Box box_union(Box a, Box b) {
  return {fminf(a.mnx, b.mnx), fminf(a.mny, b.mny), fminf(a.mnz, b.mnz),
          fmaxf(a.mxx, b.mxx), fmaxf(a.mxy, b.mxy), fmaxf(a.mxz, b.mxz)};
}

// Level 0 of the segment tree: leaf boxes in sorted order.
__global__ __launch_bounds__(256) void k_seg_level0(const Box* __restrict__ boxes, const uint32_t* __restrict__ order,
                                                    int n, Box* __restrict__ seg) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n) seg[k] = boxes[order[k]];
}
// Builds levels l+1, l+2, l+3 of the segment tree from level l in one launch: thread j owns entry j of
// level l+3 = the union of (up to) 8 entries of level l, and writes the intermediate entries on the way.
// LEAVES: level l is level 0 and does not exist yet — its entries are the leaf boxes in sorted order (boxes[order[k]]), written here too.
template <bool LEAVES>
__global__ __launch_bounds__(256) void k_seg_level3(Box* __restrict__ seg, uint32_t off0, uint32_t n0, uint32_t off1,
                                                    uint32_t n1, uint32_t off2, uint32_t n2, uint32_t off3, uint32_t n3,
                                                    const Box* __restrict__ boxes, const uint32_t* __restrict__ order) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;   // index at level l+3 (or the deepest level present)
  Box b1[4];
  bool have1[4];
  for (uint32_t a = 0; a < 4; ++a) {
    const uint32_t i1 = 4 * j + a;          // entry at level l+1
    have1[a] = i1 < n1;
    if (!have1[a]) continue;
    Box b, c;
    if (LEAVES) { b = boxes[order[2 * i1]]; seg[off0 + 2 * i1] = b; }
    else b = seg[off0 + 2 * i1];
    if (2 * i1 + 1 < n0) {
      if (LEAVES) { c = boxes[order[2 * i1 + 1]]; seg[off0 + 2 * i1 + 1] = c; }
      else c = seg[off0 + 2 * i1 + 1];
      b = box_union(b, c);
    }
    b1[a] = b;
    seg[off1 + i1] = b;
  }
  if (n2 == 0) return;
  Box b2[2];
  bool have2[2];
  for (uint32_t a = 0; a < 2; ++a) {
    const uint32_t i2 = 2 * j + a;          // entry at level l+2
    have2[a] = i2 < n2;
    if (!have2[a]) continue;
    Box b = b1[2 * a];
    if (have1[2 * a + 1]) b = box_union(b, b1[2 * a + 1]);
    b2[a] = b;
    seg[off2 + i2] = b;
  }
  if (n3 == 0 || j >= n3) return;
  Box b = b2[0];
  if (have2[1]) b = box_union(b, b2[1]);
  seg[off3 + j] = b;
}

struct SegLevels {
  uint32_t off[34];
  uint32_t cnt[34];
  int levels;
};

__device__ __forceinline__ Box seg_query(const Box* __restrict__ seg, const SegLevels& lv, int first, int last) {
  const float inf = __builtin_inff();
  Box acc = {inf, inf, inf, -inf, -inf, -inf};
  uint32_t lo = (uint32_t)first, hi = (uint32_t)last + 1u;
  int level = 0;
  while (lo < hi) {
    if (lo & 1u) { acc = box_union(acc, seg[lv.off[level] + lo]); ++lo; }
    if (hi & 1u) { --hi; acc = box_union(acc, seg[lv.off[level] + hi]); }
    lo >>= 1; hi >>= 1; ++level;
  }
  return acc;
}

// One thread per node (internal 0..n-2, leaves n-1..2n-2): pre-order slot, box, skip link,
// and for leaves the sorted triangle record.
__global__ __launch_bounds__(256) void k_emit(int n, const int2* __restrict__ range, const int* __restrict__ parent,
                                              const Box* __restrict__ seg, SegLevels lv,
                                              const uint32_t* __restrict__ order, const TriRec* __restrict__ raw,
                                              NodeRec* __restrict__ nodes, TriRec* __restrict__ tris,
                                              uint32_t* __restrict__ slot_first, float4* __restrict__ cen,
                                              TriPlanes* __restrict__ planes, uint32_t leaf_max, uint32_t* __restrict__ slot_of) {
  const int id = blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= 2 * n - 1) return;
  const bool leaf = id >= n - 1;
  int first, last;
  if (leaf) { first = last = id - (n - 1); }
  else { int2 r = range[id]; first = r.x; last = r.y; }
  // number of left turns on the root -> node path
  int lefts = 0, cur = id;
  while (true) {
    const int p = parent[cur];
    if (p == INT32_MIN) break;          // root marker
    if (p >= 0) { ++lefts; cur = p; }   // cur is a left child of p
    else cur = -p - 2;                  // right child
  }
  const uint32_t slot = 2u * (uint32_t)first + (uint32_t)lefts;
  const uint32_t cnt = (uint32_t)(last - first + 1);
  Box b = leaf ? seg[first] : seg_query(seg, lv, first, last);
  NodeRec nr;
  nr.mnx = b.mnx; nr.mny = b.mny; nr.mnz = b.mnz;
  nr.skip = slot + 2u * cnt - 1u;
  nr.mxx = b.mxx; nr.mxy = b.mxy; nr.mxz = b.mxz;
  // A subtree of at most `leaf_max` triangles is walked as ONE leaf: its triangles are contiguous in Morton
  // order ([first, first+cnt), cnt = (skip - slot + 1)/2) and `skip` already jumps over its descendants.
  nr.tri = cnt <= leaf_max ? first : -1;
  nodes[slot] = nr;
  slot_first[slot] = (uint32_t)first;
  if (leaf) {
    const TriRec r = raw[order[first]];
    tris[first] = r;
    slot_of[order[first]] = (uint32_t)first;   // input triangle -> its slot in the sorted arrays
    cen[first] = make_float4((r.ax + r.bx + r.cx) * (1.0f / 3.0f), (r.ay + r.by + r.cy) * (1.0f / 3.0f),
                             (r.az + r.bz + r.cz) * (1.0f / 3.0f), 0.0f);
    // leaf pre-test planes (common.h TriPlanes); all zero (= always evaluate) unless everything is well defined
    TriPlanes pl = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const f3 a = mk3(r.ax, r.ay, r.az), bq = mk3(r.bx, r.by, r.bz), cq = mk3(r.cx, r.cy, r.cz);
    const f3 n = cross3(mk3(r.abx, r.aby, r.abz), mk3(r.acx, r.acy, r.acz));
    const float nl = sqrtf(n.x * n.x + n.y * n.y + n.z * n.z);
    if (r.cls == TRI_REGULAR && nl > 1.0e-30f && nl < 3.0e38f) {
      const f3 nu = {n.x / nl, n.y / nl, n.z / nl};
      const f3 v[3] = {a, bq, cq};
      const f3 e[3] = {mk3(r.abx, r.aby, r.abz), mk3(r.bcx, r.bcy, r.bcz), mk3(a.x - cq.x, a.y - cq.y, a.z - cq.z)};  // ab, bc, ca
      float m[3][4];
      bool ok = true;
      const float vmax = fmaxf(fmaxf(fabsf(a.x), fmaxf(fabsf(a.y), fabsf(a.z))),
                               fmaxf(fmaxf(fabsf(bq.x), fmaxf(fabsf(bq.y), fabsf(bq.z))), fmaxf(fabsf(cq.x), fmaxf(fabsf(cq.y), fabsf(cq.z)))));
      for (int k = 0; k < 3; ++k) {
        f3 mk = cross3(e[k], nu);                                   // in the plane, perpendicular to edge k
        const float ml = sqrtf(mk.x * mk.x + mk.y * mk.y + mk.z * mk.z);
        if (!(ml > 1.0e-30f) || !(ml < 3.0e38f)) { ok = false; break; }
        mk = {mk.x / ml, mk.y / ml, mk.z / ml};
        const f3 opp = v[(k + 2) % 3];                               // the vertex not on edge k must be inside
        const f3 w0 = v[k], w1 = v[(k + 1) % 3];
        float o = fmaxf(mk.x * w0.x + mk.y * w0.y + mk.z * w0.z, mk.x * w1.x + mk.y * w1.y + mk.z * w1.z);
        if (mk.x * opp.x + mk.y * opp.y + mk.z * opp.z > o) { mk = {-mk.x, -mk.y, -mk.z}; o = fmaxf(mk.x * w0.x + mk.y * w0.y + mk.z * w0.z, mk.x * w1.x + mk.y * w1.y + mk.z * w1.z); }
        if (mk.x * opp.x + mk.y * opp.y + mk.z * opp.z > o) { ok = false; break; }
        m[k][0] = mk.x; m[k][1] = mk.y; m[k][2] = mk.z;
        m[k][3] = o + 8.0e-6f * vmax + 1.0e-30f;                     // outward: makes the bound smaller
      }
      if (ok) {
        pl.nx = nu.x; pl.ny = nu.y; pl.nz = nu.z; pl.dn = nu.x * a.x + nu.y * a.y + nu.z * a.z;
        pl.m0x = m[0][0]; pl.m0y = m[0][1]; pl.m0z = m[0][2]; pl.o0 = m[0][3];
        pl.m1x = m[1][0]; pl.m1y = m[1][1]; pl.m1z = m[1][2]; pl.o1 = m[1][3];
        pl.m2x = m[2][0]; pl.m2y = m[2][1]; pl.m2z = m[2][2]; pl.o2 = m[2][3];
      }
    }
    planes[first] = pl;
  }
}

__device__ __forceinline__ float wave_sum(float v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}
__device__ __forceinline__ float wave_min(float v) {
  for (int off = 32; off > 0; off >>= 1) v = fminf(v, __shfl_xor(v, off));
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
  return v;
}

// Oriented bound (NodeExt) of every node: one wave per pre-order slot, lanes stride over the
// triangles of the subtree (contiguous in Morton order).  Pass 1: area-weighted mean normal.
// Pass 2: extent along it and lateral radius about the AABB centre.  All roundings go outwards.
constexpr uint32_t EXT_THREAD_BELOW = 16;   // subtrees up to this many triangles: one THREAD per node

// Slab [lo, hi] along n in the (mid, half) form the walk tests with one op less: max(|t - mid| - half, 0).
// half is rounded up over both one-sided widths, so the stored slab contains [lo, hi].
__device__ __forceinline__ void set_slab(NodeExt& x, float lo, float hi) {
  const float mid = 0.5f * lo + 0.5f * hi;
  const float w = fmaxf(hi - mid, mid - lo);
  x.mid = mid;
  x.half = w + fabsf(w) * 2.4e-7f;   // >= the exact widths: each subtraction above is off by <= 1/2 ulp(w)
}

// Serial version of the same computation for small subtrees (most nodes: half of them are leaves).
__device__ __forceinline__ void node_ext_thread(const NodeRec& nr, uint32_t slot, uint32_t cnt, const uint32_t* __restrict__ slot_first,
                                                const TriRec* __restrict__ tris, NodeExt* __restrict__ ext) {
  const uint32_t first = slot_first[slot];
  float sx = 0.0f, sy = 0.0f, sz = 0.0f;
  for (uint32_t i = 0; i < cnt; ++i) {
    const TriRec& t = tris[first + i];
    const f3 n = cross3(mk3(t.abx, t.aby, t.abz), mk3(t.acx, t.acy, t.acz));
    if (fabsf(n.x) < 3.0e38f && fabsf(n.y) < 3.0e38f && fabsf(n.z) < 3.0e38f) { sx += n.x; sy += n.y; sz += n.z; }
  }
  const float len = sqrtf(sx * sx + sy * sy + sz * sz);
  float nx = 1.0f, ny = 0.0f, nz = 0.0f;
  if (len > 1.0e-30f && len < 3.0e38f) { nx = sx / len; ny = sy / len; nz = sz / len; }
  float cx = 0.5f * (nr.mnx + nr.mxx), cy = 0.5f * (nr.mny + nr.mxy), cz = 0.5f * (nr.mnz + nr.mxz);
  if (!(fabsf(cx) < 3.0e38f)) cx = 0.0f;
  if (!(fabsf(cy) < 3.0e38f)) cy = 0.0f;
  if (!(fabsf(cz) < 3.0e38f)) cz = 0.0f;
  const float inf = __builtin_inff();
  float dlo = inf, dhi = -inf, r2 = 0.0f, w2max = 0.0f;
  for (uint32_t i = 0; i < cnt; ++i) {
    const TriRec& t = tris[first + i];
    const float vx[3] = {t.ax, t.bx, t.cx}, vy[3] = {t.ay, t.by, t.cy}, vz[3] = {t.az, t.bz, t.cz};
    for (int k = 0; k < 3; ++k) {
      const float wx = vx[k] - cx, wy = vy[k] - cy, wz = vz[k] - cz;
      const float tt = nx * wx + ny * wy + nz * wz;
      const float w2 = wx * wx + wy * wy + wz * wz;
      dlo = fminf(dlo, tt);
      dhi = fmaxf(dhi, tt);
      r2 = fmaxf(r2, w2 - tt * tt);
      w2max = fmaxf(w2max, w2);
    }
  }
  const float R = sqrtf(fmaxf(r2, 0.0f) + 1.0e-6f * w2max) * 1.00001f + 1.0e-30f;
  const float e = 1.0e-5f * (fabsf(dlo) + fabsf(dhi)) + 2.0e-6f * sqrtf(w2max) + 1.0e-30f;
  NodeExt x;
  x.cx = cx; x.cy = cy; x.cz = cz; x.R = R;
  x.nx = nx; x.ny = ny; x.nz = nz; set_slab(x, dlo - e, dhi + e);
  x.skip = nr.skip * (uint32_t)sizeof(NodeExt); x.tri = nr.tri; x.pad = 0;   // BYTE offset of the skip target
  ext[slot] = x;
}

// One wave per node of more than EXT_THREAD_BELOW triangles.
__device__ __forceinline__ void node_ext_wave(const NodeRec* __restrict__ nodes, uint32_t slot, int lane, const uint32_t* __restrict__ slot_first,
                                              const TriRec* __restrict__ tris, NodeExt* __restrict__ ext) {
  const NodeRec nr = nodes[slot];
  const uint32_t cnt = (nr.skip - slot + 1u) >> 1;
  const uint32_t first = slot_first[slot];
  float cx = 0.5f * (nr.mnx + nr.mxx), cy = 0.5f * (nr.mny + nr.mxy), cz = 0.5f * (nr.mnz + nr.mxz);
  if (!(fabsf(cx) < 3.0e38f)) cx = 0.0f;
  if (!(fabsf(cy) < 3.0e38f)) cy = 0.0f;
  if (!(fabsf(cz) < 3.0e38f)) cz = 0.0f;
  if (cnt > EXT_TRIVIAL_ABOVE) {
    if (lane == 0) {  // cylinder (axis +x) around the AABB
      const float hx = fmaxf(nr.mxx - cx, cx - nr.mnx), hy = fmaxf(nr.mxy - cy, cy - nr.mny), hz = fmaxf(nr.mxz - cz, cz - nr.mnz);
      const float e = 1.0e-5f * (fabsf(hx) + fabsf(hy) + fabsf(hz)) + 1.0e-30f;
      NodeExt x;
      x.cx = cx; x.cy = cy; x.cz = cz; x.R = sqrtf(hy * hy + hz * hz) * 1.00001f + e;
      x.nx = 1.0f; x.ny = 0.0f; x.nz = 0.0f; set_slab(x, -hx - e, hx + e);
      x.skip = nr.skip * (uint32_t)sizeof(NodeExt); x.tri = nr.tri; x.pad = 0;   // BYTE offset of the skip target
      ext[slot] = x;
    }
    return;
  }
  float sx = 0.0f, sy = 0.0f, sz = 0.0f;
  for (uint32_t i = lane; i < cnt; i += 64) {
    const TriRec t = tris[first + i];
    const f3 n = cross3(mk3(t.bx - t.ax, t.by - t.ay, t.bz - t.az), mk3(t.cx - t.ax, t.cy - t.ay, t.cz - t.az));
    if (fabsf(n.x) < 3.0e38f && fabsf(n.y) < 3.0e38f && fabsf(n.z) < 3.0e38f) { sx += n.x; sy += n.y; sz += n.z; }
  }
  sx = wave_sum(sx); sy = wave_sum(sy); sz = wave_sum(sz);
  float len = sqrtf(sx * sx + sy * sy + sz * sz);
  float nx = 1.0f, ny = 0.0f, nz = 0.0f;
  if (len > 1.0e-30f && len < 3.0e38f) { nx = sx / len; ny = sy / len; nz = sz / len; }
  const float inf = __builtin_inff();
  float dlo = inf, dhi = -inf, r2 = 0.0f, w2max = 0.0f;
  for (uint32_t i = lane; i < cnt; i += 64) {
    const TriRec t = tris[first + i];
    const float vx[3] = {t.ax, t.bx, t.cx}, vy[3] = {t.ay, t.by, t.cy}, vz[3] = {t.az, t.bz, t.cz};
    for (int k = 0; k < 3; ++k) {
      const float wx = vx[k] - cx, wy = vy[k] - cy, wz = vz[k] - cz;
      const float tt = nx * wx + ny * wy + nz * wz;
      const float w2 = wx * wx + wy * wy + wz * wz;
      dlo = fminf(dlo, tt);
      dhi = fmaxf(dhi, tt);
      r2 = fmaxf(r2, w2 - tt * tt);
      w2max = fmaxf(w2max, w2);
    }
  }
  dlo = wave_min(dlo); dhi = wave_max(dhi); r2 = wave_max(r2); w2max = wave_max(w2max);
  if (lane == 0) {
    // outward rounding: l^2 = w^2 - t^2 cancels, so widen by a few ulps of w^2 before the sqrt
    const float R = sqrtf(fmaxf(r2, 0.0f) + 1.0e-6f * w2max) * 1.00001f + 1.0e-30f;
    const float e = 1.0e-5f * (fabsf(dlo) + fabsf(dhi)) + 2.0e-6f * sqrtf(w2max) + 1.0e-30f;
    NodeExt x;
    x.cx = cx; x.cy = cy; x.cz = cz; x.R = R;
    x.nx = nx; x.ny = ny; x.nz = nz; set_slab(x, dlo - e, dhi + e);
    x.skip = nr.skip * (uint32_t)sizeof(NodeExt); x.tri = nr.tri; x.pad = 0;   // BYTE offset of the skip target
    ext[slot] = x;
  }
}

// Both in one launch: a thread per small node; the few larger nodes among a block's 256 slots are queued in LDS and taken by the
// block's four waves afterwards.  (Two launches before — one thread per node, then one WAVE per node of which 94 % returned at once:
// 16 + 37 us of the build's critical path for 100 k triangles.)
__global__ __launch_bounds__(256) void k_node_ext(const NodeRec* __restrict__ nodes, const uint32_t* __restrict__ slot_first,
                                                  const TriRec* __restrict__ tris, uint32_t n_nodes, NodeExt* __restrict__ ext) {
  __shared__ uint32_t big[256];
  __shared__ uint32_t n_big;
  if (threadIdx.x == 0) n_big = 0;
  __syncthreads();
  const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
  if (slot < n_nodes) {
    const NodeRec nr = nodes[slot];
    const uint32_t cnt = (nr.skip - slot + 1u) >> 1;
    if (cnt <= EXT_THREAD_BELOW) node_ext_thread(nr, slot, cnt, slot_first, tris, ext);
    else big[atomicAdd(&n_big, 1u)] = slot;
  }
  __syncthreads();
  const uint32_t nb = n_big;
  for (uint32_t k = threadIdx.x >> 6; k < nb; k += 4u) node_ext_wave(nodes, big[k], (int)(threadIdx.x & 63u), slot_first, tris, ext);
}

// ---- treelet pass ---------------------------------------------------------------------------------------
// The LBVH's top is fine, what costs node tests is how groups of 16-512 triangles are partitioned (DESIGN.md §4,
// tools/exp_tree.py: LBVH splits down to nodes of <= 64 triangles with sweep splits inside them make the walk 5 % shorter).
// k_treelet_roots lists the maximal LBVH nodes of at most TREELET_MAX triangles; k_treelet rebuilds each with one wave:
// top-down, every segment split along the widest axis of its triangle-box centres at the position that minimises
// (sum of box extents) x (triangle count) over both sides.  The triangles of the node are reordered inside its range
// of the sorted arrays and their keys keep the node's Morton prefix followed by the path in the new treelet (prefix-free
// codes), so the radix tree over the keys (k_karras, run again) is the old tree above the node and the new one inside.
#ifndef M2S_TREELET_MAX
#define M2S_TREELET_MAX 64
#endif
constexpr int TREELET_MAX = M2S_TREELET_MAX;

__global__ __launch_bounds__(256) void k_treelet_roots(int n, const int2* __restrict__ range, const int* __restrict__ parent,
                                                        int2* __restrict__ roots, int* __restrict__ n_roots) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n - 1) return;
  const int2 r = range[i];
  const int cnt = r.y - r.x + 1;
  if (cnt > TREELET_MAX || cnt < 3) return;
  const int p = parent[i];
  bool is_root = p == INT32_MIN;
  if (!is_root) {
    const int pi = p >= 0 ? p : -p - 2;
    const int2 pr = range[pi];
    is_root = pr.y - pr.x + 1 > TREELET_MAX;
  }
  if (is_root) roots[atomicAdd(n_roots, 1)] = make_int2(r.x, cnt);
}

__global__ __launch_bounds__(TREELET_MAX) void k_treelet(const int2* __restrict__ roots, const int* __restrict__ n_roots,
                                                const Box* __restrict__ boxes, uint64_t* __restrict__ keys,
                                                uint32_t* __restrict__ order) {
  // the treelet's items by POSITION (a segment is a contiguous range of positions, so a lane loops over its own segment only:
  // the trip counts halve from level to level instead of staying at the treelet size)
  __shared__ float s_box[6][TREELET_MAX];
  __shared__ float s_cen[3][TREELET_MAX];
  __shared__ int s_key[TREELET_MAX], s_rank[TREELET_MAX], s_cost[TREELET_MAX];
  if ((int)blockIdx.x >= *n_roots) return;
  const int2 root = roots[blockIdx.x];
  const int first = root.x, m = root.y, lane = threadIdx.x;
  const bool live = lane < m;
  const uint64_t k_first = keys[first], k_last = keys[first + m - 1];
  if (k_first == k_last) return;                                  // identical centres: no room below the prefix
  const int prefix = __clzll((long long)(k_first ^ k_last));      // bits the node's keys share
  const int room = 64 - prefix;                                   // bits left for the path inside the treelet
  const uint32_t tri = live ? order[first + lane] : 0u;
  const uint64_t old_key = live ? keys[first + lane] : 0ull;
  Box b = {0, 0, 0, 0, 0, 0};
  if (live) b = boxes[tri];
  const float cen[3] = {0.5f * (b.mnx + b.mxx), 0.5f * (b.mny + b.mxy), 0.5f * (b.mnz + b.mxz)};
  const int ck[3] = {ord(cen[0]), ord(cen[1]), ord(cen[2])};      // total order, NaN included
  int pos = lane, s = 0, e = live ? m : 0;                        // my position; my segment [s, e) of positions
  uint64_t code = 0;
  int depth = 0;
  for (int level = 0; level < 64; ++level) {
    const bool open = live && e - s > 1 && depth < room;
    if (__syncthreads_or(open ? 1 : 0) == 0) break;
    if (live) {
      s_box[0][pos] = b.mnx; s_box[1][pos] = b.mny; s_box[2][pos] = b.mnz; s_box[3][pos] = b.mxx; s_box[4][pos] = b.mxy; s_box[5][pos] = b.mxz;
      s_cen[0][pos] = cen[0]; s_cen[1][pos] = cen[1]; s_cen[2][pos] = cen[2];
    }
    __syncthreads();
    // widest axis of the centres of my segment
    int axis = 2;
    if (open) {
      float mn0 = __builtin_inff(), mn1 = mn0, mn2 = mn0, mx0 = -mn0, mx1 = -mn0, mx2 = -mn0;
      for (int j = s; j < e; ++j) {
        const float ox = s_cen[0][j], oy = s_cen[1][j], oz = s_cen[2][j];
        mn0 = fminf(mn0, ox); mx0 = fmaxf(mx0, ox); mn1 = fminf(mn1, oy); mx1 = fmaxf(mx1, oy); mn2 = fminf(mn2, oz); mx2 = fmaxf(mx2, oz);
      }
      const float e0 = mx0 - mn0, e1 = mx1 - mn1, e2 = mx2 - mn2;
      axis = (e0 >= e1 && e0 >= e2) ? 0 : (e1 >= e2 ? 1 : 2);   // NaN extents: comparisons false -> axis 2; any axis is valid
    }
    const int key = axis == 0 ? ck[0] : (axis == 1 ? ck[1] : ck[2]);
    if (live) s_key[pos] = key;
    __syncthreads();
    // my rank inside my segment along that axis (strict order by (key, position))
    int rank = pos;
    if (open) {
      rank = s;
      for (int j = s; j < e; ++j) {
        const int oj = s_key[j];
        rank += (oj < key || (oj == key && j < pos)) ? 1 : 0;
      }
    }
    if (live) s_rank[pos] = rank;
    __syncthreads();
    // the split after me: boxes of the triangles up to my rank and of the rest
    int cost_key = INT32_MAX;
    if (open) {
      float l0 = __builtin_inff(), l1 = l0, l2 = l0, l3 = -l0, l4 = -l0, l5 = -l0;
      float r0 = l0, r1 = l0, r2 = l0, r3 = -l0, r4 = -l0, r5 = -l0;
      for (int j = s; j < e; ++j) {
        const bool left = s_rank[j] <= rank;
        const float a0 = s_box[0][j], a1 = s_box[1][j], a2 = s_box[2][j], a3 = s_box[3][j], a4 = s_box[4][j], a5 = s_box[5][j];
        if (left) { l0 = fminf(l0, a0); l1 = fminf(l1, a1); l2 = fminf(l2, a2); l3 = fmaxf(l3, a3); l4 = fmaxf(l4, a4); l5 = fmaxf(l5, a5); }
        else { r0 = fminf(r0, a0); r1 = fminf(r1, a1); r2 = fminf(r2, a2); r3 = fmaxf(r3, a3); r4 = fmaxf(r4, a4); r5 = fmaxf(r5, a5); }
      }
      const int n_left = rank - s + 1, n_right = e - rank - 1;
      const float cost = ((l3 - l0) + (l4 - l1) + (l5 - l2)) * (float)n_left + ((r3 - r0) + (r4 - r1) + (r5 - r2)) * (float)n_right;
      if (n_right > 0) cost_key = ord(cost);                      // the last rank is not a split
    }
    if (live) s_cost[pos] = cost_key;
    __syncthreads();
    if (open) {
      // the best split of my segment: smallest (cost, rank)
      int best_cost = INT32_MAX, best_rank = s + (e - s) / 2 - 1;   // fallback (cannot be needed: a segment of 2+ has a valid split)
      for (int j = s; j < e; ++j) {
        const int cj = s_cost[j], rj = s_rank[j];
        if (cj < best_cost || (cj == best_cost && cj != INT32_MAX && rj < best_rank)) { best_cost = cj; best_rank = rj; }
      }
      const bool right = rank > best_rank;
      pos = rank;
      code = (code << 1) | (right ? 1ull : 0ull);
      ++depth;
      if (right) s = best_rank + 1; else e = best_rank + 1;
    }
  }
  if (live) {
    // Morton prefix of the node, then the path inside the treelet, left aligned (depth <= room)
    const uint64_t mask = ~0ull << room;
    const uint64_t path = depth ? code << (room - depth) : 0ull;
    keys[first + pos] = (old_key & mask) | path;
    order[first + pos] = tri;
  }
}

__global__ void k_init_scene(int* scene, int* parent, int n_nodes) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 3) scene[i] = INT32_MAX;
  else if (i < 6) scene[i] = INT32_MIN;
  if (i < n_nodes) parent[i] = INT32_MIN;
}

inline unsigned cdiv(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }

}  // namespace

size_t bvh_workspace_bytes(size_t n_tris) {
  const size_t n = n_tris ? n_tris : 1;
  size_t sort_tmp = 0;
  (void)rocprim::radix_sort_pairs(nullptr, sort_tmp, (uint64_t*)nullptr, (uint64_t*)nullptr, (uint32_t*)nullptr,
                            (uint32_t*)nullptr, n, 0, 64, (hipStream_t)0);
  size_t b = 0;
  b += n * sizeof(TriRec) * 2 + n * 16 + 256 + n * 16 + 256 + n * 4 + 256 + n * sizeof(TriPlanes) + 256 + n * sizeof(Box) * 3 + n * (8 + 4) * 2 + sort_tmp;
  b += n * (sizeof(int2) * 2) + 2 * n * sizeof(int) + 2 * n * sizeof(NodeRec) + 2 * n * (sizeof(NodeExt) + 4);
  return b + 64 * 256 + 4096 + 24 * ((n + 255) / 256) + 256;
}

int build_device_mesh(Arena& ws, hipStream_t st, const float* d_verts, size_t n_verts, const void* d_indices,
                      size_t n_indices, int index_bytes, int topology, size_t n_tris, int* d_err, DeviceMesh* out,
                      const std::function<int(const float4*, const TriRec*, int)>* after_setup) {
  (void)n_indices;
  out->cen_raw = nullptr;
  out->slot_of = nullptr;
  out->tris = nullptr;
  out->cen = nullptr;
  out->planes = nullptr;
  out->nodes = nullptr;
  out->ext = nullptr;
  out->stats = nullptr;
  out->scene = nullptr;
  if (n_tris > (1u << 25)) {   // the walks address 96-byte records through 32-bit byte offsets
    set_error("mesh has %zu triangles; this build handles up to 33 554 432", n_tris);
    return M2S_ERR_BAD_ARG;
  }
  out->n_tris = (uint32_t)n_tris;
  out->n_nodes = n_tris ? (uint32_t)(2 * n_tris - 1) : 0;
  if (n_tris == 0) return 0;
  const int n = (int)n_tris;

  TriRec* raw = ws.take<TriRec>(n_tris);
  TriRec* tris = ws.take<TriRec>(n_tris);
  float4* cen = ws.take<float4>(n_tris);
  float4* cen_raw = ws.take<float4>(n_tris);
  uint32_t* slot_of = ws.take<uint32_t>(n_tris);
  TriPlanes* planes = ws.take<TriPlanes>(n_tris);
  Box* boxes = ws.take<Box>(n_tris);
  Box* seg = ws.take<Box>(2 * n_tris + 64);
  uint64_t* keys = ws.take<uint64_t>(n_tris);
  uint64_t* keys2 = ws.take<uint64_t>(n_tris);
  uint32_t* vals = ws.take<uint32_t>(n_tris);
  uint32_t* order = ws.take<uint32_t>(n_tris);
  int2* range = ws.take<int2>(n_tris);
  int2* child = ws.take<int2>(n_tris);
  int* parent = ws.take<int>(2 * n_tris);
  NodeRec* nodes = ws.take<NodeRec>(2 * n_tris);
  NodeExt* ext = ws.take<NodeExt>(2 * n_tris);
  uint32_t* slot_first = ws.take<uint32_t>(2 * n_tris);
  int* scene = ws.take<int>(8 + 6 * ((n_tris + 255) / 256));
  size_t sort_tmp = 0;
  (void)rocprim::radix_sort_pairs(nullptr, sort_tmp, keys, keys2, vals, order, n_tris, 0, 64, st);
  void* tmp = ws.take<char>(sort_tmp ? sort_tmp : 1);
  if (!raw || !tris || !boxes || !seg || !keys || !keys2 || !vals || !order || !range || !child || !parent || !nodes ||
      !scene || !tmp || !ext || !cen_raw || !slot_of || !slot_first || !cen || !planes) {
    set_error("internal: BVH workspace too small");
    return M2S_ERR_HIP_INTERNAL;
  }

  const unsigned B = 256;
  static const uint32_t leaf_max = getenv("M2S_LEAF_MAX") ? std::max(1u, (uint32_t)atoi(getenv("M2S_LEAF_MAX"))) : 2u;
  hipLaunchKernelGGL(k_init_scene, dim3(cdiv(2 * n_tris, B)), dim3(B), 0, st, scene, parent, 2 * n - 1);
  hipLaunchKernelGGL(k_tri_setup, dim3(cdiv(n_tris, B)), dim3(B), 0, st, d_verts, (uint32_t)n_verts, d_indices,
                     index_bytes, topology, (uint32_t)n_tris, raw, boxes, cen_raw, scene, d_err);
  hipLaunchKernelGGL(k_scene_reduce, dim3(1), dim3(256), 0, st, scene, (uint32_t)cdiv(n_tris, B));
  out->cen_raw = cen_raw;
  out->slot_of = slot_of;
  if (after_setup) {   // the caller's seed passes only need the centroids: they run beside the sort and the hierarchy
    const int rc = (*after_setup)(cen_raw, raw, 0);   // phase 0: mark this point of the stream (an event), launch nothing yet
    if (rc) return rc;
  }
  hipLaunchKernelGGL(k_morton, dim3(cdiv(n_tris, B)), dim3(B), 0, st, boxes, (uint32_t)n_tris, scene, keys, vals);
  if (const char* kf = getenv("M2S_KEYS_FILE")) {
    // Experiment knob (tools/exp_tree.py): one 64-bit key per triangle from a file instead of the Morton keys.  The
    // radix tree over prefix-free path codes IS the tree that produced them, so any binary tree of depth < 64 built
    // elsewhere can be walked by the unchanged kernels.
    std::vector<uint64_t> hk(n_tris);
    FILE* f = fopen(kf, "rb");
    const bool ok = f && fread(hk.data(), 8, n_tris, f) == n_tris;
    if (f) fclose(f);
    if (!ok) { set_error("M2S_KEYS_FILE: cannot read one key per triangle"); return M2S_ERR_BAD_ARG; }
    M2S_HIP_CHECK(hipMemcpyAsync(keys, hk.data(), 8 * n_tris, hipMemcpyHostToDevice, st));
    M2S_HIP_CHECK(hipStreamSynchronize(st));
  }
  M2S_HIP_CHECK(rocprim::radix_sort_pairs(tmp, sort_tmp, keys, keys2, vals, order, n_tris, 0, 64, st));
  if (n > 1) hipLaunchKernelGGL(k_karras, dim3(cdiv(n_tris - 1, B)), dim3(B), 0, st, keys2, n, range, child, parent);
  if (after_setup) {
    // phase 1: `st` now holds ~100 us of work (keys, sort, hierarchy) — the time the host needs to enqueue the side
    // work (the seed passes, behind the phase-0 mark).  Launching it at phase 0 left `st` idle for those ~100 us (a
    // launch costs the host ~8 us, and the thin slab of a multi-GPU rank has nothing to hide that behind); launching it
    // after the whole build serialised the passes of a 512^3 lattice (0.5 ms) behind the build instead of beside it.
    const int rc = (*after_setup)(cen_raw, raw, 1);
    if (rc) return rc;
  }
  static const bool treelets = !(getenv("M2S_TREELETS") && atoi(getenv("M2S_TREELETS")) == 0);
  if (n > 2 && treelets && !getenv("M2S_KEYS_FILE")) {
    // treelet pass: the nodes of at most TREELET_MAX triangles are rebuilt by sweep splits, then the hierarchy is derived again
    int2* roots = reinterpret_cast<int2*>(child);   // child[] is not used by the kernels below and is rewritten by the second k_karras
    hipLaunchKernelGGL(k_treelet_roots, dim3(cdiv(n_tris - 1, B)), dim3(B), 0, st, n, range, parent, roots, scene + 7);
    hipLaunchKernelGGL(k_treelet, dim3((unsigned)((n_tris + 2) / 3)), dim3(TREELET_MAX), 0, st, roots, scene + 7, boxes, keys2, order);
    hipLaunchKernelGGL(k_karras, dim3(cdiv(n_tris - 1, B)), dim3(B), 0, st, keys2, n, range, child, parent);
  }

  SegLevels lv;
  lv.levels = 0;
  {
    uint32_t off = 0, cnt = (uint32_t)n_tris;
    while (true) {
      lv.off[lv.levels] = off;
      lv.cnt[lv.levels] = cnt;
      ++lv.levels;
      if (cnt == 1) break;
      off += cnt;
      cnt = (cnt + 1) / 2;
    }
  }
  if (lv.levels == 1) hipLaunchKernelGGL(k_seg_level0, dim3(cdiv(n_tris, B)), dim3(B), 0, st, boxes, order, n, seg);   // a single triangle
  for (int l = 0; l + 1 < lv.levels; l += 3) {
    const uint32_t n1 = lv.cnt[l + 1];
    const uint32_t n2 = l + 2 < lv.levels ? lv.cnt[l + 2] : 0, n3 = l + 3 < lv.levels ? lv.cnt[l + 3] : 0;
    const uint32_t o2 = l + 2 < lv.levels ? lv.off[l + 2] : 0, o3 = l + 3 < lv.levels ? lv.off[l + 3] : 0;
    const uint32_t threads = (n1 + 3) / 4;
    if (l == 0)   // the leaf level is gathered by the launch that consumes it
      hipLaunchKernelGGL(k_seg_level3<true>, dim3(cdiv(threads, B)), dim3(B), 0, st, seg, lv.off[l], lv.cnt[l], lv.off[l + 1], n1, o2,
                         n2, o3, n3, (const Box*)boxes, (const uint32_t*)order);
    else
      hipLaunchKernelGGL(k_seg_level3<false>, dim3(cdiv(threads, B)), dim3(B), 0, st, seg, lv.off[l], lv.cnt[l], lv.off[l + 1], n1, o2,
                         n2, o3, n3, (const Box*)nullptr, (const uint32_t*)nullptr);
  }
  hipLaunchKernelGGL(k_emit, dim3(cdiv(2 * n_tris - 1, B)), dim3(B), 0, st, n, range, parent, seg, lv, order, raw,
                     nodes, tris, slot_first, cen, planes, leaf_max, slot_of);
  hipLaunchKernelGGL(k_node_ext, dim3(cdiv(2 * n_tris - 1, B)), dim3(B), 0, st, nodes, slot_first, tris,
                     (uint32_t)(2 * n_tris - 1), ext);
  M2S_HIP_CHECK(hipGetLastError());
  out->tris = tris;
  out->cen = cen;
  out->planes = planes;
  out->nodes = nodes;
  out->ext = ext;
  out->scene = scene;
  return 0;
}

}  // namespace m2s
