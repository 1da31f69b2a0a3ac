// bvh.hip — topology flatten + triangle records + LBVH in stackless pre-order layout (gfx950).
//
// Replaces, as behaviour, Topology::get_triangles (lib.rs:175-193) and the acceleration
// structures the reference builds inside every call (bvh 0.10 Bvh::build_par at
// generate/grid.rs:95-111, generic/bvh.rs:74; rstar bulk_load at generic/rtree.rs:111): they
// only select candidates, the kernels in distance.hip take the exact minimum.
//
// Pipeline (all on the call's stream, no host round trip):
//   k_tri_setup     : indices -> (a,b,c), degeneracy class, padded box (geo.rs:4-22), scene bounds
//   k_sort_tiles / k_sort_rank / k_sort_buckets (lbvh_sort.hip.h) : 63-bit Morton keys of the box centres and their sample sort;
//                     above 229 376 triangles k_morton_keys + rocPRIM's radix sort — neither is on the parity path
//   k_roots_from_keys, k_treelet_lanes : sweep-split treelets of <= 64 triangles (their keys rewritten as path codes)
//   k_hierarchy     : Karras 2012 ranges over the sorted keys, the segment tree of leaf boxes (fence-free refit) and the record sets
//                     that place a node in pre-order, side by side in one launch
//   k_emit          : node boxes by range query, pre-order index = 2*first + #left-turns, skip links
//   k_node_ext      : oriented bound (disc-shaped slab) of every node
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <cstdio>
#include <functional>
#include <vector>


#include "../../include/m2s.h"
#include "common.h"
#include "tuning.h"
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
                                                   int* __restrict__ scene /*final values, then 6 per block from part_base on*/, int part_base,
                                                   int* __restrict__ err, uint32_t* __restrict__ aux, uint32_t aux_words) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (aux && blockIdx.x == 0 && threadIdx.x < aux_words) aux[threadIdx.x] = 0u;   // the build's counters
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
  // block reduce (wave shuffles, then LDS); one partial per block, folded by k_morton_keys:
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
    scene[part_base + blockIdx.x * 6 + k] = v;   // partials live behind the final values
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

// Karras 2012.  Keys may repeat; ties are broken by position, which keeps the tree well formed.
__device__ __forceinline__ int delta(const uint64_t* __restrict__ keys, int n, int i, int j) {
  if (j < 0 || j >= n) return -1;
  const uint64_t a = keys[i], b = keys[j];
  if (a == b) return 64 + __clz((uint32_t)i ^ (uint32_t)j);
  return __clzll((long long)(a ^ b));
}

// Node ids: internal i in [0, n-2], leaf k -> (n-1) + k.  Only the RANGE of a node is derived (its direction and the far end of its
// keys): k_emit places a node by its range alone (see Recs below), so nobody needs children or parents.
__device__ __forceinline__ void karras_range(const uint64_t* __restrict__ keys, int n, int i, int2* __restrict__ range) {
  const int d = (delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1)) >= 0 ? 1 : -1;
  const int dmin = delta(keys, n, i, i - d);
  int lmax = 2;
  while (delta(keys, n, i, i + lmax * d) > dmin) lmax <<= 1;
  int l = 0;
  for (int t = lmax >> 1; t >= 1; t >>= 1)
    if (delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
  const int j = i + l * d;
  range[i] = make_int2(min(i, j), max(i, j));
}

// ---- where a node sits in pre-order, without walking to the root --------------------------------------------------------------
// The pre-order slot of a node with the keys [first, last] is 2 * first + (left turns on the way down from the root).  Going UP from
// the node, its nearest ancestor that has it in the LEFT subtree is the node that splits between `last` and `last + 1`; that
// ancestor's own last key is the next position to the right whose adjacent prefix d[] is SMALLER than d[last], and so on: the left
// turns are the prefix-minimum RECORDS of d[last], d[last + 1], ..., d[n - 2] — a function of `last` alone.  d[] takes values in
// [0, 96), so the records of an interval are a 96-bit set, and two adjacent intervals combine as
//     records(I J) = records(I)  |  (records(J) & bits below the smallest of I)
// k_hierarchy scans these sets from the right inside every block of 512 positions and over the blocks; k_emit reads two of them.
// (Round 4 counted the left turns by chasing parent links: ~25 dependent loads per node.)
struct Recs { uint64_t lo; uint32_t hi; };
__device__ __forceinline__ Recs recs_of(int d) {
  Recs r = {0ull, 0u};
  if (d >= 64) r.hi = 1u << (d - 64);
  else if (d >= 0) r.lo = 1ull << d;
  return r;
}
__device__ __forceinline__ Recs recs_join(Recs a, Recs b) {   // a: the left interval
  uint64_t mlo = ~0ull;
  uint32_t mhi = ~0u;
  if (a.lo) { mlo = (a.lo & (0ull - a.lo)) - 1ull; mhi = 0u; }
  else if (a.hi) { mhi = (a.hi & (0u - a.hi)) - 1u; }
  return {a.lo | (b.lo & mlo), a.hi | (b.hi & mhi)};
}
__device__ __forceinline__ uint4 recs_pack(Recs r) { return make_uint4((uint32_t)r.lo, (uint32_t)(r.lo >> 32), r.hi, 0u); }
__device__ __forceinline__ Recs recs_unpack(uint4 v) { return {(uint64_t)v.x | ((uint64_t)v.y << 32), v.z}; }
constexpr uint32_t RECS_BLOCK = 512;   // positions per block of k_hierarchy's segment-tree part

__device__ __forceinline__ Box box_union(Box a, Box b) {
  return {fminf(a.mnx, b.mnx), fminf(a.mny, b.mny), fminf(a.mnz, b.mnz),
          fmaxf(a.mxx, b.mxx), fmaxf(a.mxy, b.mxy), fmaxf(a.mxz, b.mxz)};
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

// Slab [lo, hi] along n in the (mid, half) form the walk tests with one op less: max(|t - mid| - half, 0).
// half is rounded up over both one-sided widths, so the stored slab contains [lo, hi].
__device__ __forceinline__ void set_slab(NodeExt& x, float lo, float hi) {
  const float mid = 0.5f * lo + 0.5f * hi;
  const float w = fmaxf(hi - mid, mid - lo);
  x.mid = mid;
  x.half = w + fabsf(w) * 2.4e-7f;   // >= the exact widths: each subtraction above is off by <= 1/2 ulp(w)
}

// One thread per node (internal 0..n-2, leaves n-1..2n-2): pre-order slot, box, skip link,
// and for leaves the sorted triangle record.
__global__ __launch_bounds__(256) void k_emit(int n, const int2* __restrict__ range, const uint4* __restrict__ recs_local, const uint4* __restrict__ recs_carry,
                                              const Box* __restrict__ seg, SegLevels lv,
                                              const uint32_t* __restrict__ order, const TriRec* __restrict__ raw,
                                              NodeRec* __restrict__ nodes, TriRec* __restrict__ tris,
                                              uint32_t* __restrict__ slot_first, float4* __restrict__ cen,
                                              TriPlanes* __restrict__ planes, uint32_t leaf_max, uint32_t* __restrict__ slot_of,
                                              float4* __restrict__ corners, float4* __restrict__ nrm, int* __restrict__ err) {
  const int id = blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= 2 * n - 1) return;
  const bool leaf = id >= n - 1;
  int first, last;
  if (leaf) { first = last = id - (n - 1); }
  else { int2 r = range[id]; first = r.x; last = r.y; }
  // number of left turns on the root -> node path: the prefix-minimum records of d[last ..] (Recs above)
  int lefts = 0;
  if (last >= 0 && last < n) {
    const Recs r = recs_join(recs_unpack(recs_local[last]), recs_unpack(recs_carry[(uint32_t)last / RECS_BLOCK]));
    lefts = __popcll(r.lo) + __popc(r.hi);
  }
  const uint32_t slot = 2u * (uint32_t)first + (uint32_t)lefts;
  if (slot >= 2u * (uint32_t)n - 1u || last < first || last >= n) { atomicOr(err, ERRF_BUILD_TIMEOUT); return; }   // never with sorted keys
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
    // the vertices alone for the ray walks of the generic Raycast sign: 36 of a record's 96 bytes are all they read, and 4.8 MB of
    // them stay in an XCD's L2 where 9.6 MB of records do not
    corners[3 * (size_t)first] = make_float4(r.ax, r.ay, r.az, r.bx);
    corners[3 * (size_t)first + 1] = make_float4(r.by, r.bz, r.cx, r.cy);
    corners[3 * (size_t)first + 2] = make_float4(r.cz, 0.0f, 0.0f, 0.0f);
    nrm[first] = make_float4(r.nrx, r.nry, r.nrz, 0.0f);   // k_node_ext's first pass
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

// Serial version of the same computation for small subtrees (most nodes: half of them are leaves).
// Only the VERTICES of a record are read (its first 48 bytes, three 16-byte loads): the edges b - a, c - a are the same f32 subtractions
// k_tri_setup stored.
struct TriVerts { float4 a, b, c; };   // (a, cls) (b, index) (c, pad)
__device__ __forceinline__ TriVerts load_verts(const TriRec* __restrict__ t) {
  const float4* q = reinterpret_cast<const float4*>(t);
  return {q[0], q[1], q[2]};
}
__device__ __forceinline__ f3 verts_normal(const TriVerts& v) {
  return cross3(mk3(v.b.x - v.a.x, v.b.y - v.a.y, v.b.z - v.a.z), mk3(v.c.x - v.a.x, v.c.y - v.a.y, v.c.z - v.a.z));
}
// Sum of the raw normals ab x ac of the triangles first + lane, first + lane + STRIDE, ... in that order.  The normals come from `nrm`
// (k_emit: the value k_tri_setup stored in the record, 16 bytes per triangle in sorted order): one coalesced load per triangle where the
// record's vertices are three loads 96 bytes apart per lane, so BATCH triangles are in flight at a time.
template <uint32_t STRIDE, uint32_t BATCH>
__device__ __forceinline__ void sum_normals(const float4* __restrict__ nrm, uint32_t first, uint32_t cnt, uint32_t lane, float& sx, float& sy, float& sz) {
  for (uint32_t i0 = lane; i0 < cnt; i0 += BATCH * STRIDE) {
    float4 n[BATCH];
#pragma unroll
    for (uint32_t u = 0; u < BATCH; ++u) n[u] = nrm[first + min(i0 + u * STRIDE, cnt - 1u)];
#pragma unroll
    for (uint32_t u = 0; u < BATCH; ++u)
      if (i0 + u * STRIDE < cnt && fabsf(n[u].x) < 3.0e38f && fabsf(n[u].y) < 3.0e38f && fabsf(n[u].z) < 3.0e38f) { sx += n[u].x; sy += n[u].y; sz += n[u].z; }
  }
}
// Extent of the same triangles along n about c, lateral radius, largest |v - c|^2, from the compact vertex array (48 bytes per
// triangle): BATCH triangles in flight (a repeated last triangle changes no minimum or maximum).
template <uint32_t STRIDE, uint32_t BATCH>
__device__ __forceinline__ void extent_along(const float4* __restrict__ corners, uint32_t first, uint32_t cnt, uint32_t lane, float nx, float ny, float nz,
                                             float cx, float cy, float cz, float& dlo, float& dhi, float& r2, float& w2max) {
  for (uint32_t i0 = lane; i0 < cnt; i0 += BATCH * STRIDE) {
    float4 q0[BATCH], q1[BATCH], q2[BATCH];
#pragma unroll
    for (uint32_t u = 0; u < BATCH; ++u) {
      const float4* q = corners + 3 * (size_t)(first + min(i0 + u * STRIDE, cnt - 1u));
      q0[u] = q[0]; q1[u] = q[1]; q2[u] = q[2];
    }
#pragma unroll
    for (uint32_t u = 0; u < BATCH; ++u) {
      const float vx[3] = {q0[u].x, q0[u].w, q1[u].z}, vy[3] = {q0[u].y, q1[u].x, q1[u].w}, vz[3] = {q0[u].z, q1[u].y, q2[u].x};
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
  }
}

__device__ __forceinline__ void node_ext_thread(const NodeRec& nr, uint32_t slot, uint32_t cnt, const uint32_t* __restrict__ slot_first,
                                                const float4* __restrict__ nrm, const float4* __restrict__ corners, NodeExt* __restrict__ ext) {
  const uint32_t first = slot_first[slot];
  // The sums run in the order i = 0, 1, 2, ... (a triangle past the count is skipped, not added as zero); all the node's normals are
  // fetched before the first is used: the records were written by the kernel before (L2 misses), and a loop that loads, waits and adds
  // per triangle is a chain of up to 2 x 16 memory round trips.  The second loop finds the records' lines on their way.
  float sx = 0.0f, sy = 0.0f, sz = 0.0f;
  sum_normals<1, 8>(nrm, first, cnt, 0u, sx, sy, sz);
  const float len = sqrtf(sx * sx + sy * sy + sz * sz);
  float nx = 1.0f, ny = 0.0f, nz = 0.0f;
  if (len > 1.0e-30f && len < 3.0e38f) { nx = sx / len; ny = sy / len; nz = sz / len; }
  float cx = 0.5f * (nr.mnx + nr.mxx), cy = 0.5f * (nr.mny + nr.mxy), cz = 0.5f * (nr.mnz + nr.mxz);
  if (!(fabsf(cx) < 3.0e38f)) cx = 0.0f;
  if (!(fabsf(cy) < 3.0e38f)) cy = 0.0f;
  if (!(fabsf(cz) < 3.0e38f)) cz = 0.0f;
  const float inf = __builtin_inff();
  float dlo = inf, dhi = -inf, r2 = 0.0f, w2max = 0.0f;
  extent_along<1, 4>(corners, first, cnt, 0u, nx, ny, nz, cx, cy, cz, dlo, dhi, r2, w2max);
  const float R = sqrtf(fmaxf(r2, 0.0f) + 1.0e-6f * w2max) * 1.00001f + 1.0e-30f;
  const float e = 1.0e-5f * (fabsf(dlo) + fabsf(dhi)) + 2.0e-6f * sqrtf(w2max) + 1.0e-30f;
  NodeExt x;
  x.cx = cx; x.cy = cy; x.cz = cz; x.R = R;
  x.nx = nx; x.ny = ny; x.nz = nz; set_slab(x, dlo - e, dhi + e);
  x.skip = nr.skip * (uint32_t)sizeof(NodeExt); x.tri = nr.tri; x.pad = 0;   // BYTE offset of the skip target
  ext[slot] = x;
}

// One wave per node of more than EXT_THREAD_BELOW triangles.  A lane takes the triangles lane, lane + 64, ... in that order (the sums are
// pinned by the golden trees).
__device__ __forceinline__ void node_ext_wave(const NodeRec& nr, uint32_t slot, uint32_t first, uint32_t cnt, int lane,
                                              const float4* __restrict__ nrm, const float4* __restrict__ corners, NodeExt* __restrict__ ext) {
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
  sum_normals<64, 8>(nrm, first, cnt, (uint32_t)lane, sx, sy, sz);
  sx = wave_sum(sx); sy = wave_sum(sy); sz = wave_sum(sz);
  float len = sqrtf(sx * sx + sy * sy + sz * sz);
  float nx = 1.0f, ny = 0.0f, nz = 0.0f;
  if (len > 1.0e-30f && len < 3.0e38f) { nx = sx / len; ny = sy / len; nz = sz / len; }
  const float inf = __builtin_inff();
  float dlo = inf, dhi = -inf, r2 = 0.0f, w2max = 0.0f;
  extent_along<64, 4>(corners, first, cnt, (uint32_t)lane, nx, ny, nz, cx, cy, cz, dlo, dhi, r2, w2max);
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

// Both in one launch: a thread per small node; the few larger nodes among a block's 256 slots are queued in LDS (with their records:
// the queuing thread holds them) and taken by the block's four waves afterwards, FOUR nodes at a time per wave — the nodes of up to
// 64 triangles among them (one triangle per lane: 12 of a typical workgroup's 15) fetch all their normals, then all their vertices,
// in two round trips where node after node took two each.  (Two launches before — one thread per node, then one WAVE per node of which
// 94 % returned at once: 16 + 37 us of the build's critical path for 100 k triangles.  Measured in round 5 and not kept: the large
// nodes found by internal node id in extra workgroups, spread evenly instead of a root-to-leaf spine per workgroup — 30.3 against
// 29.7 us at 100 k triangles, 206 against 189 at 1 M (the locality of a workgroup's own slots is worth more); eight waves per workgroup
// with the two parts side by side — 35.7 us.)
__global__ __launch_bounds__(256) void k_node_ext(const NodeRec* __restrict__ nodes, const uint32_t* __restrict__ slot_first,
                                                  const float4* __restrict__ nrm, const float4* __restrict__ corners, uint32_t n_nodes, NodeExt* __restrict__ ext) {
  __shared__ NodeRec big_nr[256];
  __shared__ uint32_t big_slot[256], big_first[256];
  __shared__ uint32_t n_big;
  if (threadIdx.x == 0) n_big = 0;
  __syncthreads();
  const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
  if (slot < n_nodes) {
    const NodeRec nr = nodes[slot];
    const uint32_t cnt = (nr.skip - slot + 1u) >> 1;
    if (cnt <= EXT_THREAD_BELOW) node_ext_thread(nr, slot, cnt, slot_first, nrm, corners, ext);
    else {
      const uint32_t at = atomicAdd(&n_big, 1u);
      big_nr[at] = nr;
      big_slot[at] = slot;
      big_first[at] = slot_first[slot];
    }
  }
  __syncthreads();
  const uint32_t nb = n_big, wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  const float inf = __builtin_inff();
  for (uint32_t k0 = wave; k0 < nb; k0 += 16u) {
    NodeRec nr[4];
    uint32_t sl[4], first[4], cnt[4];
    bool have[4], one[4];                                      // a node of this group; one triangle per lane
#pragma unroll
    for (uint32_t u = 0; u < 4; ++u) {
      const uint32_t k = k0 + 4u * u;
      have[u] = k < nb;
      nr[u] = big_nr[have[u] ? k : k0];
      sl[u] = big_slot[have[u] ? k : k0];
      first[u] = big_first[have[u] ? k : k0];
      cnt[u] = (nr[u].skip - sl[u] + 1u) >> 1;
      one[u] = have[u] && cnt[u] <= 64u;
    }
    float4 nv[4];
#pragma unroll
    for (uint32_t u = 0; u < 4; ++u) nv[u] = (one[u] && lane < cnt[u]) ? nrm[first[u] + lane] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    float4 q0[4], q1[4], q2[4];
#pragma unroll
    for (uint32_t u = 0; u < 4; ++u) {
      const float4* q = corners + 3 * (size_t)(first[u] + min(lane, cnt[u] - 1u));
      if (one[u]) { q0[u] = q[0]; q1[u] = q[1]; q2[u] = q[2]; }
    }
#pragma unroll
    for (uint32_t u = 0; u < 4; ++u) {
      if (!one[u]) continue;
      // the same sums, in the same order, as node_ext_wave takes for a node of at most 64 triangles
      float sx = 0.0f, sy = 0.0f, sz = 0.0f;
      if (lane < cnt[u] && fabsf(nv[u].x) < 3.0e38f && fabsf(nv[u].y) < 3.0e38f && fabsf(nv[u].z) < 3.0e38f) { sx += nv[u].x; sy += nv[u].y; sz += nv[u].z; }
      sx = wave_sum(sx); sy = wave_sum(sy); sz = wave_sum(sz);
      const float len = sqrtf(sx * sx + sy * sy + sz * sz);
      float nx = 1.0f, ny = 0.0f, nz = 0.0f;
      if (len > 1.0e-30f && len < 3.0e38f) { nx = sx / len; ny = sy / len; nz = sz / len; }
      float cx = 0.5f * (nr[u].mnx + nr[u].mxx), cy = 0.5f * (nr[u].mny + nr[u].mxy), cz = 0.5f * (nr[u].mnz + nr[u].mxz);
      if (!(fabsf(cx) < 3.0e38f)) cx = 0.0f;
      if (!(fabsf(cy) < 3.0e38f)) cy = 0.0f;
      if (!(fabsf(cz) < 3.0e38f)) cz = 0.0f;
      float dlo = inf, dhi = -inf, r2 = 0.0f, w2max = 0.0f;
      if (lane < cnt[u]) {
        const float vx[3] = {q0[u].x, q0[u].w, q1[u].z}, vy[3] = {q0[u].y, q1[u].x, q1[u].w}, vz[3] = {q0[u].z, q1[u].y, q2[u].x};
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
      if (lane == 0u) {
        const float R = sqrtf(fmaxf(r2, 0.0f) + 1.0e-6f * w2max) * 1.00001f + 1.0e-30f;
        const float e = 1.0e-5f * (fabsf(dlo) + fabsf(dhi)) + 2.0e-6f * sqrtf(w2max) + 1.0e-30f;
        NodeExt x;
        x.cx = cx; x.cy = cy; x.cz = cz; x.R = R;
        x.nx = nx; x.ny = ny; x.nz = nz; set_slab(x, dlo - e, dhi + e);
        x.skip = nr[u].skip * (uint32_t)sizeof(NodeExt); x.tri = nr[u].tri; x.pad = 0;   // BYTE offset of the skip target
        ext[sl[u]] = x;
      }
    }
#pragma unroll
    for (uint32_t u = 0; u < 4; ++u)
      if (have[u] && !one[u]) node_ext_wave(nr[u], sl[u], first[u], cnt[u], (int)lane, nrm, corners, ext);
  }
}

// ---- treelet pass ---------------------------------------------------------------------------------------
// The LBVH's top is fine, what costs node tests is how groups of 16-512 triangles are partitioned (LBVH splits
// down to nodes of <= 64 triangles with sweep splits inside them make the walk 5 % shorter).  k_roots_from_keys lists the maximal
// LBVH nodes of at most TREELET_MAX triangles; k_treelet_lanes rebuilds each with one wave: top-down, every segment split along the
// widest axis of its triangle-box centres at the position that minimises (sum of box extents) x (triangle count) over both sides.
// The triangles of the node are reordered inside its range of the sorted arrays and their keys keep the node's Morton prefix
// followed by the path in the new treelet (prefix-free codes), so the radix tree over the keys (k_hierarchy's Karras ranges) is the LBVH above the
// node and the new treelet inside.
#ifndef M2S_TREELET_SORT_ABOVE
#define M2S_TREELET_SORT_ABOVE 12
#endif
constexpr int TREELET_MAX = 64;   // one wave holds a treelet's items in its lanes

// The treelet roots straight from the sorted keys, without the hierarchy.  The radix tree over the keys (ties broken by position, as karras_range does: the common prefix of two EQUAL keys j < j' is
// 64 + clz(j ^ j')) has the property that the common prefix of any two positions is the minimum of the ADJACENT prefixes between
// them, so the nodes that contain position p are found by growing [l, r] from [p, p]: the next node up shares c = max(d[l - 1], d[r])
// bits and reaches as far as the adjacent prefixes stay >= c.  The largest such node of at most TREELET_MAX triangles is p's
// treelet; it is a root of the old list iff it has at least 3 triangles, and its first position reports it.  A block keeps the
// adjacent prefixes of its 256 positions plus 66 either side in LDS (a node of <= 64 reaches no further, and one step beyond shows
// that its parent is too large).
__device__ __forceinline__ int adjacent_prefix(const uint64_t* __restrict__ keys, int n, int j) {   // delta(j, j + 1) of karras_range; -1 outside
  if (j < 0 || j + 1 >= n) return -1;
  const uint64_t a = keys[j], b = keys[j + 1];
  return a == b ? 64 + __clz((uint32_t)j ^ (uint32_t)(j + 1)) : __clzll((long long)(a ^ b));
}
__global__ __launch_bounds__(256) void k_roots_from_keys(const uint64_t* __restrict__ keys, int n, int2* __restrict__ roots,
                                                         int* __restrict__ n_roots) {
  // One thread per SPLIT j (between the positions j and j + 1): the node that splits there reaches from the nearest smaller adjacent
  // prefix on its left (exclusive) to the nearest smaller one on its right (inclusive).  Both are found by descending a table of
  // range minima over the block's window of prefixes (seven dependent LDS reads each; a flat loop that let the neighbours join one
  // by one took up to 64 x 2 and 13 of this kernel's 18 us).  The node is a treelet root iff it holds 3 .. TREELET_MAX triangles and its
  // parent — the node that splits at the larger of its two boundaries — holds more than TREELET_MAX.
  constexpr int HALO = 2 * TREELET_MAX + 2, WIN = 256 + 2 * HALO, LEVELS = 7;   // searches go 64 from j, then 65 from the parent's split
  __shared__ int s_m[LEVELS][WIN];   // s_m[k][i] = min d[i .. i + 2^k - 1]; beyond the window: -2
  __shared__ int s_wcnt[4], s_base;
  const int base = blockIdx.x * 256, lo = base - HALO;
  for (int t = threadIdx.x; t < WIN; t += 256) s_m[0][t] = adjacent_prefix(keys, n, lo + t);
  __syncthreads();
  for (int k = 1; k < LEVELS; ++k) {
    const int h = 1 << (k - 1);
    for (int t = threadIdx.x; t < WIN; t += 256) s_m[k][t] = t + h < WIN ? min(s_m[k - 1][t], s_m[k - 1][t + h]) : -2;
    __syncthreads();
  }
  auto right_of = [&](int i, int v) {   // first window index > i whose prefix is < v (WIN: none within reach)
    int pos = i + 1;
#pragma unroll
    for (int k = LEVELS - 1; k >= 0; --k)
      if (pos < WIN && s_m[k][pos] >= v) pos += 1 << k;
    return min(pos, WIN);
  };
  auto left_of = [&](int i, int v) {    // last window index < i whose prefix is < v (-1: none within reach)
    int pos = i - 1;
#pragma unroll
    for (int k = LEVELS - 1; k >= 0; --k) {
      const int from = pos - (1 << k) + 1;
      if (from >= 0 && s_m[k][from] >= v) pos = from - 1;
    }
    return max(pos, -1);
  };
  const int j = base + (int)threadIdx.x, w = j - lo;
  bool is_root = false;
  int first = 0, size = 0;
  if (j < n - 1) {
    const int v = s_m[0][w];
    const int R = right_of(w, v), L = left_of(w, v);
    size = R - L;
    if (size >= 3 && size <= TREELET_MAX && L >= 0 && R < WIN) {
      const int dl = s_m[0][L], dr = s_m[0][R];
      is_root = true;
      if (dl >= 0 || dr >= 0) {                                   // not the whole array: the parent splits at the larger boundary
        const int q = dl > dr ? L : R, vq = max(dl, dr);
        const int Rq = right_of(q, vq), Lq = left_of(q, vq);
        is_root = Lq < 0 || Rq >= WIN || Rq - Lq > TREELET_MAX;
      }
      first = lo + L + 1;
    }
  }
  const unsigned long long bal = __ballot(is_root);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) s_wcnt[wave] = __popcll(bal);
  __syncthreads();
  if (threadIdx.x == 0) {
    const int total = s_wcnt[0] + s_wcnt[1] + s_wcnt[2] + s_wcnt[3];
    s_base = total ? atomicAdd(n_roots, total) : 0;
  }
  __syncthreads();
  if (is_root) {
    int off = s_base + (int)__popcll(bal & ((1ull << lane) - 1ull));
    for (int w2 = 0; w2 < wave; ++w2) off += s_wcnt[w2];
    roots[off] = make_int2(first, size);
  }
}

// The items of a treelet are held BY POSITION IN THE LANES of one wave.  (Round 2 kept them in LDS arrays that every lane looped over —
// O(m^2) LDS reads per level, one LDS round trip per iteration: 62 us for 100 k triangles, 290 us for 1 M.)  A
// segment is a run of consecutive lanes, so its reductions are log-step scans bounded by the segment's ends ([s, e) is known to
// every lane, no head flags): extents of the centres, prefix / suffix boxes in rank order, the best (cost, rank).  The scan steps
// are DPP moves on the VALU (row_shr / row_shl inside a row of 16 lanes, row_bcast across rows for the prefixes; the suffixes
// cross rows with two lane reads) — a first version on ds_bpermute throughout was no faster than the loops (47 / 306 us): a lane
// permute occupies the LDS pipe like the reads it replaced.  The items move to their rank with seven lane permutes per level; only
// the rank itself is still counted by looking at every other item of the segment (one lane read each, four in flight).  Same
// splits, same arithmetic for the cost, same tie rules as the LDS form: the same keys and order, bit for bit.
template <int CTRL, int ROWS>
__device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, ROWS, 0xf, false); }   // lanes without a source keep v
template <int CTRL, int ROWS>
__device__ __forceinline__ float dpp_f(float v) { return __int_as_float(dpp_i<CTRL, ROWS>(__float_as_int(v))); }
constexpr int DPP_SHR = 0x110, DPP_BCAST15 = 0x142, DPP_BCAST31 = 0x143, DPP_WAVE_SHL1 = 0x130;

// The segmented scans of a level all run over the same segments, so the lanes that take part in a step are the same for every scan:
// twelve lane masks per level (SGPR pairs), and a step is ONE fused DPP min / max into a scratch register plus ONE v_cndmask under
// the step's mask.  (The compiler's form of `v = src_in_segment ? min(v, dpp(v)) : v` recomputed the comparison for every scan and
// moved before it took the minimum: five instructions per step, 600 of a level's ~1 000.)  Three scans are interleaved, so that the
// DPP read of a register comes two instructions after its last write inside a block (the VALU-write -> DPP-read hazard wants two wait
// states); the compiler does not pad inline assembly, so every block starts with an s_nop 1 against whatever it put in front.
struct ScanMasks {
  unsigned long long up[6];     // prefix steps: row_shr 1, 2, 4, 8, row_bcast 15, row_bcast 31
  unsigned long long down[4];   // suffix steps: row_shl 1, 2, 4, 8 (the two row crossings go through lane reads)
};
__device__ __forceinline__ ScanMasks scan_masks(int lane, int s, int e) {
  ScanMasks m;
  const int r = lane & 15;
  m.up[0] = __ballot(r >= 1 && lane - 1 >= s);
  m.up[1] = __ballot(r >= 2 && lane - 2 >= s);
  m.up[2] = __ballot(r >= 4 && lane - 4 >= s);
  m.up[3] = __ballot(r >= 8 && lane - 8 >= s);
  m.up[4] = __ballot((lane & 16) != 0 && (lane & ~15) - 1 >= s);
  m.up[5] = __ballot(lane >= 32 && 31 >= s);
  m.down[0] = __ballot(r + 1 <= 15 && lane + 1 < e);
  m.down[1] = __ballot(r + 2 <= 15 && lane + 2 < e);
  m.down[2] = __ballot(r + 4 <= 15 && lane + 4 < e);
  m.down[3] = __ballot(r + 8 <= 15 && lane + 8 < e);
  return m;
}
// The compiler pads no hazard around or inside inline assembly.  A block's own hazard is the VALU write of a, b, c by the block in front of it ->
// DPP read: two wait states (s_nop 1).  The FIRST block of a scan follows compiler-scheduled code, which may end in a VALU write of EXEC (v_cmpx,
// five wait states before a DPP instruction) or of a, b, c: it pads five (NOP = "4").  v_min / v_max_f32 return the other operand for a quiet NaN
// as fminf / fmaxf do; box extents never carry a signalling NaN (they come out of arithmetic), and the golden digests include NaN / inf boxes.
#define M2S_SCAN3(OP, CTRL, MASK) M2S_SCAN3_N("1", OP, CTRL, MASK)
#define M2S_SCAN3_FIRST(OP, CTRL, MASK) M2S_SCAN3_N("4", OP, CTRL, MASK)
#define M2S_SCAN3_N(NOP, OP, CTRL, MASK)                                                               \
  asm volatile("s_nop " NOP "\n\t"                                                                     \
               "v_" OP "_f32_dpp %3, %0, %0 " CTRL "\n\t"                                              \
               "v_" OP "_f32_dpp %4, %1, %1 " CTRL "\n\t"                                              \
               "v_" OP "_f32_dpp %5, %2, %2 " CTRL "\n\t"                                              \
               "v_cndmask_b32 %0, %0, %3, %6\n\t"                                                      \
               "v_cndmask_b32 %1, %1, %4, %6\n\t"                                                      \
               "v_cndmask_b32 %2, %2, %5, %6"                                                          \
               : "+v"(a), "+v"(b), "+v"(c), "=&v"(t0), "=&v"(t1), "=&v"(t2)                            \
               : "s"(MASK))
// inclusive scans towards higher lanes over [s, lane] of three values at once; MIN: minima, else maxima
template <bool MIN>
__device__ __forceinline__ void seg_prefix3(float& a, float& b, float& c, const ScanMasks& m) {
  float t0, t1, t2;
  if (MIN) {
    M2S_SCAN3_FIRST("min", "row_shr:1 row_mask:0xf bank_mask:0xf", m.up[0]);
    M2S_SCAN3("min", "row_shr:2 row_mask:0xf bank_mask:0xf", m.up[1]);
    M2S_SCAN3("min", "row_shr:4 row_mask:0xf bank_mask:0xf", m.up[2]);
    M2S_SCAN3("min", "row_shr:8 row_mask:0xf bank_mask:0xf", m.up[3]);
    M2S_SCAN3("min", "row_bcast:15 row_mask:0xa bank_mask:0xf", m.up[4]);
    M2S_SCAN3("min", "row_bcast:31 row_mask:0xc bank_mask:0xf", m.up[5]);
  } else {
    M2S_SCAN3_FIRST("max", "row_shr:1 row_mask:0xf bank_mask:0xf", m.up[0]);
    M2S_SCAN3("max", "row_shr:2 row_mask:0xf bank_mask:0xf", m.up[1]);
    M2S_SCAN3("max", "row_shr:4 row_mask:0xf bank_mask:0xf", m.up[2]);
    M2S_SCAN3("max", "row_shr:8 row_mask:0xf bank_mask:0xf", m.up[3]);
    M2S_SCAN3("max", "row_bcast:15 row_mask:0xa bank_mask:0xf", m.up[4]);
    M2S_SCAN3("max", "row_bcast:31 row_mask:0xc bank_mask:0xf", m.up[5]);
  }
}
// inclusive scans towards lower lanes over [lane, e)
template <bool MIN>
__device__ __forceinline__ void seg_suffix3(float& a, float& b, float& c, const ScanMasks& m, int lane, int e) {
  float t0, t1, t2;
  if (MIN) {
    M2S_SCAN3_FIRST("min", "row_shl:1 row_mask:0xf bank_mask:0xf", m.down[0]);
    M2S_SCAN3("min", "row_shl:2 row_mask:0xf bank_mask:0xf", m.down[1]);
    M2S_SCAN3("min", "row_shl:4 row_mask:0xf bank_mask:0xf", m.down[2]);
    M2S_SCAN3("min", "row_shl:8 row_mask:0xf bank_mask:0xf", m.down[3]);
  } else {
    M2S_SCAN3_FIRST("max", "row_shl:1 row_mask:0xf bank_mask:0xf", m.down[0]);
    M2S_SCAN3("max", "row_shl:2 row_mask:0xf bank_mask:0xf", m.down[1]);
    M2S_SCAN3("max", "row_shl:4 row_mask:0xf bank_mask:0xf", m.down[2]);
    M2S_SCAN3("max", "row_shl:8 row_mask:0xf bank_mask:0xf", m.down[3]);
  }
  {
    const int src = (lane | 15) + 1;                              // first lane of the next row (rows 0 and 2 take it)
    const bool take = !(lane & 16) && src < e;
    const float ta = __shfl(a, src & 63), tb = __shfl(b, src & 63), tc = __shfl(c, src & 63);
    a = take ? (MIN ? fminf(a, ta) : fmaxf(a, ta)) : a;
    b = take ? (MIN ? fminf(b, tb) : fmaxf(b, tb)) : b;
    c = take ? (MIN ? fminf(c, tc) : fmaxf(c, tc)) : c;
  }
  {
    const bool take = lane < 32 && 32 < e;
    const float ta = __shfl(a, 32), tb = __shfl(b, 32), tc = __shfl(c, 32);
    a = take ? (MIN ? fminf(a, ta) : fmaxf(a, ta)) : a;
    b = take ? (MIN ? fminf(b, tb) : fmaxf(b, tb)) : b;
    c = take ? (MIN ? fminf(c, tc) : fmaxf(c, tc)) : c;
  }
}

__global__ __launch_bounds__(64) void k_treelet_lanes(const int2* __restrict__ roots, const int* __restrict__ n_roots,
                                                      const Box* __restrict__ boxes, uint64_t* __restrict__ keys,
                                                      uint32_t* __restrict__ order) {
  // a few thousand single-wave blocks take the roots in turn (one block per POSSIBLE root — n / 3 of them, nine in ten with
  // nothing to do — spent more time being dispatched than the treelets took)
  const int total = *n_roots, lane = threadIdx.x;
  for (int r = blockIdx.x; r < total; r += gridDim.x) {
  const int2 root = roots[r];
  const int first = root.x, m = root.y;
  const bool live = lane < m;
  const uint64_t k_first = keys[first], k_last = keys[first + m - 1];
  if (k_first == k_last) continue;                                // identical centres: no room below the prefix
  const int prefix = __clzll((long long)(k_first ^ k_last));      // bits the node's keys share
  const int room = 64 - prefix;                                   // bits left for the path inside the treelet
  uint32_t tri = live ? order[first + lane] : 0u;
  Box b = {0, 0, 0, 0, 0, 0};
  if (live) b = boxes[tri];
  int s = lane, e = live ? m : lane + 1;                          // my segment [s, e) of lanes; I am the item at position `lane`
  if (live) s = 0;
  uint64_t code = 0;
  int depth = 0;
  const float inf = __builtin_inff();
  for (int level = 0; level < 64; ++level) {
    const bool open = live && e - s > 1 && depth < room;
    if (__ballot(open) == 0ull) break;
    // widest axis of the centres of my segment
    const float cen[3] = {0.5f * (b.mnx + b.mxx), 0.5f * (b.mny + b.mxy), 0.5f * (b.mnz + b.mxz)};
    const int tail = e - 1;
    const ScanMasks sm = scan_masks(lane, s, e);
    float ext[3];
    {
      float hi0 = cen[0], hi1 = cen[1], hi2 = cen[2], lo0 = cen[0], lo1 = cen[1], lo2 = cen[2];
      seg_prefix3<false>(hi0, hi1, hi2, sm);
      seg_prefix3<true>(lo0, lo1, lo2, sm);
      ext[0] = __shfl(hi0, tail) - __shfl(lo0, tail);
      ext[1] = __shfl(hi1, tail) - __shfl(lo1, tail);
      ext[2] = __shfl(hi2, tail) - __shfl(lo2, tail);
    }
    int axis = 2;
    if (open) axis = (ext[0] >= ext[1] && ext[0] >= ext[2]) ? 0 : (ext[1] >= ext[2] ? 1 : 2);   // NaN extents: comparisons false -> axis 2
    const int key = ord(axis == 0 ? cen[0] : (axis == 1 ? cen[1] : cen[2]));                    // total order, NaN included
    // Every item to the lane of its rank inside its segment along that axis (strict order by (key, position); closed segments stay
    // where they are).  Long segments: the wave sorts the composites (segment start, key, lane) with a 64-lane bitonic network — 21
    // compare-exchange steps whatever the segments — and every lane then pulls the item whose composite ended at its position; short ones
    // (the later levels): a lane counts the items of its segment that come before it, four lane reads at a time, and pushes its item to
    // that rank.  (The count alone was 76 instructions per four items: 1 200 of the first level's 1 900.)
    int longest = open ? e - s : 0;
    for (int o = 32; o > 0; o >>= 1) longest = max(longest, __shfl_xor(longest, o));
    if (longest > M2S_TREELET_SORT_ABOVE) {
      const uint32_t ukey = open ? (uint32_t)key ^ 0x80000000u : 0u;
      unsigned long long comp = ((unsigned long long)(uint32_t)s << 38) | ((unsigned long long)ukey << 6) | (unsigned long long)(uint32_t)lane;
#pragma unroll
      for (int k = 2; k <= 64; k <<= 1)
#pragma unroll
        for (int j = k >> 1; j >= 1; j >>= 1) {
          const uint32_t ohi = (uint32_t)__shfl_xor((int)(uint32_t)(comp >> 32), j), olo = (uint32_t)__shfl_xor((int)(uint32_t)comp, j);
          const unsigned long long other = ((unsigned long long)ohi << 32) | olo;
          const bool keep_min = ((lane & j) == 0) == ((lane & k) == 0);
          const bool take = keep_min ? other < comp : other > comp;
          comp = take ? other : comp;
        }
      const int src = (int)((uint32_t)comp & 63u) << 2;
      b.mnx = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(b.mnx)));
      b.mny = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(b.mny)));
      b.mnz = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(b.mnz)));
      b.mxx = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(b.mxx)));
      b.mxy = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(b.mxy)));
      b.mxz = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(b.mxz)));
      tri = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)tri);
    } else {
      int rank = lane, cnt = 0;
      for (int it = 0; it < longest; it += 4) {
        int oj[4];
        for (int u = 0; u < 4; ++u) oj[u] = __shfl(key, min(s + it + u, 63));
        for (int u = 0; u < 4; ++u) {
          const int j = s + it + u;
          cnt += (open && j < e && (oj[u] < key || (oj[u] == key && j < lane))) ? 1 : 0;
        }
      }
      if (open) rank = s + cnt;
      const int dst = rank << 2;
      b.mnx = __int_as_float(__builtin_amdgcn_ds_permute(dst, __float_as_int(b.mnx)));
      b.mny = __int_as_float(__builtin_amdgcn_ds_permute(dst, __float_as_int(b.mny)));
      b.mnz = __int_as_float(__builtin_amdgcn_ds_permute(dst, __float_as_int(b.mnz)));
      b.mxx = __int_as_float(__builtin_amdgcn_ds_permute(dst, __float_as_int(b.mxx)));
      b.mxy = __int_as_float(__builtin_amdgcn_ds_permute(dst, __float_as_int(b.mxy)));
      b.mxz = __int_as_float(__builtin_amdgcn_ds_permute(dst, __float_as_int(b.mxz)));
      tri = (uint32_t)__builtin_amdgcn_ds_permute(dst, (int)tri);
    }
    // the split after me: boxes of the items up to my position and of the rest
    float L[6] = {b.mnx, b.mny, b.mnz, b.mxx, b.mxy, b.mxz}, S[6] = {b.mnx, b.mny, b.mnz, b.mxx, b.mxy, b.mxz};
    seg_prefix3<true>(L[0], L[1], L[2], sm);
    seg_prefix3<false>(L[3], L[4], L[5], sm);
    seg_suffix3<true>(S[0], S[1], S[2], sm, lane, e);
    seg_suffix3<false>(S[3], S[4], S[5], sm, lane, e);
    float Rr[6];
    for (int k = 0; k < 6; ++k) Rr[k] = dpp_f<DPP_WAVE_SHL1, 0xf>(S[k]);   // S of lane + 1
    int cost_key = INT32_MAX;
    {
      const int n_left = lane - s + 1, n_right = e - lane - 1;
      float r0 = inf, r1 = inf, r2 = inf, r3 = -inf, r4 = -inf, r5 = -inf;
      if (n_right > 0) { r0 = Rr[0]; r1 = Rr[1]; r2 = Rr[2]; r3 = Rr[3]; r4 = Rr[4]; r5 = Rr[5]; }
      const float cost = ((L[3] - L[0]) + (L[4] - L[1]) + (L[5] - L[2])) * (float)n_left + ((r3 - r0) + (r4 - r1) + (r5 - r2)) * (float)n_right;
      if (open && n_right > 0) cost_key = ord(cost);              // the last rank is not a split
    }
    // the best split of my segment: smallest (cost, rank); all INT32_MAX keeps the fallback
    int bc = (int)((uint32_t)cost_key ^ 0x80000000u), br = lane;   // cost as unsigned order in a signed... compared as uint below
    {
#define M2S_STEP(CTRL, ROWS, SRC) { const uint32_t tc = (uint32_t)dpp_i<CTRL, ROWS>(bc); const int tr = dpp_i<CTRL, ROWS>(br); \
        const bool less = tc < (uint32_t)bc || (tc == (uint32_t)bc && tr < br); const bool ok = (SRC) >= s && less; bc = ok ? (int)tc : bc; br = ok ? tr : br; }
      M2S_STEP(DPP_SHR + 1, 0xf, lane - 1)
      M2S_STEP(DPP_SHR + 2, 0xf, lane - 2)
      M2S_STEP(DPP_SHR + 4, 0xf, lane - 4)
      M2S_STEP(DPP_SHR + 8, 0xf, lane - 8)
      M2S_STEP(DPP_BCAST15, 0xa, (lane & ~15) - 1)
      M2S_STEP(DPP_BCAST31, 0xc, 31)
#undef M2S_STEP
    }
    const uint32_t best_cost = (uint32_t)__shfl(bc, tail);
    const int best_at = __shfl(br, tail);
    if (open) {
      int best_rank = s + (e - s) / 2 - 1;
      if (best_cost != 0xffffffffu) best_rank = best_at;
      const bool right = lane > best_rank;
      code = (code << 1) | (right ? 1ull : 0ull);
      ++depth;
      if (right) s = best_rank + 1; else e = best_rank + 1;
    }
  }
  if (live) {
    // Morton prefix of the node (shared by all its keys), then the path inside the treelet, left aligned (depth <= room)
    const uint64_t mask = ~0ull << room;
    const uint64_t path = depth ? code << (room - depth) : 0ull;
    keys[first + lane] = (k_first & mask) | path;
    order[first + lane] = tri;
  }
  }
}

#include "lbvh_sort.hip.h"

// ======== the lean build ==========================================================================================
// The build is the part of a multi-GPU rank's step that does not shard, so it is a few FULL launches rather than many small ones
// (round 2: ~45 launch-bound kernels, 0.36 ms for 100 k triangles; tests/golden/build_digests.json pins the tree that sequence and
// this one both produced, byte for byte):
//   k_tri_setup                 records, boxes, centroids, per-block scene partials; clears the build's counters
//   k_sort_tiles / _rank / _buckets   keys + sample sort (lbvh_sort.hip.h); above 229 376 triangles: k_morton_keys (every block folds the scene
//                               partials itself and writes its tile's keys) + rocPRIM radix_sort_pairs (a block sort + merges)
//   k_roots_from_keys           treelet roots straight from the sorted keys
//   k_treelet_lanes             sweep-split treelets, one wave each, items in lanes
//   k_hierarchy                 the Karras ranges, once, over the rewritten keys; beside them ten segment-tree levels per block in LDS and
//                               the record-set scans, the last block to finish adding the top levels
//   k_emit, k_node_ext          pre-order records, oriented bounds
constexpr int KEY_THREADS = 512, KEY_CHUNK = KEY_THREADS * 8;
constexpr uint32_t KEY_MAX_TILES = 128;
// Tiles of the key kernel: at most KEY_MAX_TILES blocks (each folds all scene partials), each a whole number of 4096-triangle chunks.
__host__ __device__ inline uint32_t key_tile_pairs(size_t n) {
  const size_t chunks = (n + KEY_CHUNK - 1) / KEY_CHUNK;
  return (uint32_t)((chunks + KEY_MAX_TILES - 1) / KEY_MAX_TILES) * KEY_CHUNK;
}
__host__ __device__ inline uint32_t key_tiles(size_t n) {
  const uint32_t tp = key_tile_pairs(n);
  return tp ? (uint32_t)((n + tp - 1) / tp) : 0u;
}
constexpr size_t AUX_WORDS = 16;   // [0]: k_hierarchy's finished segment-tree blocks

// 63-bit Morton key of every triangle's box centre (21 bits per axis over the scene's box of centres) and the identity permutation.
// One block per tile; every block folds the per-block scene partials of k_tri_setup itself, block 0 publishes the final values
// (scene[0..5]: order-encoded min xyz / max xyz; scene[6]: largest finite |coordinate| as float bits; scene[7]: treelet root counter).
__global__ __launch_bounds__(KEY_THREADS) void k_morton_keys(const Box* __restrict__ boxes, uint32_t n_tris, int* __restrict__ scene,
                                                              uint32_t n_partials, uint64_t* __restrict__ keys, uint32_t* __restrict__ vals,
                                                              uint32_t tile_pairs) {
  __shared__ int s_part[6][KEY_THREADS / 64];
  __shared__ int s_scene[6];
  const uint32_t tid = threadIdx.x;
  int lo[3] = {INT32_MAX, INT32_MAX, INT32_MAX}, hi[3] = {INT32_MIN, INT32_MIN, INT32_MIN};
  for (uint32_t b = tid; b < n_partials; b += KEY_THREADS)   // a few KB out of L2
    for (int k = 0; k < 3; ++k) {
      lo[k] = min(lo[k], scene[8 + b * 6 + k]);
      hi[k] = max(hi[k], scene[8 + b * 6 + 3 + k]);
    }
  for (int k = 0; k < 3; ++k) {
    int l = lo[k], h = hi[k];
    for (int off = 32; off > 0; off >>= 1) {
      l = min(l, __shfl_xor(l, off));
      h = max(h, __shfl_xor(h, off));
    }
    if ((tid & 63u) == 0) { s_part[k][tid >> 6] = l; s_part[3 + k][tid >> 6] = h; }
  }
  __syncthreads();
  if (tid < 6) {
    int v = s_part[tid][0];
    for (int w = 1; w < KEY_THREADS / 64; ++w) v = tid < 3 ? min(v, s_part[tid][w]) : max(v, s_part[tid][w]);
    s_scene[tid] = v;
  }
  __syncthreads();
  if (blockIdx.x == 0 && tid == 0) {
    float s = 0.0f;
    for (int k = 0; k < 6; ++k) {
      scene[k] = s_scene[k];
      const float f = unord(s_scene[k]);
      if (f == f && fabsf(f) < 3.0e38f) s = fmaxf(s, fabsf(f));
    }
    scene[6] = __float_as_int(s);
    scene[7] = 0;
  }
  float slo[3], shi[3];
  for (int k = 0; k < 3; ++k) { slo[k] = unord(s_scene[k]); shi[k] = unord(s_scene[3 + k]); }
  const uint32_t base = blockIdx.x * tile_pairs, end = min(n_tris, base + tile_pairs);
  for (uint32_t t = base + tid; t < end; t += KEY_THREADS) {
    const Box bx = boxes[t];
    const float c[3] = {0.5f * (bx.mnx + bx.mxx), 0.5f * (bx.mny + bx.mxy), 0.5f * (bx.mnz + bx.mxz)};
    uint32_t q[3];
    for (int k = 0; k < 3; ++k) {
      float u = (c[k] - slo[k]) / (shi[k] - slo[k]);
      u = (u == u) ? fminf(fmaxf(u, 0.0f), 1.0f) : 0.0f;
      q[k] = min((uint32_t)(u * 2097152.0f), 2097151u);
    }
    keys[t] = (expand21(q[0]) << 2) | (expand21(q[1]) << 1) | expand21(q[2]);
    vals[t] = t;
  }
}

// One launch, two independent jobs over the final keys and order (after the treelet pass):
//  * workgroups [0, seg_blocks): the segment tree of the leaf boxes in two steps — every block builds the levels 0..SEG_LOCAL of its
//    512 leaves in LDS, and the block that finishes last (a ticket behind a device-scope fence) adds the levels above from the blocks'
//    tops (unions are min / max: any order gives the same boxes) — and the prefix-minimum record sets of the adjacent prefixes (Recs
//    above): scanned from the right inside the block's 512 positions (recs_local), the block's total handed to the last block, which
//    scans the totals from the right over the blocks (recs_carry[b] = the records of everything right of block b);
//  * the others: the Karras ranges of the internal nodes, one thread each.
// (Round 4: k_karras, then k_seg_build — 15.5 + 11.3 us at 100 k triangles, neither filling the chip.)
constexpr int SEG_LOCAL = 9;
__global__ __launch_bounds__(256) void k_hierarchy(const Box* __restrict__ boxes, const uint32_t* __restrict__ order, uint32_t n,
                                                   Box* __restrict__ seg, SegLevels lv, uint32_t* __restrict__ done, uint32_t seg_blocks,
                                                   const uint64_t* __restrict__ keys, int2* __restrict__ range,
                                                   uint4* __restrict__ recs_local, uint4* __restrict__ recs_total, uint4* __restrict__ recs_carry) {
  if (blockIdx.x >= seg_blocks) {
    const int i = (int)((blockIdx.x - seg_blocks) * 256u + threadIdx.x);
    if (i < (int)n - 1) karras_range(keys, (int)n, i, range);
    return;
  }
  __shared__ Box s[1024], s2[512];
  __shared__ uint4 s_r[2][RECS_BLOCK];
  __shared__ bool s_last;
  const uint32_t t = threadIdx.x, leaf0 = blockIdx.x * 512u;
  for (uint32_t k = t; k < 512u; k += 256u) {
    const uint32_t i = leaf0 + k;
    if (i < n) { const Box b = boxes[order[i]]; s[k] = b; seg[i] = b; }
    s_r[0][k] = recs_pack(recs_of(adjacent_prefix(keys, (int)n, (int)i)));   // -1 (no bit) from the last position on
  }
  __syncthreads();
  // records of [k, 512) inside the block: log-step scan from the right, ping-pong
  int cur = 0;
  for (uint32_t h = 1; h < RECS_BLOCK; h <<= 1) {
    for (uint32_t k = t; k < RECS_BLOCK; k += 256u) {
      const Recs a = recs_unpack(s_r[cur][k]);
      s_r[cur ^ 1][k] = k + h < RECS_BLOCK ? recs_pack(recs_join(a, recs_unpack(s_r[cur][k + h]))) : s_r[cur][k];
    }
    __syncthreads();
    cur ^= 1;
  }
  for (uint32_t k = t; k < RECS_BLOCK; k += 256u)
    if (leaf0 + k < n) recs_local[leaf0 + k] = s_r[cur][k];
  const bool single = lv.levels <= SEG_LOCAL + 1;           // one block: it is its own "last block"
  if (t == 0 && !single) {
    uint32_t* dst = reinterpret_cast<uint32_t*>(recs_total + blockIdx.x);
    const uint4 v = s_r[cur][0];
    __hip_atomic_store(dst + 0, v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(dst + 1, v.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(dst + 2, v.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (single && t == 0) recs_carry[0] = make_uint4(0u, 0u, 0u, 0u);
  for (int l = 1; l <= SEG_LOCAL && l < lv.levels; ++l) {
    const uint32_t m = 512u >> l, gj = (leaf0 >> l) + t;
    Box u{};
    const bool have = t < m && gj < lv.cnt[l];
    if (have) {
      u = s[2 * t];
      if (2 * gj + 1 < lv.cnt[l - 1]) u = box_union(u, s[2 * t + 1]);
    }
    __syncthreads();
    if (have) {
      s[t] = u;
      if (l < SEG_LOCAL || lv.levels <= SEG_LOCAL + 1) seg[lv.off[l] + gj] = u;
      else {
        // the block's top entry is what the last block reads: written through to memory (device scope) and acknowledged before
        // the block counts itself — no L2 write-back fence (a release fence per block made this kernel 171 us for 1 M triangles)
        uint32_t* dst = reinterpret_cast<uint32_t*>(seg + lv.off[l] + gj);
        const float f[6] = {u.mnx, u.mny, u.mnz, u.mxx, u.mxy, u.mxz};
        for (int k = 0; k < 6; ++k) __hip_atomic_store(dst + k, __float_as_uint(f[k]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    __syncthreads();
  }
  if (single) return;
  if (t == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    s_last = __hip_atomic_fetch_add(done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == seg_blocks - 1u;
  }
  __syncthreads();
  if (!s_last) return;
  // ---- the last block: the record sets right of every block, chunks of RECS_BLOCK blocks from the right
  {
    Recs right = {0ull, 0u};                                 // records of everything right of the current chunk
    for (int c0 = (int)((seg_blocks - 1u) / RECS_BLOCK * RECS_BLOCK); c0 >= 0; c0 -= (int)RECS_BLOCK) {
      for (uint32_t k = t; k < RECS_BLOCK; k += 256u) {
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if ((uint32_t)c0 + k < seg_blocks) {
          const uint32_t* src = reinterpret_cast<const uint32_t*>(recs_total + c0 + k);
          v.x = __hip_atomic_load(src + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          v.y = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          v.z = __hip_atomic_load(src + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        s_r[0][k] = v;
      }
      __syncthreads();
      int cu = 0;
      for (uint32_t h = 1; h < RECS_BLOCK; h <<= 1) {
        for (uint32_t k = t; k < RECS_BLOCK; k += 256u) {
          const Recs a = recs_unpack(s_r[cu][k]);
          s_r[cu ^ 1][k] = k + h < RECS_BLOCK ? recs_pack(recs_join(a, recs_unpack(s_r[cu][k + h]))) : s_r[cu][k];
        }
        __syncthreads();
        cu ^= 1;
      }
      // s_r[cu][k] = records of the blocks [c0 + k, end of chunk); block c0 + k wants the blocks right of it
      for (uint32_t k = t; k < RECS_BLOCK; k += 256u)
        if ((uint32_t)c0 + k < seg_blocks) {
          Recs in_chunk = {0ull, 0u};
          if (k + 1u < RECS_BLOCK) in_chunk = recs_unpack(s_r[cu][k + 1u]);
          recs_carry[c0 + k] = recs_pack(recs_join(in_chunk, right));
        }
      const Recs whole = recs_join(recs_unpack(s_r[cu][0]), right);
      __syncthreads();
      right = whole;
    }
  }
  auto load_top = [&](uint32_t j) {          // level SEG_LOCAL, written by other blocks: read past this XCD's L2
    const uint32_t* src = reinterpret_cast<const uint32_t*>(seg + lv.off[SEG_LOCAL] + j);
    float f[6];
    for (int k = 0; k < 6; ++k) f[k] = __uint_as_float(__hip_atomic_load(src + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    return Box{f[0], f[1], f[2], f[3], f[4], f[5]};
  };
  int l = SEG_LOCAL + 1;
  for (; l < lv.levels && lv.cnt[l - 1] > 1024u; ++l) {       // too wide for LDS: through memory (meshes above 512 k triangles)
    for (uint32_t j = t; j < lv.cnt[l]; j += 256u) {
      Box u = l == SEG_LOCAL + 1 ? load_top(2 * j) : seg[lv.off[l - 1] + 2 * j];
      if (2 * j + 1 < lv.cnt[l - 1]) u = box_union(u, l == SEG_LOCAL + 1 ? load_top(2 * j + 1) : seg[lv.off[l - 1] + 2 * j + 1]);
      seg[lv.off[l] + j] = u;
    }
    __syncthreads();
  }
  if (l >= lv.levels) return;
  for (uint32_t j = t; j < lv.cnt[l - 1]; j += 256u) s[j] = l == SEG_LOCAL + 1 ? load_top(j) : seg[lv.off[l - 1] + j];
  __syncthreads();
  Box* from = s;
  Box* to = s2;
  for (; l < lv.levels; ++l) {                                 // the rest in LDS, every level written out as it appears
    const uint32_t m = lv.cnt[l - 1];
    for (uint32_t j = t; j < lv.cnt[l]; j += 256u) {
      Box u = from[2 * j];
      if (2 * j + 1 < m) u = box_union(u, from[2 * j + 1]);
      to[j] = u;
      seg[lv.off[l] + j] = u;
    }
    __syncthreads();
    Box* tmp = from; from = to; to = tmp;
  }
}

inline unsigned cdiv(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }

}  // namespace

size_t bvh_workspace_bytes(size_t n_tris) {
  const size_t n = n_tris ? n_tris : 1;
  size_t sort_tmp = 0;
  (void)sort_pairs_u64(nullptr, sort_tmp, (const uint64_t*)nullptr, (uint64_t*)nullptr, (const uint32_t*)nullptr,
                            (uint32_t*)nullptr, n, 0, 64, (hipStream_t)0);
  size_t b = AUX_WORDS * 4 + 256;
  b += n * 48 + 256 + n * sizeof(TriRec) * 2 + n * 16 + 256 + n * 16 + 256 + n * 16 + 256 + n * 4 + 256 + n * sizeof(TriPlanes) + 256 + n * sizeof(Box) * 3 + n * (8 + 4) * 2 + sort_tmp;
  b += n * (sizeof(int2) * 2) + n * 16 + 2 * (n / 512 + 2) * 16 + 768 + 2 * n * sizeof(NodeRec) + 2 * n * (sizeof(NodeExt) + 4);
  return b + 64 * 256 + 4096 + 24 * ((n + 255) / 256) + 256 + sample_sort_bytes(n);
}

int build_device_mesh(Arena& ws, hipStream_t st, const float* d_verts, size_t n_verts, const void* d_indices,
                      size_t n_indices, int index_bytes, int topology, size_t n_tris, int* d_err, DeviceMesh* out,
                      const std::function<int(const float4*, const TriRec*, int)>* after_setup, bool records_only, uint32_t leaf_max, uint64_t job_cells) {
  (void)n_indices;
  out->cen_raw = nullptr;
  out->slot_of = nullptr;
  out->tris = nullptr;
  out->corners = nullptr;
  out->cen = nullptr;
  out->planes = nullptr;
  out->nodes = nullptr;
  out->ext = nullptr;
  out->stats = nullptr;
  out->scene = nullptr;
  out->leaf_max = 2;
  out->slot_first = nullptr;
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
  float4* corners = ws.take<float4>(3 * n_tris);
  float4* nrm = ws.take<float4>(n_tris);
  Box* boxes = ws.take<Box>(n_tris);
  Box* seg = ws.take<Box>(2 * n_tris + 64);
  uint64_t* keys = ws.take<uint64_t>(n_tris);
  uint64_t* keys2 = ws.take<uint64_t>(n_tris);
  uint32_t* vals = ws.take<uint32_t>(n_tris);
  uint32_t* order = ws.take<uint32_t>(n_tris);
  int2* range = ws.take<int2>(n_tris);
  int2* child = ws.take<int2>(n_tris);
  uint4* recs_local = ws.take<uint4>(n_tris);
  uint4* recs_total = ws.take<uint4>((n_tris + RECS_BLOCK - 1) / RECS_BLOCK);
  uint4* recs_carry = ws.take<uint4>((n_tris + RECS_BLOCK - 1) / RECS_BLOCK);
  NodeRec* nodes = ws.take<NodeRec>(2 * n_tris);
  NodeExt* ext = ws.take<NodeExt>(2 * n_tris);
  uint32_t* slot_first = ws.take<uint32_t>(2 * n_tris);
  int* scene = ws.take<int>(8 + 6 * ((n_tris + 255) / 256));
  uint32_t* aux = ws.take<uint32_t>(AUX_WORDS);   // [0]: k_hierarchy's finished segment-tree blocks
  size_t sort_tmp = 0;
  (void)sort_pairs_u64(nullptr, sort_tmp, keys, keys2, vals, order, n_tris, 0, 64, st);
  void* tmp = ws.take<char>(sort_tmp ? sort_tmp : 1);
  const uint32_t sort_w = sort_tile_size(n_tris);
  SortBufs sb;
  if (sort_w && n_tris > sort_w) {
    const size_t p = (n_tris + sort_w - 1) / sort_w;
    sb.tiles = ws.take<uint4>(p * sort_w);
    sb.samples = ws.take<uint4>(p * (sort_w / SS_G));
    sb.splitters = ws.take<uint4>(p * SS_PER_TILE);
    sb.cmat = ws.take<uint8_t>(p * p * SS_PER_TILE);
    if (!sb.tiles || !sb.samples || !sb.splitters || !sb.cmat) {
      set_error("internal: BVH workspace too small");
      return M2S_ERR_HIP_INTERNAL;
    }
  }
  if (!raw || !tris || !boxes || !seg || !keys || !keys2 || !vals || !order || !range || !child || !recs_local || !recs_total || !recs_carry || !nodes ||
      !scene || !tmp || !ext || !aux || !cen_raw || !slot_of || !slot_first || !cen || !planes || !corners || !nrm) {
    set_error("internal: BVH workspace too small");
    return M2S_ERR_HIP_INTERNAL;
  }

  const unsigned B = 256;
  // triangles per collapsed leaf: 2 is the optimum of every BASELINE config (round 3, blob-1M in 512^3: 1 / 2 / 3 / 4 / 6 / 8 -> 27.1 / 25.2 /
  // 25.5 / 26.1 / 27.2 / 29.0 ms of walk); coarse grids over fine meshes would take 4 - 8 for 5 - 10 %, but a persistent mesh has no grid
  // (round 4: with the leaf work queued and run densely — distance.hip DeferQueue — a grid coarse against the mesh takes 4 or 8: the caller's
  // `leaf_max`, grid_leaf_max)
  leaf_max = std::min(std::max(leaf_max, 1u), 16u);
  out->leaf_max = leaf_max;
  hipLaunchKernelGGL(k_tri_setup, dim3(cdiv(n_tris, B)), dim3(B), 0, st, d_verts, (uint32_t)n_verts, d_indices,
                     index_bytes, topology, (uint32_t)n_tris, raw, boxes, cen_raw, scene, 8, d_err, records_only ? nullptr : aux, (uint32_t)AUX_WORDS);
  if (records_only) {
    // a tiny problem (grid_is_tiny): all voxels x all triangles needs the triangle records and nothing else — no keys, no sort, no tree
    M2S_HIP_CHECK(hipGetLastError());
    out->cen_raw = cen_raw;
    out->tris = raw;
    out->n_nodes = 0;
    return 0;
  }
  out->cen_raw = cen_raw;
  out->slot_of = slot_of;
  if (after_setup) {   // the caller's seed passes only need the centroids: they run beside the sort and the hierarchy
    const int rc = (*after_setup)(cen_raw, raw, 0);   // phase 0: mark this point of the stream (an event), launch nothing yet
    if (rc) return rc;
  }
  if (sort_w == 1024u) launch_sample_sort<1024>(st, boxes, (uint32_t)n_tris, scene, (uint32_t)cdiv(n_tris, B), sb, keys2, order, d_err);
  else if (sort_w == 2048u) launch_sample_sort<2048>(st, boxes, (uint32_t)n_tris, scene, (uint32_t)cdiv(n_tris, B), sb, keys2, order, d_err);
  else {
    hipLaunchKernelGGL(k_morton_keys, dim3(key_tiles(n_tris)), dim3(KEY_THREADS), 0, st, boxes, (uint32_t)n_tris, scene, (uint32_t)cdiv(n_tris, B), keys, vals, key_tile_pairs(n_tris));
    M2S_HIP_CHECK(sort_pairs_u64(tmp, sort_tmp, keys, keys2, vals, order, n_tris, 0, 64, st));
  }
  if (after_setup) {
    // phase 1: `st` now holds ~100 us of work (keys, sort) — the time the host needs to enqueue the side work (the seed
    // passes, behind the phase-0 mark).  Launching it at phase 0 left `st` idle for those ~100 us (a launch costs the host ~8 us,
    // and the thin slab of a multi-GPU rank has nothing to hide that behind); launching it after the whole build serialised the
    // passes of a 512^3 lattice (0.5 ms) behind the build instead of beside it.
    const int rc = (*after_setup)(cen_raw, raw, 1);
    if (rc) return rc;
  }
  // A tree whose leaves hold 8 - 16 triangles keeps the top two or three levels of a 64-triangle treelet, and its walk is short: the pass (roots + treelets, 55 us
  // of the build's 250 at 100 k triangles, 300 of 1 170 at 1 M) costs such a call more than its ~5 % of the walk (whole call with / without, tools/exp_treelets.py:
  // blob-100k 64^3 0.88 / 0.85 ms, 128^3 0.82 / 0.78; blob-1M 128^3 3.41 / 3.17, 256^3 6.26 / 6.17; blob-11k 48^3 0.43 / 0.40).  With leaves of 4 it is a wash at
  // 100 k triangles (160^3 1.13 / 1.14), a loss at 1 M (384^3 11.3 / 11.6) and still a gain for small meshes (blob-11k 64^3 - 96^3 0.39 / 0.37): M2S_TREELETS -1.
  const int treelets = tuning().treelets;
  // ... and a one-shot call over few cells walks too little for its 5 % to be worth the pass's 35 - 45 us (round 5, whole call with / without:
  // suzanne, 968 triangles, Normal sign, 48^3 0.214 / 0.187 ms, 192^3 0.467 / 0.432, 256^3 0.685 / 0.661; blob-11k 128^3 0.419 / 0.401, 160^3 0.529 / 0.525,
  // 256^3 0.931 / 0.937)
  // (blob-100k 160^3, 4.1 M cells: 1.022 with / 1.048 without — the limit is 3 M)
  const bool few_cells = job_cells < (3u << 20) || (n_tris < 4096u && job_cells < (32u << 20));
  const bool skip_treelets = leaf_max >= 8u || (leaf_max >= 4u && n_tris < 32768u) || few_cells;
  if (n > 2 && (treelets > 0 || (treelets < 0 && !skip_treelets))) {
    // treelet pass: the nodes of at most TREELET_MAX triangles are rebuilt by sweep splits (their keys rewritten), then the hierarchy is derived
    int2* roots = reinterpret_cast<int2*>(child);   // child[]: a scratch array nothing else uses
    hipLaunchKernelGGL(k_roots_from_keys, dim3(cdiv(n_tris, B)), dim3(B), 0, st, (const uint64_t*)keys2, n, roots, scene + 7);
    hipLaunchKernelGGL(k_treelet_lanes, dim3((unsigned)std::min<size_t>((n_tris + 2) / 3, 8192)), dim3(64), 0, st, roots, scene + 7, boxes, keys2, order);
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
  hipLaunchKernelGGL(k_hierarchy, dim3(cdiv(n_tris, 512) + cdiv(n_tris - 1, B)), dim3(B), 0, st, (const Box*)boxes, (const uint32_t*)order, (uint32_t)n_tris, seg, lv, aux,
                     cdiv(n_tris, 512), (const uint64_t*)keys2, range, recs_local, recs_total, recs_carry);
  hipLaunchKernelGGL(k_emit, dim3(cdiv(2 * n_tris - 1, B)), dim3(B), 0, st, n, (const int2*)range, (const uint4*)recs_local, (const uint4*)recs_carry, (const Box*)seg, lv, order, raw,
                     nodes, tris, slot_first, cen, planes, leaf_max, slot_of, corners, nrm, d_err);
  hipLaunchKernelGGL(k_node_ext, dim3(cdiv(2 * n_tris - 1, B)), dim3(B), 0, st, nodes, slot_first, (const float4*)nrm, (const float4*)corners,
                     (uint32_t)(2 * n_tris - 1), ext);
  M2S_HIP_CHECK(hipGetLastError());
  out->tris = tris;
  out->slot_first = slot_first;
  out->corners = corners;
  out->cen = cen;
  out->planes = planes;
  out->nodes = nodes;
  out->ext = ext;
  out->scene = scene;
  return 0;
}

__global__ __launch_bounds__(256) void k_releaf(NodeRec* __restrict__ nodes, NodeExt* __restrict__ ext, const uint32_t* __restrict__ slot_first,
                                                uint32_t n_nodes, uint32_t leaf_max) {
  const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
  if (slot >= n_nodes) return;
  const uint32_t cnt = (nodes[slot].skip - slot + 1u) >> 1;
  const int tri = cnt <= leaf_max ? (int)slot_first[slot] : -1;
  nodes[slot].tri = tri;
  ext[slot].tri = tri;
}
int set_leaf_size(hipStream_t st, DeviceMesh* mesh, uint32_t leaf_max) {
  leaf_max = std::min(std::max(leaf_max, 1u), 16u);
  if (mesh->n_nodes == 0 || mesh->slot_first == nullptr || mesh->leaf_max == leaf_max) return 0;
  hipLaunchKernelGGL(k_releaf, dim3(cdiv(mesh->n_nodes, 256)), dim3(256), 0, st, const_cast<NodeRec*>(mesh->nodes), const_cast<NodeExt*>(mesh->ext),
                     mesh->slot_first, mesh->n_nodes, leaf_max);
  M2S_HIP_CHECK(hipGetLastError());
  mesh->leaf_max = leaf_max;
  return 0;
}

// m2s_warmup: the first launch of a kernel of this translation unit makes the runtime load its code object (all its kernels).
__global__ void k_warm_bvh() {}
void warm_bvh(hipStream_t st) {
  hipLaunchKernelGGL(k_warm_bvh, dim3(1), dim3(64), 0, st);
  // ... and resolves a kernel FUNCTION at its own first launch (~0.3 ms each): ask for the attributes of the ones a first call uses
  const void* fns[] = {
      (const void*)k_tri_setup,
      (const void*)k_sort_tiles<1024>,
      (const void*)k_sort_rank<1024>,
      (const void*)k_sort_buckets<1024>,
      (const void*)k_sort_tiles<2048>,
      (const void*)k_sort_rank<2048>,
      (const void*)k_sort_buckets<2048>,
      (const void*)k_morton_keys,
      (const void*)k_roots_from_keys,
      (const void*)k_treelet_lanes,
      (const void*)k_hierarchy,
      (const void*)k_emit,
      (const void*)k_node_ext};
  hipFuncAttributes attr;
  for (const void* f : fns) (void)hipFuncGetAttributes(&attr, f);
  (void)hipGetLastError();
}

}  // namespace m2s
