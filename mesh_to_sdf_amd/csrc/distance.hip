// distance.hip — nearest-triangle distance + sign resolve, the dominant kernels (gfx950).
//
// Semantics: D(p) = min over ALL triangles of geo.rs:26-30 in the reference's
// f32 operation order; sign by the rule the caller's SignMethod / AccelerationMethod selects.
//
// Two kernel families, same arithmetic (geo.hip.h):
//   k_brute  : every point against every triangle, triangle records staged through LDS in tiles
//              (AccelerationMethod::None, and the on-device cross-check of the BVH path).
//   k_packet : one wave = 64 spatially adjacent points (a 4x4x4 brick of voxels, or 64 Morton-
//              sorted queries).  The wave walks the stackless pre-order BVH TOGETHER: the node
//              index is wave-uniform (SGPR), node and triangle records arrive through scalar
//              loads, each lane tests its own point, and a subtree is skipped when the ballot of
//              "my bound reaches this box" is empty.  No per-lane stack, no divergence, no gather.
//
// Pruning is conservative: a subtree is dropped only if its box is farther than the lane's
// current best by a margin that covers f32 rounding of both the box test and the reference
// arithmetic (see prune_bound), so the minimum is the brute-force minimum bit for bit.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "common.h"
#include "geo.hip.h"
#include "tuning.h"

namespace m2s {

namespace {

constexpr float F32_MAX_C = 3.402823466e+38f;
constexpr int TILE = 128;  // triangles per LDS tile in k_brute (12 KiB)

// Record `index` of a read-only array through a 32-bit BYTE offset: a wave-uniform offset then goes straight into
// the scalar load's offset operand (no 64-bit address arithmetic in the walk loops).  Arrays stay below 4 GiB:
// 96 B x n_tris with n_tris < 2^25 (checked by the build).
// The records are read through the CONSTANT address space: the mesh arrays are never written while a walk runs, and only for a
// constant-address-space load does the compiler keep a wave-uniform address on the scalar unit (s_load) whatever else the kernel
// does — a global-address-space load falls back to the vector unit as soon as the kernel stores or performs an atomic anywhere
// before it ("may be clobbered"), which is what the suspension path of the split walk does (first version: every node record
// through global_load, walk 7.9 -> 17.2 ms).
template <class T>
__device__ __forceinline__ T record_at_bytes(const void* base, uint32_t byte_offset) {
  static_assert(sizeof(T) % 16 == 0 && alignof(T) >= 16, "records are whole 16-byte words");
  // builtin vectors (they load from any address space), declared 16-byte aligned: one s_load_dwordx16 / x8 / x4 each
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  typedef uint32_t u32x8 __attribute__((ext_vector_type(8), aligned(16)));
  typedef uint32_t u32x16 __attribute__((ext_vector_type(16), aligned(16)));
  typedef const __attribute__((address_space(4))) char* const_bytes;
  const const_bytes q = (const_bytes)(uintptr_t)base + byte_offset;
  constexpr unsigned N16 = sizeof(T) / 64, R16 = sizeof(T) % 64, N8 = R16 / 32, N4 = (R16 % 32) / 16;
  union { T rec; unsigned char raw[sizeof(T)]; } u;
#pragma unroll
  for (unsigned k = 0; k < N16; ++k) { const u32x16 v = *(const __attribute__((address_space(4))) u32x16*)(q + 64 * k); __builtin_memcpy(u.raw + 64 * k, &v, 64); }
#pragma unroll
  for (unsigned k = 0; k < N8; ++k) { const u32x8 v = *(const __attribute__((address_space(4))) u32x8*)(q + 64 * N16 + 32 * k); __builtin_memcpy(u.raw + 64 * N16 + 32 * k, &v, 32); }
#pragma unroll
  for (unsigned k = 0; k < N4; ++k) { const u32x4 v = *(const __attribute__((address_space(4))) u32x4*)(q + 64 * N16 + 32 * N8 + 16 * k); __builtin_memcpy(u.raw + 64 * N16 + 32 * N8 + 16 * k, &v, 16); }
  return u.rec;
}
template <class T>
__device__ __forceinline__ T record_at(const T* base, uint32_t index) {
  return record_at_bytes<T>(base, index * (uint32_t)sizeof(T));
}

// The 96-byte triangle records of the exact evaluation are read through the VECTOR path although their address
// is wave-uniform: a uniform-address vector load is a broadcast out of the 32 KB vector L1, lands in VGPRs, waits
// on the in-order vmcnt — and keeps the 16 KB scalar cache for the node records (and the 64-byte pre-test planes)
// of the walk.  Measured on 512^3 x blob-100k: both leaf records scalar 14.9 ms (kernel), both vector 14.05,
// planes scalar + triangle vector 13.67, the other way round 14.36.  The index is laundered through a VGPR so that
// the compiler does not turn the load back into a scalar one.
template <class T>
__device__ __forceinline__ T record_at_vec(const T* base, uint32_t uniform_index) {
  uint32_t vi;
  asm("v_mov_b32_e32 %0, %1" : "=v"(vi) : "s"(uniform_index));
  return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + (size_t)vi * sizeof(T));
}

// ---- point sources -------------------------------------------------------------------------
struct GridBrick {
  uint32_t x, y, z;
  uint32_t bx, by, bz;
  bool in_range;
  bool brick_in_grid;
};

// Brick sequence number -> brick coordinates.  Bricks are walked in 8x8x8 super-bricks (z fastest inside
// and between them), so that consecutive packets of an XCD keep touching the same part of the BVH while
// its 4 MiB L2 still holds it; a plain z-y-x sweep returns to a node only after a whole z column.
// A thin x-slab (multi-GPU pieces are 16 layers = 4 bricks at 8 GPUs x 4 chunks) uses super-bricks that are
// 4, 2 or 1 bricks wide in x instead, so that at most 1/8 of the launched packets are padding.
__host__ __device__ __forceinline__ uint32_t bricks_along(uint32_t cells, uint32_t log2_extent) {
  return (cells + (1u << log2_extent) - 1u) >> log2_extent;
}
__host__ __device__ __forceinline__ uint32_t super_brick_xlog(uint32_t nbx, uint32_t xl_cap = 0) {
  // (the compiler also emits an eight-wide "vectorised" form of this loop for trip counts >= 16, which never runs: xl starts at 3 or below.
  // A straight-line rewrite — round 6 — executed the same handful of scalar instructions and moved the packet walk's code by 1 400 bytes:
  // 6.44 -> 6.48 ms on the headline, same box, three runs each.  Left as it was.)
  for (uint32_t xl = xl_cap ? xl_cap - 1u : 3u; xl > 0; --xl) {
    const uint32_t padded = ((nbx + (1u << xl) - 1u) >> xl) << xl;
    if ((padded - nbx) * 8u <= nbx) return xl;
  }
  return 0;
}
__device__ __forceinline__ uint32_t div_magic(uint32_t n, uint32_t d, uint32_t magic) {
  return magic ? __umulhi(n, magic) : n / d;   // common.h set_super_brick_magic
}
__device__ __forceinline__ void brick_coords(const GridParams& g, uint32_t brick, uint32_t* bx, uint32_t* by, uint32_t* bz) {
  const uint32_t nby = bricks_along(g.n[1], g.bl[1]), nbz = bricks_along(g.n[2], g.bl[2]);
  const uint32_t sy = (nby + 7) >> 3, sz = (nbz + 7) >> 3;
  const uint32_t xl = super_brick_xlog(bricks_along(g.xe - g.xb, g.bl[0]), g.xl_cap);
  const uint32_t sb = brick >> (6 + xl), in = brick & ((64u << xl) - 1u);   // super-brick index, position inside (padded grid)
  const uint32_t t = div_magic(sb, sz, g.sz_magic), sbz = sb - t * sz;
  const uint32_t sbx = div_magic(t, sy, g.sy_magic), sby = t - sbx * sy;
  *bx = (sbx << xl) + (in >> 6);
  *by = sby * 8 + ((in >> 3) & 7u);
  *bz = sbz * 8 + (in & 7u);
}
__device__ __forceinline__ GridBrick grid_lane_voxel(const GridParams& g, uint32_t brick, int lane) {
  uint32_t bx, by, bz;
  brick_coords(g, brick, &bx, &by, &bz);
  GridBrick v;
  const uint32_t lx = g.bl[0], ly = g.bl[1], lz = g.bl[2], l = (uint32_t)lane;   // z in the low bits: the fastest axis
  const uint32_t layers = g.xe - g.xb, xv = (bx << lx) + (l >> (ly + lz));   // layer of the (virtual) slab
  v.y = (by << ly) + ((l >> lz) & ((1u << ly) - 1u));
  v.z = (bz << lz) + (l & ((1u << lz) - 1u));
  v.in_range = xv < layers && v.y < g.n[1] && v.z < g.n[2];
  v.brick_in_grid = ((bx << lx) < layers) && ((by << ly) < g.n[1]) && ((bz << lz) < g.n[2]);
  v.bx = bx; v.by = by; v.bz = bz;
  v.x = slab_x(g, min(xv, layers - 1u));
  v.y = min(v.y, g.n[1] - 1);
  v.z = min(v.z, g.n[2] - 1);
  return v;
}
__device__ __forceinline__ f3 grid_point(const GridParams& g, const GridBrick& v) {
  return {cell_center(g.first[0], g.size[0], v.x), cell_center(g.first[1], g.size[1], v.y),
          cell_center(g.first[2], g.size[2], v.z)};  // grid.rs:135-141
}
__device__ __forceinline__ uint32_t grid_brick_count(const GridParams& g) {
  return bricks_along(g.xe - g.xb, g.bl[0]) * bricks_along(g.n[1], g.bl[1]) * bricks_along(g.n[2], g.bl[2]);
}

// Bricks in plain z-y-x order, no padding to super-bricks (k_brute_split: every packet costs the same, order is irrelevant).
__device__ __forceinline__ GridBrick grid_lane_voxel_plain(const GridParams& g, uint32_t brick, int lane) {
  const uint32_t nby = bricks_along(g.n[1], g.bl[1]), nbz = bricks_along(g.n[2], g.bl[2]);
  const uint32_t bz = brick % nbz, t = brick / nbz, by = t % nby, bx = t / nby;
  GridBrick v;
  const uint32_t lx = g.bl[0], ly = g.bl[1], lz = g.bl[2], l = (uint32_t)lane;
  const uint32_t layers = g.xe - g.xb, xv = (bx << lx) + (l >> (ly + lz));
  v.y = (by << ly) + ((l >> lz) & ((1u << ly) - 1u));
  v.z = (bz << lz) + (l & ((1u << lz) - 1u));
  v.in_range = xv < layers && v.y < g.n[1] && v.z < g.n[2];
  v.brick_in_grid = true;
  v.bx = bx; v.by = by; v.bz = bz;
  v.x = slab_x(g, min(xv, layers - 1u));
  v.y = min(v.y, g.n[1] - 1);
  v.z = min(v.z, g.n[2] - 1);
  return v;
}

// XCD-aware work order: the dispatcher places block b on XCD b % 8.  Each XCD works through runs of 2^XCD_RUN_LOG consecutive
// packets (its private L2 keeps seeing the same part of the BVH), and the runs are dealt out round-robin — XCD x takes the runs
// x, x+8, x+16, ...: every XCD still works through whole super-bricks, and no XCD is handed the expensive eighth of the grid, as
// one contiguous eighth per XCD did (matters most for the thin multi-GPU pieces, which have few runs).
#ifndef M2S_XCD_RUN_LOG
#define M2S_XCD_RUN_LOG 8
#endif
constexpr uint32_t XCD_RUN_LOG = M2S_XCD_RUN_LOG;   // 7 is as fast, 8 re-fetches less (L2 misses 363 -> 263 MB on the headline)
__device__ __forceinline__ uint32_t xcd_remap(uint32_t b) {
  const uint32_t i = b >> 3, x = b & 7u;
  return ((((i >> XCD_RUN_LOG) << 3) + x) << XCD_RUN_LOG) | (i & ((1u << XCD_RUN_LOG) - 1u));
}

// ---- per-lane search state -----------------------------------------------------------------
template <int MODE>
struct Best {
  float d2 = __builtin_inff();      // min d2 over all triangles
  float d2pos = __builtin_inff();   // MODE_NORMAL_FOLD: min d2 over triangles with positive signed distance
  uint32_t idx = 0xffffffffu;       // MODE_NEAREST_NORMAL: triangle achieving d2 (lowest index on ties)
  bool pos = false;                 // MODE_NEAREST_NORMAL: its sign
  bool nan = false;
};

template <int MODE>
__device__ __forceinline__ void eval_triangle(Best<MODE>& best, f3 p, const TriRec& tr) {
  const f3 a = mk3(tr.ax, tr.ay, tr.az), b = mk3(tr.bx, tr.by, tr.bz), c = mk3(tr.cx, tr.cy, tr.cz);
  const TriEdges e = {mk3(tr.abx, tr.aby, tr.abz), mk3(tr.acx, tr.acy, tr.acz), mk3(tr.bcx, tr.bcy, tr.bcz)};
  const uint32_t cls = tr.cls, index = tr.index;
  if (MODE == MODE_UNSIGNED) {
    const float d2 = point_triangle_dist2(p, a, b, c, e, cls);
    best.d2 = fminf(best.d2, d2);  // f32::min drops a NaN operand (default.rs:47)
  } else {
    bool positive;
    const float d2 = point_triangle_dist2_signed_n(p, a, b, c, e, cls, mk3(tr.nrx, tr.nry, tr.nrz), &positive);
    if (MODE == MODE_NORMAL_FOLD) {
      best.nan |= !(d2 == d2);  // the reference panics: "NaN distance" (lib.rs:257)
      best.d2 = fminf(best.d2, d2);
      if (positive) best.d2pos = fminf(best.d2pos, d2);
    } else {
      if (d2 < best.d2 || (d2 == best.d2 && index < best.idx)) { best.d2 = d2; best.idx = index; best.pos = positive; }
    }
  }
}

// eval_triangle for the leaf triangles of the packet walk.  `reach` = this lane's pre-test bound reaches the triangle;
// a lane without it cannot be improved (that is what the pre-test's margin guarantees) and is left alone.  If every lane
// that is reached lies in a VERTEX region of the triangle (geo.rs:97-111 — 40 % of the evaluations on the benchmark: the
// fan of triangles around a voxel's nearest vertex, all at exactly the same distance), the closest point is that vertex
// and the edge / interior half of the computation (selects, the division, the reconstruction) is skipped for the wave.
// Same arithmetic for the lanes that count, so the result is bit-identical.
template <int MODE>
__device__ __forceinline__ void eval_triangle_leaf(Best<MODE>& best, f3 p, const TriRec& tr, bool reach) {
  if (MODE == MODE_NEAREST_NORMAL || tr.cls != TRI_REGULAR) { eval_triangle<MODE>(best, p, tr); return; }
  const f3 a = mk3(tr.ax, tr.ay, tr.az), b = mk3(tr.bx, tr.by, tr.bz), c = mk3(tr.cx, tr.cy, tr.cz);
  const f3 ab = mk3(tr.abx, tr.aby, tr.abz), ac = mk3(tr.acx, tr.acy, tr.acz);
  const RegularHead h = closest_point_regular_head(p, a, b, c, ab, ac);
  const bool vertex = h.rA | h.rB | h.rC;
  f3 q;
  bool valid = true;
  if (__ballot(reach & !vertex) == 0ull) {
    q = closest_point_regular_vertex(h, a, b, c);
    valid = vertex;                      // the other lanes are not reached: nothing to learn for them
  } else {
    q = closest_point_regular_tail(h, a, b, c, ab, ac, mk3(tr.bcx, tr.bcy, tr.bcz));
  }
  const f3 d = sub3(p, q);
  float d2 = dot3(d, d);
  if (MODE == MODE_UNSIGNED) {
    d2 = valid ? d2 : __builtin_inff();
    best.d2 = fminf(best.d2, d2);  // f32::min drops a NaN operand (default.rs:47)
  } else {   // MODE_NORMAL_FOLD
    const bool positive = dot3(d, mk3(tr.nrx, tr.nry, tr.nrz)) > 0.0f;
    best.nan |= valid & !(d2 == d2);  // the reference panics: "NaN distance" (lib.rs:257)
    d2 = valid ? d2 : __builtin_inff();
    best.d2 = fminf(best.d2, d2);
    if (positive) best.d2pos = fminf(best.d2pos, d2);
  }
}

template <int MODE>
__device__ __forceinline__ float finish(const Best<MODE>& best, bool negate_unsigned) {
  if (MODE == MODE_UNSIGNED) {
    const float d = fminf(F32_MAX_C, sqrtf(best.d2));   // fold starts from f32::MAX (default.rs:45)
    return negate_unsigned ? -d : d;
  }
  if (MODE == MODE_NORMAL_FOLD) return normal_fold_result(best.d2, best.d2pos);
  const float d = sqrtf(best.d2);
  return best.pos ? d : -d;                              // rtree.rs:118-123
}

// Pruning threshold in d2 space for the current best: a node or a triangle is looked at while its lower bound is <= (d (1 + PRUNE_REL) +
// slack)^2, with `slack` absolute (~67 ulp of the coordinate scale, + the approx_eq window of 1e-6 in Normal mode).
// What the relative part has to cover ("The pruning margin"; u = 2^-24): a triangle T may only be skipped if its
// COMPUTED distance could neither beat nor tie (Normal: approx_eq, 2 ulp) the final minimum.  With delta the true distance of T and B a
// computed lower bound of something that contains T:  B <= delta (1 + e_B) + a_B  and  computed d_T >= delta (1 - e_T) - a_T, where the
// absolute parts a_B, a_T (coordinate cancellation: <= ~16 u x scale together) are what `slack` is for, and the relative parts are
//   e_T <= 4 u     dot3 of the difference vector and the square root of the final comparison
//   e_B <= 7 u     ext_dist2 beyond 6.4 node radii (closer in, the 2e-6 x radius widening of the stored slab and the 1e-6 |v|^2 taken off the
//                  lateral term are larger than its rounding), planes_dist2 / box_dist2 likewise, + 1 u for the approximate square root here
//   2 u            the approx_eq tie window of the Normal fold (float-cmp ulps = 2)
// together < 14 u = 8.3e-7.  PRUNE_REL = 4e-6 is 4.8 times that.  (Rounds 1-4 used 2e-5: config 5's walk 78.6 -> 75.7 ms at 2e-6; far from
// a flat sheet the candidates within the margin are a disc of radius sqrt(2 m) D.)
constexpr float PRUNE_REL = 4.0e-6f;
__device__ __forceinline__ float prune_bound(float best_d2, float slack) {
  const float d = __builtin_amdgcn_sqrtf(best_d2);
  const float r = __builtin_fmaf(d, 1.0f + PRUNE_REL, slack);
  return r * r;
}

__device__ __forceinline__ float box_dist2(f3 p, float mnx, float mny, float mnz, float mxx, float mxy, float mxz) {
  const float dx = fmaxf(fmaxf(mnx - p.x, p.x - mxx), 0.0f);
  const float dy = fmaxf(fmaxf(mny - p.y, p.y - mxy), 0.0f);
  const float dz = fmaxf(fmaxf(mnz - p.z, p.z - mxz), 0.0f);
  return __builtin_fmaf(dx, dx, __builtin_fmaf(dy, dy, dz * dz));
}

// Lower bound (squared) of the distance from p to anything inside the node's disc-shaped slab
// (common.h NodeExt).  Every rounding is taken towards a SMALLER bound; FMAs are fine here.
__device__ __forceinline__ float ext_dist2(f3 p, const NodeExt& e) {
  const float vx = p.x - e.cx, vy = p.y - e.cy, vz = p.z - e.cz;
  const float t = __builtin_fmaf(e.nz, vz, __builtin_fmaf(e.ny, vy, e.nx * vx));
  const float v2 = __builtin_fmaf(vz, vz, __builtin_fmaf(vy, vy, vx * vx));
  // l^2 = v2 - t^2 cancels when p sits over the disc centre: shave a few ulps of v2 off first
  const float l2 = __builtin_fmaf(-1.0e-6f, v2, __builtin_fmaf(-t, t, v2));
  // l2 < 0 (rounding) gives sqrt = NaN and fmaxf(NaN - R, 0) = 0: still a valid lower bound
  const float lat = fmaxf(__builtin_amdgcn_sqrtf(l2) - e.R, 0.0f);
  const float s = fmaxf(fabsf(t - e.mid) - e.half, 0.0f);
  return __builtin_fmaf(s, s, lat * lat);
}

// Leaf pre-test (common.h TriPlanes): squared lower bound of the distance from p to the triangle itself.
__device__ __forceinline__ float planes_dist2(f3 p, const TriPlanes& t) {
  const float h = __builtin_fmaf(t.nz, p.z, __builtin_fmaf(t.ny, p.y, t.nx * p.x)) - t.dn;
  const float e0 = __builtin_fmaf(t.m0z, p.z, __builtin_fmaf(t.m0y, p.y, t.m0x * p.x)) - t.o0;
  const float e1 = __builtin_fmaf(t.m1z, p.z, __builtin_fmaf(t.m1y, p.y, t.m1x * p.x)) - t.o1;
  const float e2 = __builtin_fmaf(t.m2z, p.z, __builtin_fmaf(t.m2y, p.y, t.m2x * p.x)) - t.o2;
  const float e = fmaxf(fmaxf(e0, e1), fmaxf(e2, 0.0f));
  return __builtin_fmaf(h, h, e * e);
}

template <int AXIS>
__device__ __forceinline__ uint32_t stab_count(const DeviceMesh& mesh, f3 p) {
  uint32_t count = 0;
  uint32_t node = 0;
  while (node < mesh.n_nodes) {
    node = __builtin_amdgcn_readfirstlane(node);
    const NodeRec nr = record_at(mesh.nodes, node);
    const bool hit = ray_meets_box<AXIS>(p, mk3(nr.mnx, nr.mny, nr.mnz), mk3(nr.mxx, nr.mxy, nr.mxz));
    if (__ballot(hit) == 0ull) { node = nr.skip; continue; }
    if (nr.tri >= 0) {
      const uint32_t cnt = (nr.skip - node + 1u) >> 1;
      for (uint32_t k = 0; k < cnt; ++k) {
        uint32_t vi;                                   // (a wave-uniform index through a VGPR: a broadcast out of the vector L1, as record_at_vec)
        asm("v_mov_b32_e32 %0, %1" : "=v"(vi) : "s"(3u * ((uint32_t)nr.tri + k)));
        const float4 c0 = mesh.corners[vi], c1 = mesh.corners[vi + 1u], c2 = mesh.corners[vi + 2u];
        const f3 a = mk3(c0.x, c0.y, c0.z), b = mk3(c0.w, c1.x, c1.y), c = mk3(c1.z, c1.w, c2.x);
        f3 mn, mx;
        triangle_bounding_box(a, b, c, &mn, &mx);     // the candidate rule is per triangle: ITS padded box
        float t;
        const bool h = ray_meets_box<AXIS>(p, mn, mx) & ray_triangle_aligned<AXIS>(p, a, b, c, &t);
        count += h ? 1u : 0u;
      }
      node = nr.skip;
    } else {
      node = node + 1;
    }
  }
  return count;
}

// The same count with the ray-triangle tests run densely (cf. DeferQueue below): the packet's 64 sorted queries are neighbours, not
// a brick — a leaf triangle's padded box is met by a few lanes' rays, and its test (box, 18 + ~40 instructions) ran wave-wide for
// them.  Here a leaf queues (lane, triangle) for the lanes whose ray meets the leaf's box, 64 pairs are tested at a time (each lane
// one pair: the owner's point by lane permutes, the corners by a gather), and a hit bumps the owner's LDS counter.  Same candidates
// per lane (a lane whose ray misses the leaf's box cannot meet a triangle's box inside it), same test: the same count.
// `lds`: 192 words of this wave's.
template <int AXIS>
__device__ __forceinline__ uint32_t stab_count_dense(const DeviceMesh& mesh, f3 p, uint32_t* lds) {
  const uint32_t lane = threadIdx.x & 63u;
  uint32_t* q = lds;
  uint32_t* hits = lds + 128;
  uint32_t head = 0, n = 0;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  hits[lane] = 0u;
  auto flush = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const uint32_t take = min(n, 64u);
    const bool valid = lane < take;
    const uint32_t e = valid ? q[(head + lane) & 127u] : 0u;
    const uint32_t v = e & 63u, vi = 3u * (e >> 6);
    const f3 pv = mk3(__shfl(p.x, (int)v), __shfl(p.y, (int)v), __shfl(p.z, (int)v));
    const float4 c0 = mesh.corners[vi], c1 = mesh.corners[vi + 1u], c2 = mesh.corners[vi + 2u];
    const f3 a = mk3(c0.x, c0.y, c0.z), b = mk3(c0.w, c1.x, c1.y), c = mk3(c1.z, c1.w, c2.x);
    f3 mn, mx;
    triangle_bounding_box(a, b, c, &mn, &mx);     // the candidate rule is per triangle: ITS padded box
    float t;
    const bool h = ray_meets_box<AXIS>(pv, mn, mx) & ray_triangle_aligned<AXIS>(pv, a, b, c, &t);
    if (valid & h) atomicAdd(&hits[v], 1u);
    head = (head + take) & 127u;
    n -= take;
  };
  uint32_t node = 0;
  while (node < mesh.n_nodes) {
    node = __builtin_amdgcn_readfirstlane(node);
    const NodeRec nr = record_at(mesh.nodes, node);
    const bool hit = ray_meets_box<AXIS>(p, mk3(nr.mnx, nr.mny, nr.mnz), mk3(nr.mxx, nr.mxy, nr.mxz));
    const unsigned long long hb = __ballot(hit);
    if (hb == 0ull) { node = nr.skip; continue; }
    if (nr.tri >= 0) {
      const uint32_t cnt = (nr.skip - node + 1u) >> 1;
      const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(hb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)hb, 0u));
      for (uint32_t k = 0; k < cnt; ++k) {
        if (hit) q[(head + n + rank) & 127u] = lane | (((uint32_t)nr.tri + k) << 6);
        n += (uint32_t)__popcll(hb);
        if (n >= 64u) flush();
      }
      node = nr.skip;
    } else {
      node = node + 1;
    }
  }
  while (n != 0u) flush();
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  return hits[lane];
}

// ---- cut lists ------------------------------------------------------------------------------
// A third of a packet's node tests fall on nodes much larger than the packet (512^3 x blob-100k: 46 of 158 on nodes
// wider than 28 voxels, 69 on nodes wider than 14), and neighbouring packets repeat them with the same outcome.
// k_cut walks that top part of the tree ONCE per block of 2^log bricks per axis, against a bound that holds for every
// voxel of the block, and leaves at most CUT_MAX pre-order ranges [start, end) of NodeExt byte offsets per block: the
// subtrees that can still matter there.  k_packet then walks those ranges instead of starting at the root.
// A subtree is dropped only if   bound(block centre, subtree) - r_block  >  D + margins,   where D bounds the
// distance of every voxel of the block to the seed triangle of its own packet (which k_packet evaluates first):
// such a subtree cannot hold a triangle nearer than, or tied with, any voxel's final minimum.
#ifndef M2S_CUT_MAX
#define M2S_CUT_MAX 15   // 7: 12.0 ms, 15: 11.7 ms (512^3 x blob-100k)
#endif
// A list is ONE 64-byte record (round 2: 128 B of (start, end) byte offsets — 268 MB of lists for a 537 MB output):
//   word 0        number of ranges
//   word 1 + k    low S bits: start of range k (node record index), S = bits needed for the tree's node count; the other 32 - S bits:
//                 its length as a small float — 5 bits of exponent e, M = 27 - S bits of mantissa m: m << e records, the smallest
//                 such value that is >= the true length
// Lengths below 2^M records are exact and longer ones exceed the truth by less than 2^-(M-1) (100 k triangles: M = 9, 0.4 %; 1 M: M = 6,
// 3 %; the 2^25-triangle limit: M = 1): a superset of the subtrees, which a walk may always take, and one that hardly costs — a first
// form with 6-bit power-of-two lengths walked up to twice a long range: 14 % more node tests and 4.8 % more time on 512^3 x blob-1M
// (same box: 26.60 against 25.37 ms; the headline 8.40 against 8.43 ms).  A range that
// reaches into the next one is walked there twice (harmless: a minimum).  k_cut writes a word when its range closes, as before (ranges
// kept in LDS or scratch until the end and written as one record cost k_cut 20-35 %: five waves per SIMD, or scratch traffic).
__host__ __device__ __forceinline__ uint32_t cut_start_bits(uint32_t n_nodes) {
  uint32_t b = 1;
  while (b < 27u && (1u << b) < n_nodes) ++b;
  return b;
}
// length code: 5 bits of exponent e above M bits of mantissa, len = mantissa << e (explicit leading bit: no special case in the walk's
// decode, which runs once per range of every packet on the scalar unit)
__host__ __device__ __forceinline__ uint32_t cut_decode_len(uint32_t code, uint32_t M) { return (code & ((1u << M) - 1u)) << (code >> M); }
__host__ __device__ __forceinline__ uint32_t cut_encode_len(uint32_t len, uint32_t M) { // smallest representable value >= len (len >= 1)
  if (len < (1u << M)) return len;                                                      // exact, e = 0
  uint32_t e = (32u - (uint32_t)__builtin_clz(len)) - M;                                // len >> e lies in [2^(M-1), 2^M)
  uint32_t mant = (len + (1u << e) - 1u) >> e;
  if (mant == (1u << M)) { mant >>= 1; ++e; }
  return (e << M) | mant;
}
constexpr uint32_t CUT_MAX = M2S_CUT_MAX, CUT_WORDS = M2S_CUT_MAX + 1;
static_assert(M2S_CUT_MAX <= 15, "the range count has four bits");
struct CutList {
  const uint32_t* lists;   // CUT_WORDS words per block, nullptr: walk the whole tree
  uint32_t log, ny, nz;    // bricks per block per axis = 2^log; blocks along y and z
  uint32_t bx_off;         // grid: this launch covers a piece of the slab the seed lattice and the lists were built for,
                           // starting bx_off bricks into it along x (a multiple of 2^log)
  const float4* centres;   // generic queries: (centre, radius) of every packet's bounding box (k_qpacket_bounds); one list per packet
};

// Generic queries: the sorted queries [first, first + cnt) of packet k (table of launch_query_distance / k_qcells).
__device__ __forceinline__ bool query_packet_range(const uint32_t* __restrict__ table, uint32_t packet, uint32_t n_q,
                                                   uint32_t* first, uint32_t* cnt) {
  uint32_t f = packet * 64u, c = 64u;
  if (table != nullptr) {
    const uint32_t count = table[0];
    if (packet >= count) return false;
    if (table[1] == 0u) {
      f = table[2u + packet];
      c = (packet + 1u < count ? table[3u + packet] : n_q) - f;
    }
  }
  if (f >= n_q) return false;
  *first = f;
  *cnt = min(c, n_q - f);
  return true;
}
// Cell of the generic path's seed lattice that holds x (k_qlattice).
__device__ __forceinline__ uint32_t query_lattice_cell(const GridParams& L, float x, float y, float z) {
  const float q0[3] = {x, y, z};
  uint32_t cell[3];
  for (int k = 0; k < 3; ++k) {
    float f = (q0[k] - L.first[k]) / L.size[k] + 0.5f;
    f = (f == f) ? fminf(fmaxf(f, 0.0f), (float)(L.n[k] - 1)) : 0.0f;
    cell[k] = min((uint32_t)f, L.n[k] - 1);
  }
  return (cell[0] * L.n[1] + cell[1]) * L.n[2] + cell[2];
}

// ---- the packet walk ------------------------------------------------------------------------
// Wave-uniform counters of one packet's walk (SGPRs; only the M2S_STATS variant keeps them).
struct WalkStats {
  uint32_t box = 0, ext = 0, leaf = 0, pruned = 0, slab = 0, sphere = 0, pairs = 0;
  uint32_t node_lanes = 0, pre_lanes = 0;   // lanes whose bound reaches the node / that take part in a leaf's pre-test by that measure
};

// ---- split walk -------------------------------------------------------------------------------
// When the dispatcher has handed out the last packet of a launch, the wave slots fall idle one by one while the packets still
// running — the heavy ones: blob-1M's heaviest packet does 4 004 node tests + 2 220 exact evaluations and is 3.4 ms alone on a
// SIMD — decide when the launch ends.  So a walk may be SUSPENDED: from then on it keeps walking the TOP of what is left (the node
// tests there prune most of it), but a subtree of at most `emit_max` bytes of records that survives its test is not entered: it
// becomes an ITEM of a follow-up launch in which any wave may take it, starting from the bests the packet ends with (the seed is
// within 10 % of a perfect bound, so an item loses little by not seeing what the others find).  The items' minima are merged
// with atomic minima on per-voxel words (non-negative floats order like their bit patterns; min is associative and commutative:
// the same bits as one walk), and k_split_finish turns the merged minima into signed distances.  (A first form cut the unwalked
// record RANGES into eight equal pieces: a piece that starts in the middle of a subtree has lost its ancestors' pruning, seven of
// eight pieces lie where one test of an ancestor would have dropped them, and the follow-up rounds of 128^3 x blob-100k took longer
// than the walk — 46 000 items, 1.7 ms.)
//
// WHEN to suspend needs no model of the work: the launch measures itself.  The first workgroup an XCD is handed stamps the time
// (s_memrealtime, 100 MHz) into that XCD's start word, the last one into its flag ("the dispatcher has run dry here": from now on
// wave slots of this XCD fall idle).  The time between the two, divided by the rounds of the chip's wave slots that the launch is deep
// (host: packets / slots - 1, at least 1), is how long an ordinary packet of THIS launch takes at full occupancy; a walk that has done
// `grace` work units looks every CHECK_EVERY units, and once the flag has been up for `patience` such packet times it is suspended:
// every ordinary packet that was running when the flag went up has finished by then, what is left are the stragglers.  The
// follow-up rounds work through their items with a fixed set of waves striding the list; the first wave of an XCD to run out raises
// the round's flag, and an item still being walked `grace` units later is suspended in its turn (items are bounded — at most emit_max
// records — so a work count serves there).  The last round walks to the end.  (Forms that did not work: suspending everything still
// running when the flag goes up — at 128^3, four rounds deep, that is a quarter of all packets, ordinary ones, and the rounds did more
// work than the walk had left; a fixed number of work units after the flag — right for one grid, 40 % slower on the next: 96^3 wants
// 1024, 128^3 512, a 1 M-triangle slab 256; counting running packets with two atomics per packet on a per-XCD word — 45 ns each,
// serialised: 1.2 -> 7 ms.)
// Work is counted where it is done, at the leaves — 3 units per leaf visited (the node tests that led to it), 1 per
// pre-test, 4 per exact evaluation (25 : 20 : 140 vector instructions) — so the inner nodes' path carries no bookkeeping at all.
// The words are per XCD because a word everybody reads at device scope is a serial resource (one flag, one agent-scope load per
// packet: + 10 ms on the 512^3 walk — 2 M loads, ~5 ns each at the memory side).  Workgroup b of a one-dimensional launch runs on
// XCD b % 8; writer and readers share that XCD's L2, so the stores are the plain kind (the line stays in that L2) and the loads
// only have to pass the CU's own L1, which other CUs' stores never refresh: non-temporal loads (L2-served; a workgroup-scope `sc0`
// load hits the L1 like a plain one and kept seeing the flag down).  A stale word (or another dispatch order) would cost time,
// never a result.  All of it is inline asm: to the compiler these are not memory operations, so nothing around them is reordered
// or demoted for their sake (a store at the top of k_packet moved every wave-uniform load of its prologue — seed index, mesh scale —
// from the scalar to the vector unit, "may be clobbered", behind the cut list's cold miss: + 20 %).  Every string starts with
// s_nop 4: the compiler does not see into it, and an address that has just come out of a spill lane (a VALU write of an SGPR) needs
// five wait states before a memory instruction may read it — without them a follow-up round loaded its flag from garbage addresses.
constexpr uint32_t SPLIT_CHECK_EVERY = 64;
constexpr uint32_t SPLIT_CONTINUATION = 0x80000000u;   // item tag (with the slot): a suspended packet, not a subtree
struct SplitState {
  uint32_t units = 0, next_check = 0xffffffffu;   // work done so far; next look at the flag (never, unless armed)
  const uint32_t* flag_addr = nullptr;            // this XCD's flag of the launch (k_packet: the time it went up, odd; rounds: 1)
  uint32_t patience_q8 = 0;                       // k_packet: patience / rounds-before-the-flag, in 1/256
  uint32_t grace = 0;
  bool suspended = false, flag_seen = false;
  uint32_t resume_end = 0;                        // suspended: the end of the range that was being walked
  uint32_t* n_ranges = nullptr;                   // k_packet: the caller's count of ranges (zeroed on suspension: its loop ends too)
};
__device__ __forceinline__ uint32_t* split_flag_addr(const SplitCtl& ctl, uint32_t round) {   // this XCD's flag of the round
  return ctl.cnt + 16u + round * 16u + (blockIdx.x & 7u);
}
__device__ __forceinline__ uint32_t split_peek(const uint32_t* addr) {          // wave-uniform address, wave-uniform result
  uint32_t v;
  const uint32_t zero = 0;
  asm volatile("s_nop 4\n\tglobal_load_dword %0, %1, %2 nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(zero), "s"(addr));
  return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}
__device__ __forceinline__ void split_poke(uint32_t* addr, uint32_t value) {   // call with one lane active
  const uint32_t zero = 0;
  asm volatile("s_nop 4\n\tglobal_store_dword %0, %1, %2" : : "v"(zero), "v"(value), "s"(addr));
}
__device__ __forceinline__ uint32_t split_now() { return (uint32_t)__builtin_amdgcn_s_memrealtime(); }   // 10 ns ticks; differences survive the wrap
__device__ __forceinline__ void split_arm(SplitState& sp, const SplitCtl& ctl, uint32_t round, bool may_suspend) {
  if (ctl.cnt == nullptr || !may_suspend) return;
  sp.flag_addr = split_flag_addr(ctl, round);
  sp.next_check = ctl.grace;
  sp.grace = ctl.grace;
  sp.patience_q8 = ctl.patience_q8;
}
// k_packet: has this XCD's flag been up for `patience` ordinary packet times?  (the start word lies 8 words behind the flag)
__device__ __forceinline__ bool split_stragglers_only(const SplitState& sp) {
  const uint32_t up = split_peek(sp.flag_addr);
  if (up == 0u) return false;
  const uint32_t start = split_peek(sp.flag_addr + 8), now = split_now();
  const uint32_t fill = up - start;                                       // time it took to hand out this XCD's packets
  const uint32_t allowed = (uint32_t)(((unsigned long long)fill * sp.patience_q8) >> 8);
  return now - up >= allowed;
}

// A suspended PACKET (k_packet) leaves its bests and ONE item — "go on at record `off` of range `range`" — and gives up its wave slot;
// a wave of round 1 walks the top of what is left and turns the subtrees that survive into the items of round 2 (k_split_round).
// false: no accumulator slot or list entry left (the packet then walks on by itself).
template <int MODE>
__device__ __forceinline__ bool split_handover(const SplitCtl& split, uint32_t packet, uint32_t range, uint32_t off, const Best<MODE>& best, int* err) {
  const uint32_t lane = threadIdx.x & 63u;
  uint32_t slot = 0, idx = 0;
  if (lane == 0u) slot = atomicAdd(&split.cnt[0], 1u);
  slot = __builtin_amdgcn_readfirstlane(slot);
  if (slot >= split.cap_slots) return false;
  if (lane == 0u) idx = atomicAdd(&split.cnt[2], 1u);
  idx = __builtin_amdgcn_readfirstlane(idx);
  if (idx >= split.cap_items) {
    if (lane == 0u) split.slot_packet[slot] = 0xffffffffu;                  // the slot stays empty
    return false;
  }
  constexpr uint32_t AW = MODE == MODE_NORMAL_FOLD ? 128u : 64u;
  uint32_t* acc = split.acc + (size_t)slot * AW;
  acc[lane] = __float_as_uint(best.d2);
  if (MODE == MODE_NORMAL_FOLD) {
    acc[64 + lane] = __float_as_uint(best.d2pos);
    if (best.nan) atomicOr(err, ERRF_NAN);
  }
  if (lane == 0u) {
    split.items[idx] = make_uint4(packet, off, range, slot | SPLIT_CONTINUATION);
    split.slot_packet[slot] = packet;
  }
  return true;
}

// Where a suspended walk leaves the subtrees it does not enter: slots of the next round's list, reserved 64 at a time.
struct EmitState {
  uint4* list = nullptr;
  uint32_t* count = nullptr;          // the list's fill counter
  uint32_t cap = 0;
  uint32_t base = 0, used = 0, room = 0;
  uint32_t min_bytes = 0xffffffffu, max_bytes = 0;   // subtrees of min_bytes .. max_bytes of records are handed over; min = ~0: nothing is
  uint32_t packet = 0, slot = 0;
};
__device__ __forceinline__ void emit_begin(EmitState& em, const SplitCtl& ctl, uint32_t next_round, uint32_t packet, uint32_t slot) {
  em.list = ctl.items + (size_t)(next_round - 1u) * ctl.cap_items;
  em.count = ctl.cnt + 1u + next_round;
  em.cap = ctl.cap_items;
  em.min_bytes = ctl.emit_min;
  em.max_bytes = ctl.emit_max;
  em.packet = packet;
  em.slot = slot;
}
// The unused part of the reserved block becomes empty items (first == end): the list has no holes of stale data.
__device__ __forceinline__ void emit_close(EmitState& em) {
  const uint32_t i = em.base + em.used + (threadIdx.x & 63u);
  if (i < em.base + em.room && i < em.cap) em.list[i] = make_uint4(em.packet, 0u, 0u, em.slot);
  em.used = em.room;
}
// One more block of 64 slots; when the list is full, emission is switched off (the walk then enters everything itself).
__device__ __forceinline__ void emit_reserve(EmitState& em) {
  uint32_t base = 0;
  if ((threadIdx.x & 63u) == 0u) base = atomicAdd(em.count, 64u);
  em.base = __builtin_amdgcn_readfirstlane(base);
  em.used = 0;
  em.room = 64u;
  if (em.base + 64u > em.cap) {       // (what lies below the cap is this wave's to blank)
    emit_close(em);
    em.room = 0;
    em.min_bytes = 0xffffffffu;
  }
}

// ---- dense exact evaluations (DEFER) ----------------------------------------------------------
// Where a brick meets many triangles (a grid coarse against the mesh: 128^3 x blob-100k) an exact evaluation serves 7 - 18 of the
// wave's 64 lanes — the others' pre-test bound does not reach the triangle — and costs the wave its ~140 instructions all the same.
// Here the leaf only QUEUES (voxel lane, triangle) pairs for the lanes that are reached (one LDS word per pair, written at the lane's
// rank in the ballot), and whenever 64 pairs have come together every lane evaluates ONE of them: it fetches that voxel's point with
// three lane permutes and the pair's triangle record with a gather (pairs of one triangle sit in neighbouring lanes: the same cache
// line), and folds the result into the voxel's slot with an LDS minimum on the f32 bits (non-negative floats order like their
// bits; a NaN's bits lie above +inf's and never win, as fminf never takes it).  The lanes' bounds are refreshed from the slots after
// every batch: they lag by at most one batch, which costs pairs (cheap: a 64th of an evaluation each), never a result — the set of
// triangles evaluated for a voxel only grows, the minimum of the same arithmetic is the same bits.
// The slots hold what Best<MODE> holds: d2 bits; Normal fold: + d2pos bits and a NaN flag; nearest-with-normal: the 64-bit key
// (d2 bits, triangle index, !positive) whose minimum is the lexicographic rule of rtree.rs:118-123 (as the lane walk's LaneShare).
template <int MODE>
struct DeferLayout {
  static constexpr uint32_t SLOT_WORDS = MODE == MODE_NORMAL_FOLD ? 192u : MODE == MODE_NEAREST_NORMAL ? 128u : 64u;
  static constexpr uint32_t DWORDS = (256u + SLOT_WORDS) / 2u;     // the LDS block of one wave, in 8-byte words: two rings + the slots
};
// DEFER = 2: a triangle that DEFER_DIRECT_LANES or more lanes reach is evaluated wave-wide at once, as without the queue.  For grids
// much finer than the mesh — 1024^3 over a flat 100 k-triangle sheet, 0.006 triangles per brick: far from the surface ONE triangle
// is the nearest for most of a brick, 47 of 64 lanes per evaluation — the queue has nothing to compact (walk, direct / queued /
// both: sheet-100k 1024^3 69.0 / 71.9 / 65.4 ms, blob-100k 1024^3 35.6 / 34.0 / 33.6).  Its own variant, because the second
// evaluation body costs the dense regime 3 - 6 % by being there (128^3 x blob-100k 1.02 -> 1.09 ms, 256^3 1.74 -> 1.80).
constexpr uint32_t DEFER_DIRECT_LANES = 48;
struct DeferQueue {
  uint32_t* q;          // LDS: 128 pair words (ring), lane | triangle slot << 6: pairs that passed the pre-test
  uint32_t* slot;       // LDS: the running results of the 64 voxels (DeferLayout)
  uint32_t head, n;     // wave-uniform
  uint32_t* q1;         // LDS: 128 pair words (ring): pairs whose lane's bound reaches the LEAF, waiting for the pre-test (DEFER = 3)
  uint32_t head1, n1;
};
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
template <int MODE>
__device__ __forceinline__ unsigned long long defer_key(const Best<MODE>& x) {
  return ((unsigned long long)__float_as_uint(x.d2) << 32) | ((unsigned long long)(x.idx & 0x7fffffffu) << 1) | (x.pos ? 0ull : 1ull);
}
template <int MODE>
__device__ __forceinline__ DeferQueue defer_begin(unsigned long long* lds) {
  const uint32_t lane = threadIdx.x & 63u;
  DeferQueue dq = {reinterpret_cast<uint32_t*>(lds), reinterpret_cast<uint32_t*>(lds) + 256, 0u, 0u, reinterpret_cast<uint32_t*>(lds) + 128, 0u, 0u};
  if (MODE == MODE_NEAREST_NORMAL) reinterpret_cast<unsigned long long*>(dq.slot)[lane] = 0x7f800000ffffffffull;   // (+inf, none)
  else {
    dq.slot[lane] = 0x7f800000u;
    if (MODE == MODE_NORMAL_FOLD) { dq.slot[64u + lane] = 0x7f800000u; dq.slot[128u + lane] = 0u; }
  }
  return dq;
}
// Evaluates up to 64 queued pairs (all of them when fewer are left) and refreshes the lanes' results from their slots.
template <int MODE>
__device__ __forceinline__ void defer_flush(const DeviceMesh& mesh, f3 p, DeferQueue& dq, Best<MODE>& best) {
  const uint32_t lane = threadIdx.x & 63u;
  wave_lds_sync();
  const uint32_t take = min(dq.n, 64u);
  const bool valid = lane < take;
  const uint32_t e = valid ? dq.q[(dq.head + lane) & 127u] : 0u;
  const uint32_t v = e & 63u, t = e >> 6;
  const f3 pv = mk3(__shfl(p.x, (int)v), __shfl(p.y, (int)v), __shfl(p.z, (int)v));
  Best<MODE> one;
  eval_triangle<MODE>(one, pv, mesh.tris[t]);
  if (valid) {
    if (MODE == MODE_NEAREST_NORMAL) atomicMin(&reinterpret_cast<unsigned long long*>(dq.slot)[v], defer_key<MODE>(one));
    else {
      atomicMin(&dq.slot[v], __float_as_uint(one.d2));
      if (MODE == MODE_NORMAL_FOLD) {
        atomicMin(&dq.slot[64u + v], __float_as_uint(one.d2pos));
        if (one.nan) dq.slot[128u + v] = 1u;
      }
    }
  }
  dq.head = (dq.head + take) & 127u;
  dq.n -= take;
  wave_lds_sync();
  if (MODE == MODE_NEAREST_NORMAL) {
    const unsigned long long kk = reinterpret_cast<const unsigned long long*>(dq.slot)[lane];
    if (kk < defer_key<MODE>(best)) { best.d2 = __uint_as_float((uint32_t)(kk >> 32)); best.idx = (uint32_t)(kk >> 1) & 0x7fffffffu; best.pos = (kk & 1ull) == 0ull; }
  } else {
    best.d2 = fminf(best.d2, __uint_as_float(dq.slot[lane]));
    if (MODE == MODE_NORMAL_FOLD) { best.d2pos = fminf(best.d2pos, __uint_as_float(dq.slot[64u + lane])); best.nan |= dq.slot[128u + lane] != 0u; }
  }
}
// DEFER = 3: the leaf pre-test run densely too.  It costs the wave its 20 instructions (and a 64-byte scalar load) per leaf triangle
// for the 11 (128^3 x blob-100k) ... 30 (headline) of 64 lanes whose bound reaches the leaf at all — the node test has just said
// which.  Those lanes are queued per leaf triangle (first ring), 64 such pairs are pre-tested at a time — the owner's point and bound
// by lane permutes, the planes by a gather — and the ones that pass move on to the second ring, the exact evaluations' (above).
// Returns true if that ring was flushed (the lanes' bounds have moved).
template <int MODE>
__device__ __forceinline__ bool defer_pretest(const DeviceMesh& mesh, f3 p, float thr, DeferQueue& dq, Best<MODE>& best) {
  const uint32_t lane = threadIdx.x & 63u;
  wave_lds_sync();
  const uint32_t take = min(dq.n1, 64u);
  const bool valid = lane < take;
  const uint32_t e = valid ? dq.q1[(dq.head1 + lane) & 127u] : 0u;
  const uint32_t v = e & 63u, t = e >> 6;
  const f3 pv = mk3(__shfl(p.x, (int)v), __shfl(p.y, (int)v), __shfl(p.z, (int)v));
  const float tv = __shfl(thr, (int)v);
  const bool pass = valid & !(planes_dist2(pv, mesh.planes[t]) > tv);
  dq.head1 = (dq.head1 + take) & 127u;
  dq.n1 -= take;
  const unsigned long long rb = __ballot(pass);
  if (rb == 0ull) return false;
  const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(rb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)rb, 0u));
  if (pass) dq.q[(dq.head + dq.n + rank) & 127u] = e;
  dq.n += (uint32_t)__popcll(rb);
  if (dq.n < 64u) return false;
  defer_flush<MODE>(mesh, p, dq, best);
  return true;
}
template <int MODE>
__device__ __forceinline__ void defer_drain(const DeviceMesh& mesh, f3 p, float slack, DeferQueue& dq, Best<MODE>& best) {
  while (dq.n1 != 0u) defer_pretest<MODE>(mesh, p, prune_bound(best.d2, slack), dq, best);
  while (dq.n != 0u) defer_flush<MODE>(mesh, p, dq, best);
}

// (Round 6, measured and not kept — two cursors per wave, so that two record loads are in flight and two node tests issue back to back.  First
// form: idle cursors take the next range of the list or the rest of the other cursor's range; ~100 scalar instructions of bookkeeping per
// iteration; headline walk 7.1 -> 12.6 ms.  Second form: two ranges in lockstep while both last, the same instruction count per node plus ~6
// scalar instructions; 6.47 -> 6.86 ms, 512^3 x sheet-100k Normal 13.3 -> 14.1.  The record loads hit the scalar cache: what a wave waits for
// is its own instruction stream, and at eight waves per SIMD a wave's time follows its instruction count.  profiles/r06_two_cursors_*.)
// The pre-order records [off, end) of the oriented-bound tree for the 64 points of a wave: position wave-uniform (SGPR), node
// records and pre-test planes through scalar loads, a subtree left when no lane's bound reaches it.  BUDGET: the walk may stop
// early (sp.suspended, off = the first record not yet looked at).  EMIT: a suspended walk — surviving subtrees of em.min_bytes ..
// em.max_bytes are written to the next round's list instead of being entered.
template <int MODE, bool STATS, bool BUDGET, bool EMIT = false, bool HANDOVER = false, int DEFER = 0>
__device__ __forceinline__ void walk_span(const DeviceMesh& mesh, f3 p, float slack, Best<MODE>& best, float& thr, uint32_t& off,
                                          uint32_t end, WalkStats& st, SplitState& sp, EmitState* emp = nullptr, DeferQueue* dqp = nullptr) {
  // The walk addresses NodeExt by BYTE offset (its skip links are stored that way): the scalar loads then take
  // the offset operand directly and the loop carries no address arithmetic.
  constexpr uint32_t NB = (uint32_t)sizeof(NodeExt);
  while (off < end) {
    off = __builtin_amdgcn_readfirstlane(off);
    const NodeExt nr = record_at_bytes<NodeExt>(mesh.ext, off);
    if (STATS) ++st.box;
    const float ed2 = ext_dist2(p, nr);
    if (STATS) st.node_lanes += (uint32_t)__popcll(__ballot(!(ed2 > thr)));
    if (STATS && __ballot(!(ed2 > thr)) == 0ull) {
      ++st.pruned;
      const float vx = p.x - nr.cx, vy = p.y - nr.cy, vz = p.z - nr.cz;
      const float t = nr.nz * vz + nr.ny * vy + nr.nx * vx, sl = fmaxf(fabsf(t - nr.mid) - nr.half, 0.0f);
      if (__ballot(!(sl * sl > thr)) == 0ull) ++st.slab;
      const float rs = __builtin_amdgcn_sqrtf(nr.R * nr.R + (fabsf(nr.mid) + nr.half) * (fabsf(nr.mid) + nr.half));
      const float sq = fmaxf(__builtin_amdgcn_sqrtf(vx * vx + vy * vy + vz * vz) - rs, 0.0f);
      if (__ballot(!(sq * sq > thr)) == 0ull) ++st.sphere;
    }
    if (__ballot(!(ed2 > thr)) == 0ull) { off = nr.skip; continue; }   // a NaN bound keeps the node
    if (nr.tri >= 0) {
      const uint32_t cnt = (nr.skip - off + NB) / (2u * NB);    // triangles of this (possibly collapsed) leaf
      if (DEFER == 3) {
        DeferQueue& dq = *dqp;
        const bool want = !(ed2 > thr);
        const unsigned long long wb = __ballot(want);
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(wb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)wb, 0u));
        const uint32_t wn = (uint32_t)__popcll(wb);
        for (uint32_t k = 0; k < cnt; ++k) {
          if (want) dq.q1[(dq.head1 + dq.n1 + rank) & 127u] = (threadIdx.x & 63u) | (((uint32_t)nr.tri + k) << 6);
          dq.n1 += wn;
          if (dq.n1 >= 64u) {
            if (BUDGET) sp.units += 4u;
            if (defer_pretest<MODE>(mesh, p, thr, dq, best)) thr = prune_bound(best.d2, slack);
          }
        }
      } else
      for (uint32_t k = 0; k < cnt; ++k) {
        if (STATS) { ++st.ext; st.pre_lanes += (uint32_t)__popcll(__ballot(!(ed2 > thr))); }
        const TriPlanes tp = record_at(mesh.planes, (uint32_t)nr.tri + k);   // scalar: small, needed for every leaf triangle
        const bool reach = !(planes_dist2(p, tp) > thr);
        const unsigned long long rb = __ballot(reach);
        if (rb != 0ull) {   // some lane's bound reaches the triangle itself
          if (STATS) { ++st.leaf; st.pairs += (uint32_t)__popcll(rb); }
          if (DEFER != 0 && !(DEFER == 2 && (uint32_t)__popcll(rb) >= DEFER_DIRECT_LANES)) {
            DeferQueue& dq = *dqp;
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(rb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)rb, 0u));
            if (reach) dq.q[(dq.head + dq.n + rank) & 127u] = (threadIdx.x & 63u) | (((uint32_t)nr.tri + k) << 6);
            dq.n += (uint32_t)__popcll(rb);
            if (dq.n >= 64u) {
              if (BUDGET) sp.units += 4u;
              defer_flush<MODE>(mesh, p, dq, best);
              thr = prune_bound(best.d2, slack);
            }
          } else {
            if (BUDGET) sp.units += 4u;
            const TriRec tr = record_at_vec(mesh.tris, (uint32_t)nr.tri + k);
            eval_triangle_leaf<MODE>(best, p, tr, reach);
            thr = prune_bound(best.d2, slack);
          }
        }
      }
      off = nr.skip;
      if (BUDGET) {
        sp.units += 3u + cnt;
        if (sp.units >= sp.next_check) {
          // (no exit of its own: a second way out of this loop cost the walk 11 % although it was never taken; the loop ends by its
          // own condition)
          bool go;
          if (HANDOVER) go = split_stragglers_only(sp);                   // k_packet: measured patience
          else {                                                           // a follow-up round: `grace` units after the flag was first seen up
            go = sp.flag_seen;
            if (!go && split_peek(sp.flag_addr) != 0u) { sp.flag_seen = true; sp.next_check = sp.units + sp.grace; }
          }
          if (go) {
            // Suspended.  No exit of its own and nothing but three scalar moves here: the loop ends by its condition (and so does the
            // caller's loop over the ranges, whose count is taken away).  Every other form tried in k_packet — an early return, a
            // hand-over in this branch that ends the wave, a retry loop around the call, a rewound range counter — cost the 512^3
            // walk 10 ... 40 % although none of it was ever executed there.
            if (off < end || HANDOVER) { sp.suspended = true; sp.resume_end = end; end = off; if (HANDOVER) *sp.n_ranges = 0u; }
            sp.next_check = 0xffffffffu;
          } else if (HANDOVER || !sp.flag_seen) {
            sp.next_check = sp.units + SPLIT_CHECK_EVERY;
          }
        }
      }
    } else {
      if (EMIT) {
        EmitState& em = *emp;
        const uint32_t bytes = nr.skip - off;
        if (bytes >= em.min_bytes && bytes <= em.max_bytes) {
          if (em.used == em.room) emit_reserve(em);
          if (em.used < em.room) {
            if ((threadIdx.x & 63u) == 0u) em.list[em.base + em.used] = make_uint4(em.packet, off, nr.skip, em.slot);
            ++em.used;
            off = nr.skip;                                      // somebody else's from here
            continue;
          }
        }
      }
      off = off + NB;
    }
  }
}

// The value of one voxel / query, stored the way the call's delivery asks for (plain, peer stores, trailing push).
__device__ __forceinline__ void store_grid_result(float* __restrict__ out, size_t out_index, float result, bool store, const GridParams& g,
                                                  const GridBrick& vox, const PeerOut& peers, int lane) {
  if (peers.progress != nullptr) {
    // M2S_PEER_TRAIL: the copy kernel that trails this walk runs on other XCDs, whose L2s are separate.  The values are
    // stored write-through at device scope (no L2 write-back fence: a release fence per wave — buffer_wbl2 — made the walk ten
    // times slower), the wave waits until the store has been acknowledged, and only then counts the packet.
    if (store) __hip_atomic_store(&out[out_index], result, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // one counter per (unit, brick row): thousands of device-scope atomics on ONE address serialise at the memory side
    // and the packet that completes a row counts the row on the unit's own counter, the only address the copy kernel polls
    if (lane == 0) {
      const uint32_t unit = vox.bx >> peers.unit_log;
      const uint32_t nbx = bricks_along(g.xe - g.xb, g.bl[0]), nbz = bricks_along(g.n[2], g.bl[2]);
      const uint32_t bricks = min((unit + 1u) << peers.unit_log, nbx) - (unit << peers.unit_log);
      const uint32_t old = __hip_atomic_fetch_add(&peers.progress[peers.units + unit * peers.rows + vox.by], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (old + 1u == bricks * nbz) __hip_atomic_fetch_add(&peers.progress[unit], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return;
  }
  if (store) out[out_index] = result;
  // M2S_PEER_STORE: the same value into every peer's whole-grid buffer (indices are whole-grid there)
  if (peers.n != 0u && store) {
    const size_t gi = out_index + (size_t)g.out_off;
    for (uint32_t i = 0; i < peers.n; ++i) peers.p[i][gi] = result;
  }
}

// ---- k_packet -------------------------------------------------------------------------------
// `seed_in` (one TriRec slot per 2^seed_shift bricks per axis, may be null) replaces the greedy descent:
// the packet starts from a triangle near its own centre (jump-flooding seed pass below).
// (eight waves per SIMD: the split variant's bookkeeping would otherwise take the kernel to 106 SGPRs — seven waves, - 12 %; the
// compiler parks what does not fit in spare VGPR lanes)
template <bool GRID, int MODE, int SIGN, bool STATS, bool SPLIT, int DEFER = 0>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_packet(DeviceMesh mesh, GridParams g, const float4* __restrict__ qsorted,
                                               const uint32_t* __restrict__ perm, uint32_t n_q,
                                               const uint32_t* __restrict__ plane, float* __restrict__ out,
                                               int* __restrict__ err, uint32_t n_packets,
                                               const uint32_t* __restrict__ seed_in, uint32_t seed_shift,
                                               uint32_t seed_ny, uint32_t seed_nz,
                                               const GridParams* __restrict__ seed_lattice, CutList cut, PeerOut peers, SplitCtl split) {
  const int lane = threadIdx.x & 63;
  // the first and the last workgroup an XCD is handed stamp the time: start of the launch there; from here on its wave slots fall idle
  if (SPLIT && (blockIdx.x < 8u || blockIdx.x + 8u >= gridDim.x) && lane == 0) {
    const uint32_t t = split.idle_below /* forced */ ? 0u : split_now();
    if (blockIdx.x < 8u) split_poke(split_flag_addr(split, 0u) + 8, t);
    if (blockIdx.x + 8u >= gridDim.x) split_poke(split_flag_addr(split, 0u), t | 1u);
  }
  const uint32_t block = xcd_remap(blockIdx.x);
  // wave-uniform, and said so: the brick decode below (two divisions by multiplication, shifts, bounds) then runs on the scalar unit
  const uint32_t packet = (uint32_t)__builtin_amdgcn_readfirstlane((int)block);   // one packet per single-wave workgroup: the slot is free as soon as the walk ends (4 waves per group: +3.8 %)
  if (packet >= n_packets) return;

  f3 p;
  size_t out_index;
  bool store;
  GridBrick vox{};
  if (GRID) {
    vox = grid_lane_voxel(g, packet, lane);
    if (!vox.brick_in_grid) return;      // padding of the super-brick order
    p = grid_point(g, vox);
    out_index = ((size_t)vox.x * g.n[1] + vox.y) * g.n[2] + vox.z - (size_t)g.out_off;
    store = vox.in_range;
  } else {
    // generic queries: `plane` carries the packet table of launch_query_distance (k_qcells): [0] packets, [1] mode, [2 + k] the
    // first sorted query of packet k.  mode 1 (more packets than the launch has waves): 64 consecutive queries per packet.
    uint32_t first, cnt;
    if (!query_packet_range(plane, packet, n_q, &first, &cnt)) return;
    const uint32_t i = first + min((uint32_t)lane, cnt - 1u);
    const float4 q = qsorted[i];
    p = mk3(q.x, q.y, q.z);
    out_index = perm[i];
    store = (uint32_t)lane < cnt;
  }

  Best<MODE> best;
  WalkStats st;
  uint32_t st_ranges = 0, st_band = 0, st_rbytes = 0;
  if (mesh.n_nodes) {
    const float scale = fmaxf(mesh_scale(mesh), fmaxf(fabsf(p.x), fmaxf(fabsf(p.y), fabsf(p.z))));
    const float slack = 4.0e-6f * scale + (MODE == MODE_NORMAL_FOLD ? 2.5e-6f : 0.0f);
    SplitState sp;
    if (SPLIT) split_arm(sp, split, 0u, true);
    __shared__ unsigned long long defer_lds[DEFER ? DeferLayout<MODE>::DWORDS : 1];
    DeferQueue dq = {nullptr, nullptr, 0u, 0u};
    if (DEFER) dq = defer_begin<MODE>(defer_lds);

    // pre-order ranges to walk: the brick's cut list (grid path), or the whole tree.  The list is 64 bytes that nobody has
    // touched before (written by k_cut, read once): it is requested here, in front of the seed evaluation, so that the ~1 us of
    // the miss passes under its 120 instructions instead of in front of the walk.
    constexpr uint32_t NB = (uint32_t)sizeof(NodeExt);
    const uint32_t* cl = nullptr;
    uint32_t n_ranges = 1, cut_off_v = 0, cut_end_v = (lane == 1) ? mesh.n_nodes * NB : 0u;   // no list: lane 1 holds the whole tree
    if (SPLIT) sp.n_ranges = &n_ranges;
    if (cut.lists != nullptr) {
      const uint32_t cb = GRID ? __builtin_amdgcn_readfirstlane((((vox.bx + cut.bx_off) >> cut.log) * cut.ny + (vox.by >> cut.log)) * cut.nz + (vox.bz >> cut.log))
                               : packet;
      cl = cut.lists + (size_t)cb * CUT_WORDS;
      // The whole 64-byte record with ONE vector load — lane l takes word l — requested here, in front of the seed evaluation;
      // lanes 1..15 then decode their range side by side (eight VALU instructions for the whole list), and the range loop below
      // fetches (start, end) of range k from lane 1 + k with two v_readlane: no load and no scalar arithmetic per range.  (Decoding
      // on the scalar unit, one range at a time, cost the Normal-sign walk of 1024^3 1.3 %: 9 SALU x ~10 ranges per packet.)
      const uint32_t cw = cl[(uint32_t)lane & 15u];
      const uint32_t cS = cut_start_bits(mesh.n_nodes), cfirst = cw & ((1u << cS) - 1u);
      const uint32_t clen = ((cw >> cS) & ((1u << (27u - cS)) - 1u)) << (cw >> 27);
      cut_off_v = cfirst * NB;
      cut_end_v = min(cfirst + clen, mesh.n_nodes) * NB;
      n_ranges = __builtin_amdgcn_readfirstlane(cw);
    }
    if (seed_in != nullptr) {
      // seed: a triangle near this packet's centre, from the seed pass
      uint32_t sidx = packet;
      if (GRID) {  // 2^seed_shift bricks per axis share one seed point
        sidx = (((vox.bx + cut.bx_off) >> seed_shift) * seed_ny + (vox.by >> seed_shift)) * seed_nz + (vox.bz >> seed_shift);
      } else {     // generic queries: the lattice cell that holds the packet's centre (its first point without the packet boxes)
        const GridParams L = *seed_lattice;
        float q0[3] = {__shfl(p.x, 0), __shfl(p.y, 0), __shfl(p.z, 0)};
        if (cut.centres != nullptr) {      // the same cell k_cut<false> took this packet's seed from
          const float4 c = cut.centres[packet];
          q0[0] = c.x; q0[1] = c.y; q0[2] = c.z;
        }
        sidx = __builtin_amdgcn_readfirstlane(query_lattice_cell(L, q0[0], q0[1], q0[2]));
      }
      // one seed per packet: wave-uniform, so the 96-byte record comes through scalar loads
      const uint32_t slot = __builtin_amdgcn_readfirstlane(min(seed_in[sidx], mesh.n_tris - 1));
      const TriRec tr = record_at(mesh.tris, slot);
      eval_triangle<MODE>(best, p, tr);
    } else {
      // seed: greedy descent towards the packet's first point, evaluate that leaf for every lane
      const f3 c = {__shfl(p.x, 0), __shfl(p.y, 0), __shfl(p.z, 0)};
      uint32_t n = 0;
      NodeRec nr = mesh.nodes[0];
      while (nr.tri < 0) {
        const uint32_t l = n + 1;
        const NodeRec nl = mesh.nodes[l];
        const uint32_t r = nl.skip;
        const NodeRec nrr = mesh.nodes[r];
        const float dl = box_dist2(c, nl.mnx, nl.mny, nl.mnz, nl.mxx, nl.mxy, nl.mxz);
        const float dr = box_dist2(c, nrr.mnx, nrr.mny, nrr.mnz, nrr.mxx, nrr.mxy, nrr.mxz);
        const bool go_left = __builtin_amdgcn_readfirstlane((int)(dl <= dr)) != 0;
        n = go_left ? l : r;
        nr = go_left ? nl : nrr;
      }
      const TriRec tr = mesh.tris[nr.tri];
      eval_triangle<MODE>(best, p, tr);
    }

    float thr = prune_bound(best.d2, slack);
    if (STATS && GRID) {
      const float cells = __shfl(__builtin_amdgcn_sqrtf(best.d2), 0) / fabsf(g.size[0]);
      st_band = cells < 1.0f ? 0u : min(7u, 1u + (uint32_t)__builtin_amdgcn_readfirstlane((int)floorf(log2f(cells))));
      st_band = __builtin_amdgcn_readfirstlane(st_band);
    }
    // M2S_STATS=2: a second, counting-only traversal that starts from the final bound ("perfect seed")
    const int passes = (STATS && mesh.stats != nullptr && mesh.stats[7] == 2ull) ? 2 : 1;
    for (int pass = 0; pass < passes; ++pass) {
      if (pass == 1) { st.box = 0; st.ext = 0; st.leaf = 0; }
      if (STATS) st_ranges = n_ranges;
      uint32_t range = 0, off = 0;
      for (; range < n_ranges; ++range) {
        // (a rounded-up range may reach into the next one: those records are then walked twice, which changes no minimum)
        off = (uint32_t)__builtin_amdgcn_readlane((int)cut_off_v, (int)(1u + range));
        const uint32_t end = (uint32_t)__builtin_amdgcn_readlane((int)cut_end_v, (int)(1u + range));
        if (STATS) st_rbytes += end - off;
        walk_span<MODE, STATS, SPLIT, false, SPLIT, DEFER>(mesh, p, slack, best, thr, off, end, st, sp, nullptr, &dq);
      }
      if (DEFER) defer_drain<MODE>(mesh, p, slack, dq, best);   // what is still queued (a suspended packet hands over complete minima)
      if (SPLIT && sp.suspended) {
        // (range has been stepped once more by the loop's increment)
        if (!split_handover<MODE>(split, packet, range - 1u, off, best, err)) atomicOr(err, ERRF_SPLIT_OVERFLOW);   // cannot happen: a slot per packet
        return;                                                                // k_split_finish writes this packet's voxels
      }
    }
  }

  if (STATS && mesh.stats != nullptr && lane == 0) {
    atomicAdd(&mesh.stats[0], (unsigned long long)st.box);
    atomicAdd(&mesh.stats[1], (unsigned long long)st.ext);
    atomicAdd(&mesh.stats[2], (unsigned long long)st.leaf);
    atomicAdd(&mesh.stats[3], 1ull);
    atomicAdd(&mesh.stats[4], (unsigned long long)st.pruned);
    atomicAdd(&mesh.stats[5], (unsigned long long)st.slab);
    atomicAdd(&mesh.stats[6], (unsigned long long)st.sphere);
    atomicMax(&mesh.stats[72], (unsigned long long)st.box);
    atomicMax(&mesh.stats[73], (unsigned long long)st.leaf);
    atomicMax(&mesh.stats[74], (unsigned long long)(st.box + st.ext + 4u * st.leaf));
    atomicAdd(&mesh.stats[75], (unsigned long long)st.node_lanes);
    atomicAdd(&mesh.stats[76], (unsigned long long)st.pre_lanes);
    atomicAdd(&mesh.stats[77], (unsigned long long)st.pairs);
    unsigned long long* q = mesh.stats + 8 + 8 * st_band;
    atomicAdd(&q[0], (unsigned long long)st.box);
    atomicAdd(&q[1], (unsigned long long)st.ext);
    atomicAdd(&q[2], (unsigned long long)st.leaf);
    atomicAdd(&q[3], 1ull);
    atomicAdd(&q[4], (unsigned long long)st_ranges);
    atomicAdd(&q[5], (unsigned long long)st.pairs);
    atomicAdd(&q[6], (unsigned long long)st_rbytes);
    atomicAdd(&q[7], (unsigned long long)(st_rbytes <= 4096u ? 1u : 0u));
    // histogram of the packets' work units (node tests + pre-tests + 4 x exact evaluations) in octaves: stats[80 + log2]
    const uint32_t units = st.box + st.ext + 4u * st.leaf;
    atomicAdd(&mesh.stats[80 + (31 - __builtin_clz(units | 1u))], 1ull);
  }

  bool negate = false;
  if (MODE == MODE_UNSIGNED) {
    if (SIGN == SIGN_GRID_PLANE) {
      const size_t w = ((size_t)vox.x * g.n[1] + vox.y) * g.nzw + (vox.z >> 5);
      negate = (plane[w] >> (vox.z & 31u)) & 1u;                       // grid.rs:630-636
    } else if (SIGN == SIGN_RAYS3) {
      // best of three (bvh.rs:131-141, rtree_bvh.rs:161-171): where the +X and +Y parities agree they ARE the majority, so the +Z
      // walk only runs for a packet in which some lane's first two rays disagree (never, for a watertight mesh, but for rays
      // through an edge; 10 M queries x blob-100k: a third of the 2 ms of stabbing).  The vote's result is the same.
      __shared__ uint32_t ray_lds[DEFER != 0 ? 192 : 1];
      const uint32_t cx = DEFER != 0 ? stab_count_dense<0>(mesh, p, ray_lds) : stab_count<0>(mesh, p);
      const uint32_t cy = DEFER != 0 ? stab_count_dense<1>(mesh, p, ray_lds) : stab_count<1>(mesh, p);
      uint32_t cz = cx;
      if (__ballot(((cx ^ cy) & 1u) != 0u) != 0ull) cz = DEFER != 0 ? stab_count_dense<2>(mesh, p, ray_lds) : stab_count<2>(mesh, p);
      negate = ((cx & 1u) + (cy & 1u) + (cz & 1u)) > 1u;
    }
  }
  if (MODE == MODE_NORMAL_FOLD && best.nan) atomicOr(err, ERRF_NAN);
  const float result = finish<MODE>(best, negate);
  if (GRID) store_grid_result(out, out_index, result, store, g, vox, peers, lane);
  else if (store) out[out_index] = result;
}

// ---- packet groups: several waves per packet (round 6) ------------------------------------------------------------------------
// A launch shallower than the chip (64^3 ... 128^3 over 100 k triangles: 4 096 ... 32 768 packets on 8 192 wave slots, triangles much finer
// than the bricks) lasts as long as its slowest packet's chain of dependent loads — 0.58 - 0.64 ms whatever the grid (round 5) — while
// most wave slots idle.  Here a packet is a workgroup of GROUP_WAVES = 2 or 4 waves: every wave evaluates the packet's seed, then walks its share
// of the tree — the subtrees w, w + GROUP_WAVES, ... of the TOP_SUBTREES subtrees TOP_LOG levels below the root (k_tree_top; neighbours
// in Morton order go to different waves, so the part of the mesh next to the brick is dealt out evenly) — with its own queues but ONE
// set of per-voxel minima in LDS: the queued evaluations fold into them with atomic minima (as they always did) and every wave refreshes
// its bounds from them after each batch, so a wave prunes with what the others have found.  A minimum is a minimum: the same bits as
// one wave walking everything (parity suite in this form too).  Grid calls without cut lists and without the split walk, leaf work fully
// queued (DEFER = 3).  Walk, one wave / group (profiles/r06_groups.txt): blob-100k 32^3 1.22 -> 0.46 ms, 48^3 0.87 -> 0.42, 64^3 0.66 -> 0.50, 80^3 0.56 ->
// 0.49, 96^3 0.62 -> 0.55; blob-11k 32^3 0.30 -> 0.12, 64^3 0.23 -> 0.14.  More waves do not keep helping — every wave starts from the seed's bound
// alone and learns what the others found only batch by batch, so the group's total work grows: 16 waves at 32^3 0.56 ms, 8 at 64^3 0.59 — and
// from ~110^3 on (four rounds of the chip's wave slots) one wave per packet is faster again.
constexpr uint32_t GROUP_MAX_WAVES = 4, TOP_LOG = 8, TOP_SUBTREES = 1u << TOP_LOG;   // 2 or 4 waves per packet, chosen by the launch (blockDim.x / 64)
__global__ __launch_bounds__(TOP_SUBTREES) void k_tree_top(DeviceMesh mesh, uint2* __restrict__ top) {
  constexpr uint32_t NB = (uint32_t)sizeof(NodeExt);
  const uint32_t sub = threadIdx.x;
  uint32_t off = 0, end = mesh.n_nodes * NB;
  for (uint32_t lv = 0; lv < TOP_LOG && off < end; ++lv) {
    const NodeExt* nr = reinterpret_cast<const NodeExt*>(reinterpret_cast<const char*>(mesh.ext) + off);
    const uint32_t rest = sub & ((1u << (TOP_LOG - lv)) - 1u);
    if (nr->tri >= 0) { if (rest != 0u) end = off; break; }      // a leaf on the way belongs to the subtree whose remaining bits are zero
    const uint32_t skip = nr->skip, left = off + NB;
    const uint32_t right = reinterpret_cast<const NodeExt*>(reinterpret_cast<const char*>(mesh.ext) + left)->skip;
    if ((sub >> (TOP_LOG - 1u - lv)) & 1u) { off = right; end = skip; } else { off = left; end = right; }
  }
  top[sub] = make_uint2(off, end);
}
template <int MODE, int SIGN>
__global__ __launch_bounds__(64 * GROUP_MAX_WAVES) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_packet_group(
    DeviceMesh mesh, GridParams g, const uint32_t* __restrict__ plane, float* __restrict__ out, int* __restrict__ err, uint32_t n_packets,
    const uint32_t* __restrict__ seed_in, uint32_t seed_shift, uint32_t seed_ny, uint32_t seed_nz, uint32_t bx_off, const uint2* __restrict__ top, PeerOut peers) {
  static_assert(MODE == MODE_UNSIGNED || MODE == MODE_NORMAL_FOLD, "grid modes");
  const int lane = threadIdx.x & 63;
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const uint32_t packet = (uint32_t)__builtin_amdgcn_readfirstlane((int)xcd_remap(blockIdx.x));
  if (packet >= n_packets) return;
  const GridBrick vox = grid_lane_voxel(g, packet, lane);
  if (!vox.brick_in_grid) return;        // padding of the super-brick order (the same for every wave of the group)
  const f3 p = grid_point(g, vox);
  const size_t out_index = ((size_t)vox.x * g.n[1] + vox.y) * g.n[2] + vox.z - (size_t)g.out_off;

  extern __shared__ unsigned long long group_lds[];   // blockDim.x / 64 x 256 ring words, then DeferLayout<MODE>::SLOT_WORDS slot words
  const uint32_t GROUP_WAVES = blockDim.x >> 6;
  uint32_t* const lds = reinterpret_cast<uint32_t*>(group_lds);
  DeferQueue dq = {lds + wave * 256u, lds + GROUP_WAVES * 256u, 0u, 0u, lds + wave * 256u + 128u, 0u, 0u};   // own rings, shared minima
  if (wave == 0u) {
    dq.slot[lane] = 0x7f800000u;
    if (MODE == MODE_NORMAL_FOLD) { dq.slot[64 + lane] = 0x7f800000u; dq.slot[128 + lane] = 0u; }
  }
  Best<MODE> best;
  if (mesh.n_nodes) {
    const float scale = fmaxf(mesh_scale(mesh), fmaxf(fabsf(p.x), fmaxf(fabsf(p.y), fabsf(p.z))));
    const float slack = 4.0e-6f * scale + (MODE == MODE_NORMAL_FOLD ? 2.5e-6f : 0.0f);
    uint32_t slot = 0;
    if (seed_in != nullptr) {
      const uint32_t sidx = (((vox.bx + bx_off) >> seed_shift) * seed_ny + (vox.by >> seed_shift)) * seed_nz + (vox.bz >> seed_shift);
      slot = __builtin_amdgcn_readfirstlane(min(seed_in[sidx], mesh.n_tris - 1));
    }
    eval_triangle<MODE>(best, p, record_at(mesh.tris, slot));
    float thr = prune_bound(best.d2, slack);
    __syncthreads();                     // the shared minima are initialised
    WalkStats st;
    SplitState sp;
    for (uint32_t k = wave; k < TOP_SUBTREES; k += GROUP_WAVES) {
      const uint2 r = top[k];            // wave-uniform
      uint32_t off = (uint32_t)__builtin_amdgcn_readfirstlane((int)r.x);
      const uint32_t end = (uint32_t)__builtin_amdgcn_readfirstlane((int)r.y);
      if (off >= end) continue;
      walk_span<MODE, false, false, false, false, 3>(mesh, p, slack, best, thr, off, end, st, sp, nullptr, &dq);
    }
    defer_drain<MODE>(mesh, p, slack, dq, best);
    // this wave's minima (the seed's among them) join the group's
    atomicMin(&dq.slot[lane], __float_as_uint(best.d2));
    if (MODE == MODE_NORMAL_FOLD) {
      atomicMin(&dq.slot[64 + lane], __float_as_uint(best.d2pos));
      if (best.nan) dq.slot[128 + lane] = 1u;
    }
    __syncthreads();
    if (wave != 0u) return;
    best.d2 = __uint_as_float(dq.slot[lane]);
    if (MODE == MODE_NORMAL_FOLD) { best.d2pos = __uint_as_float(dq.slot[64 + lane]); best.nan = dq.slot[128 + lane] != 0u; }
  } else if (wave != 0u) return;

  bool negate = false;
  if (MODE == MODE_UNSIGNED && SIGN == SIGN_GRID_PLANE) {
    const size_t w = ((size_t)vox.x * g.n[1] + vox.y) * g.nzw + (vox.z >> 5);
    negate = (plane[w] >> (vox.z & 31u)) & 1u;                       // grid.rs:630-636
  }
  if (MODE == MODE_NORMAL_FOLD && best.nan) atomicOr(err, ERRF_NAN);
  store_grid_result(out, out_index, finish<MODE>(best, negate), vox.in_range, g, vox, peers, lane);
}

// One follow-up round of the split walk (grid path): a fixed set of single-wave workgroups strides the round's list.  An item is either
// a CONTINUATION — a suspended packet: the rest of its range `range` from record `first` on, and the ranges behind it (round 1) — or a
// subtree [first, end) that a suspended walk did not enter.  The wave rebuilds the packet's 64 points, starts from the slot's current
// minima (plain loads: a stale value is merely a looser bound), walks, and folds what it found into the slot.  A continuation is walked
// in emit mode from the start; a subtree may be suspended in its turn (except in the last round, `final`) and is then finished in emit
// mode: what is not entered goes to the next round's list.
template <int MODE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_split_round(DeviceMesh mesh, GridParams g, SplitCtl split, CutList cut, uint32_t round, bool final, int* __restrict__ err) {
  constexpr uint32_t AW = MODE == MODE_NORMAL_FOLD ? 128u : 64u;
  constexpr uint32_t NB = (uint32_t)sizeof(NodeExt);
  const int lane = threadIdx.x & 63;
  const uint32_t n_items = min(split.cnt[1u + round], split.cap_items);
  const uint4* list = split.items + (size_t)(round - 1u) * split.cap_items;
  for (uint32_t i = blockIdx.x; i < n_items; i += gridDim.x) {
    const uint4 it = list[i];
    const uint32_t packet = __builtin_amdgcn_readfirstlane(it.x), third = __builtin_amdgcn_readfirstlane(it.z), tag = __builtin_amdgcn_readfirstlane(it.w);
    const uint32_t slot = tag & ~SPLIT_CONTINUATION;
    const bool continuation = (tag & SPLIT_CONTINUATION) != 0u;
    uint32_t off = __builtin_amdgcn_readfirstlane(it.y);
    if (!continuation && off >= third) continue;                           // an empty item (the unused part of a reserved block)
    const GridBrick vox = grid_lane_voxel(g, packet, lane);
    const f3 p = grid_point(g, vox);
    const float scale = fmaxf(mesh_scale(mesh), fmaxf(fabsf(p.x), fmaxf(fabsf(p.y), fabsf(p.z))));
    const float slack = 4.0e-6f * scale + (MODE == MODE_NORMAL_FOLD ? 2.5e-6f : 0.0f);
    uint32_t* acc = split.acc + (size_t)slot * AW;
    Best<MODE> best;
    const uint32_t d2_in = acc[lane];
    uint32_t d2pos_in = 0x7f800000u;
    best.d2 = __uint_as_float(d2_in);
    if (MODE == MODE_NORMAL_FOLD) { d2pos_in = acc[64 + lane]; best.d2pos = __uint_as_float(d2pos_in); }
    float thr = prune_bound(best.d2, slack);
    WalkStats st;
    SplitState idle;
    EmitState em;
    __shared__ unsigned long long defer_lds[DeferLayout<MODE>::DWORDS];
    wave_lds_sync();                                                         // (the previous item's slots have been read)
    DeferQueue dq = defer_begin<MODE>(defer_lds);
    if (continuation) {
      // the packet's ranges again (k_packet's decode), from range `third` on
      uint32_t n_ranges = 1, cut_off_v = 0, cut_end_v = (lane == 1) ? mesh.n_nodes * NB : 0u;
      if (cut.lists != nullptr) {
        const uint32_t cb = __builtin_amdgcn_readfirstlane((((vox.bx + cut.bx_off) >> cut.log) * cut.ny + (vox.by >> cut.log)) * cut.nz + (vox.bz >> cut.log));
        const uint32_t cw = cut.lists[(size_t)cb * CUT_WORDS + ((uint32_t)lane & 15u)];
        const uint32_t cS = cut_start_bits(mesh.n_nodes), cfirst = cw & ((1u << cS) - 1u);
        const uint32_t clen = ((cw >> cS) & ((1u << (27u - cS)) - 1u)) << (cw >> 27);
        cut_off_v = cfirst * NB;
        cut_end_v = min(cfirst + clen, mesh.n_nodes) * NB;
        n_ranges = __builtin_amdgcn_readfirstlane(cw);
      }
      if (!final) emit_begin(em, split, round + 1u, packet, slot);           // (a one-round configuration walks everything here)
      for (uint32_t range = third; range < n_ranges; ++range) {
        if (range != third) off = (uint32_t)__builtin_amdgcn_readlane((int)cut_off_v, (int)(1u + range));
        const uint32_t end = (uint32_t)__builtin_amdgcn_readlane((int)cut_end_v, (int)(1u + range));
        walk_span<MODE, false, false, true, false, 3>(mesh, p, slack, best, thr, off, end, st, idle, &em, &dq);
      }
      emit_close(em);
    } else {
      SplitState sp;
      split_arm(sp, split, round, !final);
      walk_span<MODE, false, true, false, false, 3>(mesh, p, slack, best, thr, off, third, st, sp, nullptr, &dq);
      if (sp.suspended) {
        emit_begin(em, split, round + 1u, packet, slot);
        walk_span<MODE, false, false, true, false, 3>(mesh, p, slack, best, thr, off, sp.resume_end, st, idle, &em, &dq);
        emit_close(em);
      }
    }
    defer_drain<MODE>(mesh, p, slack, dq, best);
    // NaN never enters a minimum (fminf drops it), so the words stay ordered like non-negative floats
    if (__float_as_uint(best.d2) < d2_in) atomicMin(&acc[lane], __float_as_uint(best.d2));
    if (MODE == MODE_NORMAL_FOLD) {
      if (__float_as_uint(best.d2pos) < d2pos_in) atomicMin(&acc[64 + lane], __float_as_uint(best.d2pos));
      if (best.nan) atomicOr(err, ERRF_NAN);
    }
  }
  // out of items: from here on this wave's slot is idle (many waves write the same word: harmless)
  if (!final && lane == 0) split_poke(split_flag_addr(split, round), 1u);
}

// The voxels of the suspended packets, from their merged minima.
template <int MODE, int SIGN>
__global__ __launch_bounds__(256) void k_split_finish(GridParams g, const uint32_t* __restrict__ plane, float* __restrict__ out, SplitCtl split, PeerOut peers) {
  constexpr uint32_t AW = MODE == MODE_NORMAL_FOLD ? 128u : 64u;
  const int lane = threadIdx.x & 63;
  const uint32_t n_slots = min(split.cnt[0], split.cap_slots);
  for (uint32_t slot = blockIdx.x * 4u + (threadIdx.x >> 6); slot < n_slots; slot += gridDim.x * 4u) {
    const uint32_t packet = split.slot_packet[slot];
    if (packet == 0xffffffffu) continue;                                   // walked to the end by its own wave after all
    const GridBrick vox = grid_lane_voxel(g, packet, lane);
    const uint32_t* acc = split.acc + (size_t)slot * AW;
    Best<MODE> best;
    best.d2 = __uint_as_float(acc[lane]);
    if (MODE == MODE_NORMAL_FOLD) best.d2pos = __uint_as_float(acc[64 + lane]);
    bool negate = false;
    if (MODE == MODE_UNSIGNED && SIGN == SIGN_GRID_PLANE) {
      const size_t w = ((size_t)vox.x * g.n[1] + vox.y) * g.nzw + (vox.z >> 5);
      negate = (plane[w] >> (vox.z & 31u)) & 1u;
    }
    const float result = finish<MODE>(best, negate);
    const size_t out_index = ((size_t)vox.x * g.n[1] + vox.y) * g.n[2] + vox.z - (size_t)g.out_off;
    store_grid_result(out, out_index, result, vox.in_range, g, vox, peers, lane);
  }
}
// Clears the counters and flags of a split walk; `forced`: the flags start raised (every walk is suspended at its first check).
__global__ void k_split_init(uint32_t* __restrict__ cnt, uint32_t forced) {
  for (uint32_t i = threadIdx.x; i < SPLIT_CNT_WORDS; i += blockDim.x) cnt[i] = i >= 16u ? forced : 0u;   // forced: every flag starts raised
}

// M2S_PEER_TRAIL: pushes the slab to the peers unit by unit while the walk is still running.  Unit u = the x-layers of
// 2^unit_log bricks; it is complete when progress[u] has reached the number of packets that lie in it.  Every workgroup waits
// for the unit (one lane polls, s_sleep between polls), then copies its share of it with 16 B per lane to every peer.
// A walk that never finishes (a fault on its stream) would leave this kernel spinning: after ~2 s without progress it
// raises ERRF_TRAIL_TIMEOUT and leaves.
__global__ __launch_bounds__(256) void k_push_trailing(const float* __restrict__ src, PeerOut peers, uint64_t slab_first, uint64_t row_cells,
                                                       uint32_t layers, uint32_t layers_per_unit, uint32_t n_units, int* __restrict__ err) {
  __shared__ int ok;
  for (uint32_t u = 0; u < n_units; ++u) {
    if (threadIdx.x == 0) {
      int good = 1;
      uint32_t spins = 0;
      while (__hip_atomic_load(&peers.progress[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < peers.rows) {   // all brick rows of the unit
        __builtin_amdgcn_s_sleep(127);
        if (++spins > 10000000u) { good = 0; atomicOr(err, ERRF_TRAIL_TIMEOUT); break; }
      }
      ok = good;
    }
    __syncthreads();
    if (!ok) return;
    const uint32_t x0 = u * layers_per_unit, x1 = min(layers, x0 + layers_per_unit);
    const uint64_t first = slab_first + (uint64_t)x0 * row_cells, count = (uint64_t)(x1 - x0) * row_cells;
    // row_cells * 4 B and the slab start need not be 16-byte multiples: head / body / tail as in k_push_cells
    const uint64_t head = min(count, (uint64_t)((4u - (uint32_t)(first & 3u)) & 3u));
    const uint64_t n4 = (count - head) >> 2, tail0 = head + (n4 << 2);
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (uint64_t)gridDim.x * blockDim.x;
    // the values were written by waves on other XCDs: read them past this XCD's L2 (device-scope loads; they only exist in 32 bits)
    const float* s1 = src + first + head;
    for (uint64_t i = tid; i < n4; i += stride) {
      float4 v;
      v.x = __hip_atomic_load(s1 + 4 * i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      v.y = __hip_atomic_load(s1 + 4 * i + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      v.z = __hip_atomic_load(s1 + 4 * i + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      v.w = __hip_atomic_load(s1 + 4 * i + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      for (uint32_t k = 0; k < peers.n; ++k) reinterpret_cast<float4*>(peers.p[k] + first + head)[i] = v;
    }
    if (tid < head) {
      const float v = __hip_atomic_load(src + first + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      for (uint32_t k = 0; k < peers.n; ++k) peers.p[k][first + tid] = v;
    }
    if (tid < count - tail0) {
      const float v = __hip_atomic_load(src + first + tail0 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      for (uint32_t k = 0; k < peers.n; ++k) peers.p[k][first + tail0 + tid] = v;
    }
    __syncthreads();                                         // `ok` is rewritten for the next unit
  }
}

// M2S_PEER_PUSH: one slab piece of the finished whole-grid buffer to every peer, 16 B per lane.
__global__ __launch_bounds__(256) void k_push_cells(const float* __restrict__ src, PeerOut peers, uint64_t first, uint64_t count) {
  // head: up to the next 16-byte boundary; body: float4; tail: the rest
  const uint64_t head = min(count, (uint64_t)((4u - (uint32_t)(first & 3u)) & 3u));
  const uint64_t n4 = (count - head) >> 2, tail0 = head + (n4 << 2);
  const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (uint64_t)gridDim.x * blockDim.x;
  const float4* s4 = reinterpret_cast<const float4*>(src + first + head);
  for (uint64_t i = tid; i < n4; i += stride) {
    const float4 v = s4[i];
    for (uint32_t k = 0; k < peers.n; ++k) reinterpret_cast<float4*>(peers.p[k] + first + head)[i] = v;
  }
  if (tid < head) {
    const float v = src[first + tid];
    for (uint32_t k = 0; k < peers.n; ++k) peers.p[k][first + tid] = v;
  }
  if (tid < count - tail0) {
    const float v = src[first + tail0 + tid];
    for (uint32_t k = 0; k < peers.n; ++k) peers.p[k][first + tail0 + tid] = v;
  }
}

// The lane walks gather per-lane records; the compiler's own choice (104 VGPRs, 4 waves per SIMD) hides less of that latency than six
// waves with 96 bytes of scratch do: 128^3 x blob-1M 8.9 -> 7.7 ms, 1 M queries 2.75 -> 2.63 ms, the small cases unchanged.
#ifndef M2S_LANE_WAVES
#define M2S_LANE_WAVES 6
#endif
// ---- the lane walk's tree traversal -------------------------------------------------------------------------
// walk_range: the pre-order records [off, limit) of the oriented-bound tree, one lane on its own, at most max_steps node tests.
// ANY slot may start a range: every slot holds a valid record (k_emit / k_node_ext write one per node, the descendants of a
// collapsed leaf included), a range that starts inside a subtree simply meets that subtree's nodes without their ancestors'
// pruning, and every leaf of the range is either met or skipped with a pruned ancestor that lies in the range itself.
template <int MODE>
__device__ __forceinline__ void walk_range(const DeviceMesh& mesh, f3 p, float slack, Best<MODE>& best, float& thr, uint32_t& off,
                                           uint32_t limit, uint32_t max_steps, uint32_t& st_nodes, uint32_t& st_exact) {
  constexpr uint32_t NB = (uint32_t)sizeof(NodeExt);
  const char* ext_bytes = reinterpret_cast<const char*>(mesh.ext);
  for (uint32_t s = 0; s < max_steps && off < limit; ++s) {
    const NodeExt nr = *reinterpret_cast<const NodeExt*>(ext_bytes + off);
    ++st_nodes;
    if (ext_dist2(p, nr) > thr) { off = nr.skip; continue; }
    if (nr.tri >= 0) {
      const uint32_t cnt = (nr.skip - off + NB) / (2u * NB);
      for (uint32_t k = 0; k < cnt; ++k) {
        if (!(planes_dist2(p, mesh.planes[nr.tri + k]) > thr)) {
          eval_triangle<MODE>(best, p, mesh.tris[nr.tri + k]);
          thr = prune_bound(best.d2, slack);
          ++st_exact;
        }
      }
      off = nr.skip;
    } else {
      off += NB;
    }
  }
}
// lane_tree_walk: every lane walks the whole tree for its own point — ROUND node tests at a time — and lanes that run out of work
// take over a share of their neighbours'.  A lane near the centre of curvature of a dimple meets hundreds of tied triangles (1 300
// node tests against 160 for the average lane of a 16^3 grid) and a small launch lasts as long as its longest chain of dependent
// loads; the wave as a whole has 64 x 160 tests to do.  After every round the unfinished ranges are published in LDS and ALL lanes
// are dealt out over them again (range k goes to the lanes with lane % K == k, cut into equal pieces): a lane then walks a piece of
// another lane's range for that lane's point, starting from that lane's current bound, and folds what it finds into the owner's
// slot with LDS atomic minima.  Any slot can start a range (walk_range).  Same triangles or more, same arithmetic per triangle:
// bit-identical.
struct LaneShare {
  unsigned long long key[64];   // MODE_NEAREST_NORMAL: (d2 bits, index, !positive): the lexicographic minimum rtree.rs:118-123 asks for
  uint32_t a[64], b[64];        // d2 bits (and d2pos bits for the Normal fold); non-negative floats order like their bit patterns
  uint32_t flag[64];            // Normal fold: a NaN distance was met
  uint32_t rng[64][3];          // unfinished ranges of this round: first, end, owner
  float4 pt[64];                // every lane's point and slack
};
template <int MODE>
__device__ __forceinline__ void lane_tree_walk(const DeviceMesh& mesh, f3 p, float slack, Best<MODE>& best, bool valid,
                                               uint32_t& st_nodes, uint32_t& st_exact, LaneShare& sh) {
  constexpr uint32_t NB = (uint32_t)sizeof(NodeExt);
  constexpr uint32_t ROUND = 64;
  const uint32_t end = mesh.n_nodes * NB;
  const uint32_t lane = threadIdx.x & 63u;
  auto key_of = [](const Best<MODE>& x) {
    return ((unsigned long long)__float_as_uint(x.d2) << 32) | ((unsigned long long)(x.idx & 0x7fffffffu) << 1) | (x.pos ? 0ull : 1ull);
  };
  sh.pt[lane] = make_float4(p.x, p.y, p.z, slack);
  sh.a[lane] = __float_as_uint(best.d2);
  sh.b[lane] = __float_as_uint(best.d2pos);
  sh.flag[lane] = best.nan ? 1u : 0u;
  if (MODE == MODE_NEAREST_NORMAL) sh.key[lane] = key_of(best);
  uint32_t owner = lane, off = valid ? 0u : end, limit = end;
  f3 tp = p;
  float tsl = slack;
  Best<MODE> b = best;
  float thr = prune_bound(b.d2, tsl);
  for (;;) {
    walk_range<MODE>(mesh, tp, tsl, b, thr, off, limit, ROUND, st_nodes, st_exact);
    const unsigned long long unf = __ballot(off < limit);
    if (unf == ~0ull) continue;                             // nobody is idle
    // fold what this lane has found into its owner's slot
    if (MODE == MODE_NEAREST_NORMAL) atomicMin(&sh.key[owner], key_of(b));
    else {
      atomicMin(&sh.a[owner], __float_as_uint(b.d2));
      if (MODE == MODE_NORMAL_FOLD) { atomicMin(&sh.b[owner], __float_as_uint(b.d2pos)); if (b.nan) atomicOr(&sh.flag[owner], 1u); }
    }
    if (unf == 0ull) break;
    const uint32_t K = (uint32_t)__popcll(unf);
    if (off < limit) {
      const uint32_t r = (uint32_t)__popcll(unf & ((1ull << lane) - 1ull));
      sh.rng[r][0] = off; sh.rng[r][1] = limit; sh.rng[r][2] = owner;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const uint32_t k = lane % K, idx = lane / K, helpers = (64u - k + K - 1u) / K;
    const uint32_t so = sh.rng[k][0], se = sh.rng[k][1], sowner = sh.rng[k][2];
    const uint32_t records = (se - so) / NB, piece = (records + helpers - 1u) / helpers;
    off = min(se, so + idx * piece * NB);
    limit = min(se, off + piece * NB);
    owner = sowner;
    const float4 q = sh.pt[owner];
    tp = mk3(q.x, q.y, q.z);
    tsl = q.w;
    // start from the owner's current result: the tightest bound anybody has for that point
    b = Best<MODE>();
    if (MODE == MODE_NEAREST_NORMAL) {
      const unsigned long long kk = sh.key[owner];
      b.d2 = __uint_as_float((uint32_t)(kk >> 32)); b.idx = (uint32_t)(kk >> 1) & 0x7fffffffu; b.pos = (kk & 1ull) == 0ull;
      if (b.idx == 0x7fffffffu) b.idx = 0xffffffffu;        // "none yet" survives the 31-bit trip
    } else {
      b.d2 = __uint_as_float(sh.a[owner]);
      if (MODE == MODE_NORMAL_FOLD) b.d2pos = __uint_as_float(sh.b[owner]);
    }
    thr = prune_bound(b.d2, tsl);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();                        // the ranges are read: the next round may overwrite them
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  if (MODE == MODE_NEAREST_NORMAL) {
    const unsigned long long kk = sh.key[lane];
    best.d2 = __uint_as_float((uint32_t)(kk >> 32)); best.idx = (uint32_t)(kk >> 1) & 0x7fffffffu; best.pos = (kk & 1ull) == 0ull;
  } else {
    best.d2 = __uint_as_float(sh.a[lane]);
    if (MODE == MODE_NORMAL_FOLD) { best.d2pos = __uint_as_float(sh.b[lane]); best.nan = sh.flag[lane] != 0u; }
  }
}

// Per-lane greedy descent (towards the child box nearer to p) and evaluation of the leaf it ends in: a second starting candidate for
// the lane walks.  Their lattice seed is only as good as the lattice is fine — one point per 4^3 voxels of a 16^3 grid is 64 seeds
// for the whole box — and a lane that starts with a loose bound walks long: the launch lasts as long as its slowest lane.
template <int MODE>
__device__ __forceinline__ void greedy_leaf(const DeviceMesh& mesh, f3 p, Best<MODE>& best) {
  uint32_t n = 0;
  NodeRec nr = mesh.nodes[0];
  while (nr.tri < 0) {
    const uint32_t l = n + 1;
    const NodeRec nl = mesh.nodes[l];
    const uint32_t r = nl.skip;
    const NodeRec nrr = mesh.nodes[r];
    const float dl = box_dist2(p, nl.mnx, nl.mny, nl.mnz, nl.mxx, nl.mxy, nl.mxz);
    const float dr = box_dist2(p, nrr.mnx, nrr.mny, nrr.mnz, nrr.mxx, nrr.mxy, nrr.mxz);
    const bool go_left = dl <= dr;
    n = go_left ? l : r;
    nr = go_left ? nl : nrr;
  }
  const uint32_t cnt = (nr.skip - n + 1u) >> 1;
  for (uint32_t k = 0; k < cnt; ++k) eval_triangle<MODE>(best, p, mesh.tris[(uint32_t)nr.tri + k]);
}

// ---- k_lane ---------------------------------------------------------------------------------
// One VOXEL per lane, every lane walking the tree on its own (per-lane offsets, records by vector gathers).
// For the opposite regime of k_packet: when the triangles are much smaller than the voxels (a 1 M-triangle
// scan into a 128^3 grid), the 64 voxels of a brick each need a different handful of triangles; the packet walk
// then runs every exact evaluation wave-wide for the benefit of one lane (blob-1M in 128^3: 11.8 ms), while
// independent walks only pay for what each voxel needs.  Same bounds, same leaf pre-test, same arithmetic, same
// per-brick seed; the result is the exact minimum either way.
template <int MODE, int SIGN>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(M2S_LANE_WAVES, 8))) void k_lane(DeviceMesh mesh, GridParams g, const uint32_t* __restrict__ plane,
                                              float* __restrict__ out, int* __restrict__ err, uint32_t n_packets,
                                              const uint32_t* __restrict__ seed_in, uint32_t seed_shift, uint32_t seed_ny, uint32_t seed_nz,
                                              uint32_t bx_off, PeerOut peers) {
  const int lane = threadIdx.x & 63;
  const uint32_t packet = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (packet >= n_packets) return;
  const GridBrick vox = grid_lane_voxel(g, packet, lane);
  if (!vox.brick_in_grid) return;
  const f3 p = grid_point(g, vox);
  const size_t out_index = ((size_t)vox.x * g.n[1] + vox.y) * g.n[2] + vox.z - (size_t)g.out_off;
  const bool LANE_VALID = vox.in_range;                // lanes beyond the grid's edge hold a clamped copy: nothing to walk for them

  Best<MODE> best;
  if (mesh.n_nodes) {
    const float scale = fmaxf(mesh_scale(mesh), fmaxf(fabsf(p.x), fmaxf(fabsf(p.y), fabsf(p.z))));
    const float slack = 4.0e-6f * scale + (MODE == MODE_NORMAL_FOLD ? 2.5e-6f : 0.0f);
    uint32_t slot = 0;
    if (seed_in != nullptr)
      slot = min(seed_in[(((vox.bx + bx_off) >> seed_shift) * seed_ny + (vox.by >> seed_shift)) * seed_nz + (vox.bz >> seed_shift)], mesh.n_tris - 1);
    eval_triangle<MODE>(best, p, mesh.tris[slot]);
    greedy_leaf<MODE>(mesh, p, best);
    uint32_t st_nodes = 0, st_exact = 0;               // M2S_STATS
    __shared__ LaneShare lane_share[4];                // one per wave of the workgroup
    lane_tree_walk<MODE>(mesh, p, slack, best, LANE_VALID, st_nodes, st_exact, lane_share[threadIdx.x >> 6]);
    if (mesh.stats != nullptr && LANE_VALID) {         // per lane: the lane walk's unit is the lane
      atomicAdd(&mesh.stats[0], (unsigned long long)st_nodes);
      atomicAdd(&mesh.stats[2], (unsigned long long)st_exact);
      atomicAdd(&mesh.stats[3], 1ull);
      atomicMax(&mesh.stats[72], (unsigned long long)st_nodes);
      atomicMax(&mesh.stats[73], (unsigned long long)st_exact);
    }
  }
  bool negate = false;
  if (MODE == MODE_UNSIGNED && SIGN == SIGN_GRID_PLANE) {
    const size_t w = ((size_t)vox.x * g.n[1] + vox.y) * g.nzw + (vox.z >> 5);
    negate = (plane[w] >> (vox.z & 31u)) & 1u;                         // grid.rs:630-636
  }
  if (MODE == MODE_NORMAL_FOLD && best.nan) atomicOr(err, ERRF_NAN);
  const float result = finish<MODE>(best, negate);
  if (vox.in_range) {
    out[out_index] = result;
    for (uint32_t i = 0; i < peers.n; ++i) peers.p[i][out_index + (size_t)g.out_off] = result;
  }
}

// ---- jump-flooding seed pass ------------------------------------------------------------------
// Seeds only have to be GOOD, never exact (they bound the first prune, nothing else), so the seed
// lattice (one point per 4^3 brick) is filled by jump flooding (Rong & Tan 2006) over triangle
// CENTROIDS instead of a second exact tree walk: fully data parallel, cost proportional to the
// lattice (no long-running waves), ~10 ops per candidate.  k_jfa_splat drops every triangle into
// the lattice cell of its centroid (clamped, so triangles outside an x-slab still enter at the
// border); each k_jfa_pass lets a cell adopt the best candidate of its 26 neighbours at +-step.
__device__ __forceinline__ f3 lattice_point(const GridParams& g, uint32_t x, uint32_t y, uint32_t z) {
  // x is numbered along the virtual slab (interleaved chunks laid end to end); g.xb of a lattice is 0
  return {cell_center(g.first[0], g.size[0], slab_x(g, x)), cell_center(g.first[1], g.size[1], y), cell_center(g.first[2], g.size[2], z)};
}
// Lattice index along x (virtual numbering) of the point nearest to real position index `ir` (lattice units from the
// slab's first point): inside another rank's chunks it is the nearer end of the neighbouring own chunk.
__device__ __forceinline__ uint32_t lattice_x_from_real(const GridParams& g, float fr) {
  if (g.chunk_log >= 31u) return min((uint32_t)fminf(fmaxf(fr, 0.0f), (float)(g.n[0] - 1u)), g.n[0] - 1u);
  const float C = (float)(1u << g.chunk_log), P = (float)g.period;
  fr = fmaxf(fr, 0.0f);
  float j = floorf(fr / P), o = fr - j * P;                         // period, offset inside it
  if (o >= C) {                                                      // between two own chunks
    if (o - C < P - o) o = C - 1.0f;                                 // nearer to the end of chunk j
    else { j += 1.0f; o = 0.0f; }                                    // nearer to the start of chunk j + 1
  }
  const float v = j * C + o;
  return min((uint32_t)fminf(v, (float)(g.n[0] - 1u)), g.n[0] - 1u);
}

// `gp` (device pointer) overrides `g0` when the lattice is only known on the device (generic queries).
__global__ __launch_bounds__(256) void k_jfa_splat(DeviceMesh mesh, GridParams g0, const GridParams* __restrict__ gp,
                                                   unsigned long long* __restrict__ keys) {
  const GridParams g = gp ? *gp : g0;
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= mesh.n_tris) return;
  const float4 c = mesh.cen[t];
  const float cc[3] = {c.x, c.y, c.z};
  uint32_t cell[3];
  for (int k = 0; k < 3; ++k) {
    float f = (cc[k] - g.first[k]) / g.size[k] + 0.5f;
    if (!(f == f)) return;                                   // NaN centroid: not a useful seed
    if (k == 0) { cell[0] = lattice_x_from_real(g, f); continue; }
    f = fminf(fmaxf(f, 0.0f), (float)(g.n[k] - 1));
    cell[k] = min((uint32_t)f, g.n[k] - 1);
  }
  // several triangles land in one cell (all those clamped onto a border cell in particular): keep the one
  // whose centroid is nearest to the cell centre — 64-bit min over (distance bits, slot)
  const f3 p = lattice_point(g, cell[0], cell[1], cell[2]);
  const float ex = p.x - c.x, ey = p.y - c.y, ez = p.z - c.z;
  const float d2 = __builtin_fmaf(ex, ex, __builtin_fmaf(ey, ey, ez * ez));
  if (!(d2 == d2)) return;
  const unsigned long long key = ((unsigned long long)__float_as_uint(d2) << 32) | t;
  atomicMin(&keys[((size_t)cell[0] * g.n[1] + cell[1]) * g.n[2] + cell[2]], key);
}

// The lattice carries the candidate's centroid beside its id (xyz, id bits in w): a pass then reads 27 neighbouring 16-byte
// records — structured, cache-friendly reads — instead of 27 ids plus a dependent random gather of each candidate's centroid.
__global__ __launch_bounds__(256) void k_jfa_load(DeviceMesh mesh, const unsigned long long* __restrict__ keys, size_t n,
                                                  float4* __restrict__ lat) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t id = (uint32_t)(keys[i] & 0xffffffffull);    // untouched cells hold ~0: id 0xffffffff = none
  float4 c = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  if (id != 0xffffffffu) c = mesh.cen[id];
  c.w = __uint_as_float(id);
  lat[i] = c;
}

__global__ __launch_bounds__(256) void k_jfa_pass(GridParams g0, const GridParams* __restrict__ gp, const float4* __restrict__ in,
                                                  float4* __restrict__ out, int step, uint32_t* __restrict__ ids_out) {
  const GridParams g = gp ? *gp : g0;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)g.n[0] * g.n[1] * g.n[2];
  if (i >= total) return;
  const int z = (int)(i % g.n[2]), y = (int)((i / g.n[2]) % g.n[1]), x = (int)(i / ((size_t)g.n[2] * g.n[1]));
  const f3 p = lattice_point(g, (uint32_t)x, (uint32_t)y, (uint32_t)z);
  uint32_t best = 0xffffffffu;
  float4 bc = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(0xffffffffu));
  float bd = __builtin_inff();
  // Nine records per x-plane are requested TOGETHER (clamped addresses, validity as a flag) and only then compared.  Written with a
  // `continue` per out-of-range or empty neighbour the loop was a chain of 27 dependent memory round trips per point — 56 us per pass of
  // the 512^3 call's lattice, 400 us for 1024^3, whatever the cache hit rate (an LDS-tiled form was no faster for the same reason).
  const int n0 = (int)g.n[0], n1 = (int)g.n[1], n2 = (int)g.n[2];
#pragma unroll
  for (int dx = -1; dx <= 1; ++dx) {
    const int xx = x + dx * step;
    const bool okx = xx >= 0 && xx < n0;
    const size_t xbase = (size_t)min(max(xx, 0), n0 - 1) * (size_t)n1;
    float4 c[9];
    bool ok[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const int yy = y + (k / 3 - 1) * step, zz = z + (k % 3 - 1) * step;
      ok[k] = okx && yy >= 0 && yy < n1 && zz >= 0 && zz < n2;
      c[k] = in[(xbase + (size_t)min(max(yy, 0), n1 - 1)) * (size_t)n2 + (size_t)min(max(zz, 0), n2 - 1)];
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const uint32_t cand = __float_as_uint(c[k].w);
      const float ex = p.x - c[k].x, ey = p.y - c[k].y, ez = p.z - c[k].z;
      const float d = __builtin_fmaf(ex, ex, __builtin_fmaf(ey, ey, ez * ez));
      const bool take = ok[k] && cand != 0xffffffffu && (d < bd || (d == bd && cand < best));
      bd = take ? d : bd;
      best = take ? cand : best;
      bc.x = take ? c[k].x : bc.x;
      bc.y = take ? c[k].y : bc.y;
      bc.z = take ? c[k].z : bc.z;
      bc.w = take ? c[k].w : bc.w;
    }
  }
  out[i] = bc;
  if (ids_out) ids_out[i] = best;
}

// The pass for lattices of fewer than 2^31 points, written for the VALU: the kernel above spends ~900 vector instructions per point —
// 64-bit index arithmetic per neighbour, three compares and six selects per candidate — and is bound by exactly that (an LDS-tiled form
// and one with all 27 loads in flight took the same 56 us per pass of the 512^3 call's lattice, 390 us for 1024^3).  Here a neighbour's
// index is the point's own 32-bit index plus a wave-uniform offset, (distance bits, id) is one 64-bit key so that "nearer, or as near
// with the smaller id" is one unsigned compare, only the key and the neighbour's number are carried (the winner's record is fetched
// again at the end), and out-of-range neighbours are bits of a precomputed mask.  Same candidates, same tie rule: the same seeds.
__global__ __launch_bounds__(256) void k_jfa_pass32(GridParams g, const float4* __restrict__ in, float4* __restrict__ out, int step,
                                                    uint32_t* __restrict__ ids_out) {
  const uint32_t n0 = g.n[0], n1 = g.n[1], n2 = g.n[2];
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n0 * n1 * n2) return;
  const uint32_t z = i % n2, xy = i / n2, y = xy % n1, x = xy / n1;
  const f3 p = lattice_point(g, x, y, z);
  const uint32_t s = (uint32_t)step;
  // bit (3 a + b) of m[axis]... one flag per axis and direction: is the neighbour at -step / 0 / +step inside the lattice?
  const bool okx[3] = {x >= s, true, x + s < n0}, oky[3] = {y >= s, true, y + s < n1}, okz[3] = {z >= s, true, z + s < n2};
  const int sx = (int)(s * n1 * n2), sy = (int)(s * n2), sz = (int)s;
  unsigned long long key = 0x7f800000ffffffffull;             // (+inf, no triangle)
  uint32_t kbest = 13u;                                      // the point itself
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    float4 c[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const bool ok = okx[a] && oky[k / 3] && okz[k % 3];
      const int off = (a - 1) * sx + (k / 3 - 1) * sy + (k % 3 - 1) * sz;
      c[k] = in[ok ? (uint32_t)((int)i + off) : i];            // an out-of-range neighbour reads the point's own record and is masked below
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const bool ok = okx[a] && oky[k / 3] && okz[k % 3];
      const uint32_t cand = __float_as_uint(c[k].w);
      const float ex = p.x - c[k].x, ey = p.y - c[k].y, ez = p.z - c[k].z;
      const float d = __builtin_fmaf(ex, ex, __builtin_fmaf(ey, ey, ez * ez));
      // d >= +0 orders like its bit pattern; NaN (bits above +inf's) never wins, as `d < bd || d == bd` never held for it
      const unsigned long long kk = ((unsigned long long)__float_as_uint(d) << 32) | cand;
      const bool take = ok && cand != 0xffffffffu && kk < key;
      key = take ? kk : key;
      kbest = take ? (uint32_t)(9 * a + k) : kbest;
    }
  }
  const uint32_t best = (uint32_t)key;
  float4 bc = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(0xffffffffu));
  if (best != 0xffffffffu) {
    const int a = (int)(kbest / 9u), k = (int)(kbest % 9u);
    bc = in[(uint32_t)((int)i + (a - 1) * sx + (k / 3 - 1) * sy + (k % 3 - 1) * sz)];
  }
  out[i] = bc;
  if (ids_out) ids_out[i] = best;
}
// Lattices of at most JFA_SMALL_MAX points (a 64^3 grid: 16^3 brick centres): the whole flood — clear, splat, load, every pass — in ONE
// workgroup with the lattice in LDS.  Seven to nine launches of 2 - 4 us kernels cost the host ~45 us to enqueue, during which the caller's
// stream sat idle behind the build's sort (suzanne, 968 triangles, 64^3: hierarchy started 55 us after the sort had ended; timeline in
// profiles/r06_small_calls.txt).  Same candidates, same tie rule as k_jfa_splat / k_jfa_pass32: the same seeds.
constexpr uint32_t JFA_SMALL_MAX = 4096, JFA_SMALL_THREADS = 1024;
__global__ __launch_bounds__(JFA_SMALL_THREADS) void k_jfa_small(DeviceMesh mesh, GridParams g, uint32_t* __restrict__ ids_out) {
  __shared__ float4 lat_a[JFA_SMALL_MAX], lat_b[JFA_SMALL_MAX];
  unsigned long long* const keys = reinterpret_cast<unsigned long long*>(lat_b);   // the splat's keys live in the second buffer until the load
  const uint32_t n0 = g.n[0], n1 = g.n[1], n2 = g.n[2], n = n0 * n1 * n2, tid = threadIdx.x;
  for (uint32_t i = tid; i < n; i += JFA_SMALL_THREADS) keys[i] = ~0ull;
  __syncthreads();
  for (uint32_t t = tid; t < mesh.n_tris; t += JFA_SMALL_THREADS) {          // k_jfa_splat
    const float4 c = mesh.cen[t];
    const float cc[3] = {c.x, c.y, c.z};
    uint32_t cell[3];
    bool ok = true;
    for (int k = 0; k < 3; ++k) {
      float f = (cc[k] - g.first[k]) / g.size[k] + 0.5f;
      if (!(f == f)) { ok = false; break; }                                  // NaN centroid: not a useful seed
      if (k == 0) { cell[0] = lattice_x_from_real(g, f); continue; }
      f = fminf(fmaxf(f, 0.0f), (float)(g.n[k] - 1));
      cell[k] = min((uint32_t)f, g.n[k] - 1);
    }
    if (!ok) continue;
    const f3 p = lattice_point(g, cell[0], cell[1], cell[2]);
    const float ex = p.x - c.x, ey = p.y - c.y, ez = p.z - c.z;
    const float d2 = __builtin_fmaf(ex, ex, __builtin_fmaf(ey, ey, ez * ez));
    if (!(d2 == d2)) continue;
    atomicMin(&keys[(cell[0] * n1 + cell[1]) * n2 + cell[2]], ((unsigned long long)__float_as_uint(d2) << 32) | t);
  }
  __syncthreads();
  for (uint32_t i = tid; i < n; i += JFA_SMALL_THREADS) {                    // k_jfa_load
    const uint32_t id = (uint32_t)(keys[i] & 0xffffffffull);
    float4 c = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (id != 0xffffffffu) c = mesh.cen[id];
    c.w = __uint_as_float(id);
    lat_a[i] = c;
  }
  __syncthreads();
  const uint32_t maxdim = max(n0, max(n1, n2));
  uint32_t step = 1;
  while (step * 2u < maxdim) step *= 2u;
  float4* in = lat_a;
  float4* out = lat_b;
  for (bool last = false;; ) {                                                // steps ... 2, 1, then one more unit pass (launch_grid_seeds)
    for (uint32_t i = tid; i < n; i += JFA_SMALL_THREADS) {                  // k_jfa_pass32
      const uint32_t z = i % n2, xy = i / n2, y = xy % n1, x = xy / n1;
      const f3 p = lattice_point(g, x, y, z);
      const bool okx[3] = {x >= step, true, x + step < n0}, oky[3] = {y >= step, true, y + step < n1}, okz[3] = {z >= step, true, z + step < n2};
      const int sx = (int)(step * n1 * n2), sy = (int)(step * n2), sz = (int)step;
      unsigned long long key = 0x7f800000ffffffffull;
      float4 bc = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(0xffffffffu));
      for (int a = 0; a < 3; ++a)
        for (int k = 0; k < 9; ++k) {
          if (!(okx[a] && oky[k / 3] && okz[k % 3])) continue;
          const float4 c = in[(uint32_t)((int)i + (a - 1) * sx + (k / 3 - 1) * sy + (k % 3 - 1) * sz)];
          const uint32_t cand = __float_as_uint(c.w);
          const float ex = p.x - c.x, ey = p.y - c.y, ez = p.z - c.z;
          const float d = __builtin_fmaf(ex, ex, __builtin_fmaf(ey, ey, ez * ez));
          const unsigned long long kk = ((unsigned long long)__float_as_uint(d) << 32) | cand;
          if (cand != 0xffffffffu && kk < key) { key = kk; bc = c; }
        }
      out[i] = bc;
      if (last) ids_out[i] = (uint32_t)key;
    }
    __syncthreads();
    float4* t = in; in = out; out = t;
    if (last) break;
    if (step == 1u) last = true; else step >>= 1;
  }
}
// One flooding pass over the lattice g.
static void launch_jfa_pass(hipStream_t st, const GridParams& g, const float4* in, float4* out, int step, uint32_t* ids) {
  const size_t total = (size_t)g.n[0] * g.n[1] * g.n[2];
  const unsigned nb = (unsigned)((total + 255) / 256);
  if (total < (1ull << 30) && (unsigned long long)step * g.n[1] * g.n[2] < (1ull << 30))
    hipLaunchKernelGGL(k_jfa_pass32, dim3(nb), dim3(256), 0, st, g, in, out, step, ids);
  else
    hipLaunchKernelGGL(k_jfa_pass, dim3(nb), dim3(256), 0, st, g, (const GridParams*)nullptr, in, out, step, ids);
}

// ---- k_lane_q: the lane walk for generic queries ---------------------------------------------
// One sorted query per lane, every lane on its own through the tree (as k_lane) and, for the best-of-three-rays sign, through the
// box tree along each axis.  For SPARSE query sets: a packet of 64 of 100 000 queries in the benchmark box is 44 cells of the 512^3
// grid wide, the wave-uniform walk pays for the union of what its lanes need (800 node tests and 340 exact evaluations per packet
// on average, 4 300 and 2 200 for the worst one) and, with fewer packets than the GPU has wave slots, the launch lasts as long as
// that one wave's chain of dependent loads: 3.2 ms for 100 000 queries, 2.7 ms for 1 M.  A lane alone needs ~60 node tests.
template <int AXIS>
__device__ __forceinline__ uint32_t stab_count_lane(const DeviceMesh& mesh, f3 p) {
  uint32_t count = 0, node = 0;
  while (node < mesh.n_nodes) {
    const NodeRec nr = mesh.nodes[node];
    if (!ray_meets_box<AXIS>(p, mk3(nr.mnx, nr.mny, nr.mnz), mk3(nr.mxx, nr.mxy, nr.mxz))) { node = nr.skip; continue; }
    if (nr.tri >= 0) {
      const uint32_t cnt = (nr.skip - node + 1u) >> 1;
      for (uint32_t k = 0; k < cnt; ++k) {
        const uint32_t vi = 3u * ((uint32_t)nr.tri + k);
        const float4 c0 = mesh.corners[vi], c1 = mesh.corners[vi + 1u], c2 = mesh.corners[vi + 2u];
        const f3 a = mk3(c0.x, c0.y, c0.z), b = mk3(c0.w, c1.x, c1.y), c = mk3(c1.z, c1.w, c2.x);
        f3 mn, mx;
        triangle_bounding_box(a, b, c, &mn, &mx);     // the candidate rule is per triangle: ITS padded box
        float t;
        const bool h = ray_meets_box<AXIS>(p, mn, mx) && ray_triangle_aligned<AXIS>(p, a, b, c, &t);
        count += h ? 1u : 0u;
      }
      node = nr.skip;
    } else {
      node = node + 1;
    }
  }
  return count;
}
template <int MODE, int SIGN>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(M2S_LANE_WAVES, 8))) void k_lane_q(DeviceMesh mesh, const float4* __restrict__ qsorted, const uint32_t* __restrict__ perm,
                                                uint32_t n_q, float* __restrict__ out, int* __restrict__ err,
                                                const uint32_t* __restrict__ seed_in, const GridParams* __restrict__ seed_lattice) {
  const uint32_t i_raw = blockIdx.x * blockDim.x + threadIdx.x;
  const bool LANE_VALID = i_raw < n_q;                 // the last wave's spare lanes stay: the wave-cooperative tail of the walk needs all 64
  const uint32_t i = min(i_raw, n_q - 1u);
  const float4 q = qsorted[i];
  const f3 p = mk3(q.x, q.y, q.z);
  Best<MODE> best;
  if (mesh.n_nodes) {
    const float scale = fmaxf(mesh_scale(mesh), fmaxf(fabsf(p.x), fmaxf(fabsf(p.y), fabsf(p.z))));
    const float slack = 4.0e-6f * scale + (MODE == MODE_NORMAL_FOLD ? 2.5e-6f : 0.0f);
    uint32_t slot = 0;
    if (seed_in != nullptr) slot = min(seed_in[query_lattice_cell(*seed_lattice, p.x, p.y, p.z)], mesh.n_tris - 1);
    eval_triangle<MODE>(best, p, mesh.tris[slot]);
    greedy_leaf<MODE>(mesh, p, best);
    uint32_t st_nodes = 0, st_exact = 0;               // M2S_STATS
    __shared__ LaneShare lane_share[4];                // one per wave of the workgroup
    lane_tree_walk<MODE>(mesh, p, slack, best, LANE_VALID, st_nodes, st_exact, lane_share[threadIdx.x >> 6]);
    if (mesh.stats != nullptr && LANE_VALID) {         // per lane: the lane walk's unit is the lane
      atomicAdd(&mesh.stats[0], (unsigned long long)st_nodes);
      atomicAdd(&mesh.stats[2], (unsigned long long)st_exact);
      atomicAdd(&mesh.stats[3], 1ull);
      atomicMax(&mesh.stats[72], (unsigned long long)st_nodes);
      atomicMax(&mesh.stats[73], (unsigned long long)st_exact);
    }
  }
  bool negate = false;
  if (MODE == MODE_UNSIGNED && SIGN == SIGN_RAYS3) {
    const uint32_t cx = stab_count_lane<0>(mesh, p), cy = stab_count_lane<1>(mesh, p);
    uint32_t cz = cx;                                                  // two agreeing parities are the majority (see k_packet)
    if (((cx ^ cy) & 1u) != 0u) cz = stab_count_lane<2>(mesh, p);
    negate = ((cx & 1u) + (cy & 1u) + (cz & 1u)) > 1u;                 // bvh.rs:131-141, rtree_bvh.rs:161-171
  }
  if (!LANE_VALID) return;
  if (MODE == MODE_NORMAL_FOLD && best.nan) atomicOr(err, ERRF_NAN);
  out[perm[i]] = finish<MODE>(best, negate);
}

// ---- k_cut: one wave per 4 x 4 x 4 bricks, lane = brick, one cut list per brick (see CutList) ---------------
// The 64 bricks of a wave are neighbours, so they visit nearly the same top of the tree: the wave walks it ONCE, like
// k_packet does (wave-uniform position, node records through scalar loads, a subtree left when no lane keeps it), one
// test per lane per node; a lane that has dropped or emitted a subtree sits out until the walk has left it (`resume`).
//
// What a brick may drop.  Brick: centre q, every voxel centre v = q + w with |w| <= r.  k_packet evaluates the brick's seed
// triangle T first, so voxel v ends with a minimum <= dist(v, T) <= |v - s|, s = the point of T closest to q, a = q - s,
// D = |a|, e = a / D.  Subtree X lies inside its convex disc-slab C_X; c = the point of C_X closest to q, L = |q - c|,
// n = (q - c) / L, and convexity gives dist(v, C_X) >= n . (v - c) = L + n . w.  X holds nothing within (or tied with) any
// voxel's minimum if   L + n . w > |a + w|   for all |w| <= r.  Two sufficient conditions, either drops X:
//   sphere     L - r > D + r                                                       (1-Lipschitz; the only test so far)
//   gradient   L - D > |n - e| r + r^2 / (2 D)          from |a + w| <= D + e . w + |w|^2 / (2 D)
// The sphere test wastes 2 r = 5 cells: far from the surface (D = 64 cells) it keeps every triangle of a cap of ~25
// cells radius, so the lists had to stay coarse and the packets walked the rest voxel by voxel — 64 lanes repeating
// nearly the same decision (84 % of the packets of 512^3 x blob-100k are farther than 16 cells from the surface and
// they are the expensive ones: 129 node tests at D >= 64 cells against 45 next to the surface).  The gradient test sees
// that all voxels of the brick look at X from (almost) the same direction as at their seed: its slack is
// |n - e| r ~ (lateral offset / D) r, a few tenths of a cell, so the lists can go down to subtrees of a few triangles.
// All margins are far above f32 rounding (relative 1e-4 on lengths, sqrt(2e-5) on |n - e|), always towards keeping.
//
// Earlier versions (512^3 x blob-100k / the 64-layer slab of an 8-GPU rank; lists per block of 2 x 2 x 2 bricks, sphere
// test): one lane per block, per-lane record fetches: 0.27 / 0.24 ms; eight lanes per block: 0.40 / 0.15 ms; one wave per
// eight blocks with scalar record loads: 0.22 / 0.07 ms.
// GRID = false (generic queries): "brick" = packet of sorted queries, lane = packet, 64 consecutive packets (neighbours in
// the Morton order) per wave; centre and radius from `centres` (k_qpacket_bounds), the seed from the lattice cell of the
// centre (as k_packet<false> does), `nbx` = the number of wave slots the packet walk was launched with.
constexpr uint32_t NB_CUT = (uint32_t)sizeof(NodeExt);
// Two levels (round 6; grids only).  Of a fine wave's node visits on 512^3 x blob-100k 44 % fall on nodes larger than the wave's own block
// of 4 x 4 x 4 bricks (16 % on nodes larger than four blocks; 1024^3 x sheet-100k: 61 % / 29 %; counted, profiles/r06_kcut_visits.txt),
// and its 63 neighbours inside a 64^3-voxel region repeat them with the same outcome.  LEVEL 1 walks that top ONCE per region: lane =
// a block of 4 x 4 x 4 bricks (the same tests with the block's radius; witness triangle = the seed of the brick at the block's centre —
// any triangle bounds the final minimum from above), one wave per 4 x 4 x 4 blocks AND per one of CUTC_S subtrees of the tree's top
// (a single wave per region would be a chain of ~300 dependent visits on a launch of a few hundred waves: as long as what it saves),
// and leaves CUTC_S sub-lists of <= CUTC_MAX ranges per block.  LEVEL 2 is the fine walk started from its block's ranges instead of
// the root.  A subtree the coarse level drops holds nothing a voxel of the block can need, whatever the fine level's own seeds say,
// so the fine lists can only get shorter; the packets' results cannot change (parity suite, soaks).
constexpr uint32_t CUTC_S_LOG = 3, CUTC_S = 1u << CUTC_S_LOG, CUTC_WORDS = 8, CUTC_MAX = CUTC_WORDS - 1;   // 64 words = 256 B per block
static_assert(CUTC_S * CUTC_WORDS == 64, "a fine wave fetches its block's coarse record with one load, lane = word");
template <bool GRID, int LEVEL = 0>
__global__ __launch_bounds__(64) void k_cut(DeviceMesh mesh, GridParams g, const uint32_t* __restrict__ seeds, uint32_t seed_shift,
                                            uint32_t seed_ny, uint32_t seed_nz, uint32_t nbx, uint32_t nby, uint32_t nbz,
                                            uint32_t* __restrict__ lists, float emit_near, float emit_far, uint32_t budget, uint32_t wave_cap,
                                            const float4* __restrict__ centres, const uint32_t* __restrict__ table,
                                            const GridParams* __restrict__ seed_lattice, const uint32_t* __restrict__ coarse = nullptr) {
  static_assert(GRID || LEVEL == 0, "the two-level form is the grid's");
  // LEVEL 1: the "bricks" of this launch are blocks of 4 x 4 x 4 packet bricks (nbx, nby, nbz count blocks), eight waves per 4 x 4 x 4 of them
  constexpr uint32_t UL = LEVEL == 1 ? 2u : 0u;               // log2 packet bricks per lane unit and axis
  constexpr uint32_t NMAX = LEVEL == 1 ? CUTC_MAX : CUT_MAX;  // ranges per list
  constexpr uint32_t OUT_WORDS = LEVEL == 1 ? CUTC_WORDS : CUT_WORDS;
  const uint32_t nsy = (nby + 3u) >> 2, nsz = (nbz + 3u) >> 2;
  const uint32_t sb = LEVEL == 1 ? blockIdx.x >> CUTC_S_LOG : blockIdx.x;   // 4 x 4 x 4 units
  const uint32_t sub = LEVEL == 1 ? blockIdx.x & (CUTC_S - 1u) : 0u;        // LEVEL 1: which subtree of the top
  const uint32_t sz = sb % nsz, sy = (sb / nsz) % nsy, sx = sb / (nsz * nsy);
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t bk[3] = {4u * sx + (lane >> 4), 4u * sy + ((lane >> 2) & 3u), 4u * sz + (lane & 3u)};
  const uint32_t pk = blockIdx.x * 64u + lane;               // !GRID: this lane's packet
  bool in_grid = bk[0] < nbx && bk[1] < nby && bk[2] < nbz;
  float r = 0.0f, qq[3];
  if (GRID) {
    // first cell of the unit in the grid (a brick never straddles two chunks of an interleaved slab: capi.hip checks; a block
    // does not either where the coarse level is used: prepare_grid_walk)
    const uint32_t cell0[3] = {slab_x(g, bk[0] << (g.bl[0] + UL)), bk[1] << (g.bl[1] + UL), bk[2] << (g.bl[2] + UL)};
    for (int k = 0; k < 3; ++k) {
      const float hb = 0.5f * (float)((1u << (g.bl[k] + UL)) - 1u) * fabsf(g.size[k]);   // half extent between voxel centres
      r = __builtin_fmaf(hb, hb, r);
      qq[k] = g.first[k] + ((float)cell0[k] + 0.5f * (float)((1u << (g.bl[k] + UL)) - 1u)) * g.size[k];
    }
    r = sqrtf(r) * 1.0001f;
  } else {
    in_grid = pk < nbx && pk < table[0];
    const float4 c = centres[in_grid ? pk : 0u];
    qq[0] = c.x; qq[1] = c.y; qq[2] = c.z;
    r = c.w;                                                 // already rounded up; NaN / inf (non-finite queries): nothing is dropped
    if (!(r < 3.0e37f)) r = __builtin_inff();
  }
  const f3 q = mk3(qq[0], qq[1], qq[2]);
  const float scale = fmaxf(mesh_scale(mesh), fmaxf(fabsf(q.x), fmaxf(fabsf(q.y), fabsf(q.z))) + r);
  const float abs_margin = 6.4e-5f * scale + 4.0e-5f;        // the packet walk's own slack is <= 4e-6 * scale + 2.5e-6
  float R2 = -1.0f, R = 0.0f, D = 0.0f;                      // unit outside the grid: never keeps anything
  f3 e = mk3(0.0f, 0.0f, 0.0f);
  float grad_c0 = __builtin_inff(), grad_c1 = 0.0f;          // gradient test: drop if L * (1 - 1e-4) - grad_c0 > grad_c1 * |n - e|
  if (in_grid) {
    uint32_t sidx;
    if (!GRID) sidx = query_lattice_cell(*seed_lattice, q.x, q.y, q.z);
    else if (LEVEL == 1) {                                    // the brick at the block's centre (seed lattice: one point per brick)
      const uint32_t lx = bricks_along(g.xe - g.xb, g.bl[0]);
      const uint32_t b0 = min((bk[0] << 2) + 2u, lx - 1u), b1 = min((bk[1] << 2) + 2u, seed_ny - 1u), b2 = min((bk[2] << 2) + 2u, seed_nz - 1u);
      sidx = (b0 * seed_ny + b1) * seed_nz + b2;
    } else sidx = ((bk[0] >> seed_shift) * seed_ny + (bk[1] >> seed_shift)) * seed_nz + (bk[2] >> seed_shift);
    const uint32_t slot = min(seeds[sidx], mesh.n_tris - 1);
    const TriRec& t = mesh.tris[slot];
    const f3 a = mk3(t.ax, t.ay, t.az), bq = mk3(t.bx, t.by, t.bz), c = mk3(t.cx, t.cy, t.cz);
    const TriEdges ed = {mk3(t.abx, t.aby, t.abz), mk3(t.acx, t.acy, t.acz), mk3(t.bcx, t.bcy, t.bcz)};
    const f3 s = closest_point_triangle(q, a, bq, c, ed, t.cls);
    const f3 av = sub3(q, s);
    const float d2 = dot3(av, av);
    D = (d2 == d2) ? sqrtf(d2) : __builtin_inff();
    // sphere: the packet walk keeps a node while bound <= d * (1 + PRUNE_REL) + slack: stay well above that
    R = (D * 1.0001f + 2.0f * r) * 1.0003f + abs_margin;
    R2 = R * R;                                              // inf: nothing is dropped
    if (D > r && D < 3.0e37f) {                              // (valid for any D > 0; useless when r^2 / 2D is large)
      const float inv = 1.0f / D;
      e = mk3(av.x * inv, av.y * inv, av.z * inv);
      grad_c0 = D * 1.0003f + (r * r * 0.5f * inv) * 1.01f + abs_margin;
      grad_c1 = r * 1.001f;
    }
  }
  const float grad_c1sq = grad_c1 * grad_c1 * 1.000001f;     // the test compares squares (no root per node): rounded up
  const float emit_radius = fmaxf(emit_near * r, R * emit_far);

  constexpr uint32_t NB = (uint32_t)sizeof(NodeExt);
  const uint32_t tree_end = mesh.n_nodes * NB;
  uint32_t* out = lists + ((size_t)(in_grid ? (GRID ? (bk[0] * nby + bk[1]) * nbz + bk[2] : pk) : 0u)) * (LEVEL == 1 ? CUTC_S * CUTC_WORDS : OUT_WORDS)
                  + sub * CUTC_WORDS;
  const uint32_t cut_S = cut_start_bits(mesh.n_nodes);
  auto cut_word = [cut_S](uint32_t start, uint32_t stop) {   // byte offsets -> list word (see CUT_WORDS)
    return (start / NB_CUT) | (cut_encode_len((stop - start) / NB_CUT, 27u - cut_S) << cut_S);
  };
  uint32_t n = 0, last_start = 0, last_end = 0, resume = 0, opened = 0;   // per lane
  uint32_t off = 0, end = tree_end, steps = 0;                // wave-uniform
  if (LEVEL == 1) {
    // this wave's subtree: CUTC_S_LOG levels down from the root, left or right by the bits of `sub`.  A leaf met on the way belongs to
    // the wave whose remaining bits are zero; the others have nothing to walk.
    for (uint32_t lv = 0; lv < CUTC_S_LOG && off < end; ++lv) {
      const NodeExt* nr = reinterpret_cast<const NodeExt*>(reinterpret_cast<const char*>(mesh.ext) + off);
      const uint32_t rest = sub & ((1u << (CUTC_S_LOG - lv)) - 1u);
      if (__builtin_amdgcn_readfirstlane(nr->tri) >= 0) { if (rest != 0u) end = off; break; }
      const uint32_t left = off + NB;
      const uint32_t right = __builtin_amdgcn_readfirstlane(reinterpret_cast<const NodeExt*>(reinterpret_cast<const char*>(mesh.ext) + left)->skip);
      const uint32_t skip = __builtin_amdgcn_readfirstlane(nr->skip);
      if ((sub >> (CUTC_S_LOG - 1u - lv)) & 1u) { off = right; end = skip; } else { off = left; end = right; }
    }
  }
  // LEVEL 2: the block's coarse record, lane = word; ranges are taken from it one by one
  uint32_t cw = 0, c_sub = 0, c_k = 0, c_cnt = 0;
  if (LEVEL == 2) {
    cw = coarse[(size_t)sb * (CUTC_S * CUTC_WORDS) + lane];
    off = end = 0;
    c_cnt = (uint32_t)__builtin_amdgcn_readlane((int)cw, 0);
  }
#ifdef M2S_STATS_BUILD
  // M2S_STATS: where a wave's node visits go — on nodes larger than the wave's own block of 4 x 4 x 4 bricks (what a coarser level
  // of lists could decide once for several waves) or below
  uint32_t st_visits = 0, st_above1 = 0, st_above4 = 0;
  const float st_block = (LEVEL == 1 ? 1.0f : 4.0f) * __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, r)));
#endif
  for (;;) {
    if (LEVEL == 2) {
      while (c_k >= c_cnt) {                                  // next non-empty sub-list
        if (++c_sub >= CUTC_S) break;
        c_k = 0;
        c_cnt = (uint32_t)__builtin_amdgcn_readlane((int)cw, (int)(c_sub * CUTC_WORDS));
      }
      if (c_sub >= CUTC_S) break;
      const uint32_t w = (uint32_t)__builtin_amdgcn_readlane((int)cw, (int)(c_sub * CUTC_WORDS + 1u + c_k));
      ++c_k;
      const uint32_t first = w & ((1u << cut_S) - 1u);
      const uint32_t len = ((w >> cut_S) & ((1u << (27u - cut_S)) - 1u)) << (w >> 27);
      off = max(first * NB, off);                             // (a rounded-up range may reach into the next one: never walk back)
      end = min(first + len, mesh.n_nodes) * NB;
    }
  while (off < end) {
    off = __builtin_amdgcn_readfirstlane(off);
    ++steps;
    const NodeExt nr = *reinterpret_cast<const NodeExt*>(reinterpret_cast<const char*>(mesh.ext) + off);
#ifdef M2S_STATS_BUILD
    {
      const float ext = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, fmaxf(nr.R, nr.half))));
      ++st_visits;
      st_above1 += ext > st_block ? 1u : 0u;
      st_above4 += ext > 4.0f * st_block ? 1u : 0u;
    }
#endif
    const bool active = off >= resume;                        // this brick has not dropped / emitted an ancestor
    // closest point of the disc-slab to q:  q - c = ax * n_s + lat * l / |l|   (common.h NodeExt, ext_dist2)
    const float vx = q.x - nr.cx, vy = q.y - nr.cy, vz = q.z - nr.cz;
    const float t = __builtin_fmaf(nr.nz, vz, __builtin_fmaf(nr.ny, vy, nr.nx * vx));
    const float v2 = __builtin_fmaf(vz, vz, __builtin_fmaf(vy, vy, vx * vx));
    const float l2 = fmaxf(__builtin_fmaf(-1.0e-6f, v2, __builtin_fmaf(-t, t, v2)), 0.0f);
    const float inv_ell = __builtin_amdgcn_rsqf(l2);                  // one transcendental for ell and 1 / ell (inf at l2 = 0: guarded below)
    const float ell = l2 > 0.0f ? l2 * inv_ell : 0.0f;
    const float lat = fmaxf(ell - nr.R, 0.0f);
    const float dt = t - nr.mid;
    const float ax = copysignf(fmaxf(fabsf(dt) - nr.half, 0.0f), dt);
    const float L2 = __builtin_fmaf(ax, ax, lat * lat);
    bool keep = active & !(L2 > R2);                          // sphere test; NaN keeps the node
    if (__ballot(keep) == 0ull) { off = nr.skip; continue; }
    {
      // gradient test (lanes without it carry grad_c0 = inf: never dropped).  rcp / rsq instead of IEEE divisions: their
      // 1-ulp error is nothing beside the 2e-5 added under the root
      const float ne_s = __builtin_fmaf(nr.nz, e.z, __builtin_fmaf(nr.ny, e.y, nr.nx * e.x));            // n_s . e
      const float ve = __builtin_fmaf(vz, e.z, __builtin_fmaf(vy, e.y, vx * e.x));                       // (q - c0) . e
      const float le = ve - t * ne_s;                                                                    // l . e
      const float lat_dir = lat > 0.0f ? lat * inv_ell : 0.0f;       // lat > 0 means ell > R >= 0
      const float num = __builtin_fmaf(ax, ne_s, lat_dir * le);                                          // (q - c) . e
      const float inv_L = __builtin_amdgcn_rsqf(L2);
      const float cosne = fminf(num * inv_L, 1.0f);                                                      // n . e (NaN / inf if L == 0: kept)
      const float nme2 = fmaxf(2.0f - 2.0f * cosne, 0.0f) + 2.0e-5f;                                     // >= |n - e|^2
      const float A = __builtin_fmaf(L2 * inv_L, 0.9999f, -grad_c0);                                     // L (1 - 1e-4) - c0
      const bool drop = (A > 0.0f) & (A * A > grad_c1sq * nme2);                                         // A > c1 |n - e| without the root; false on NaN
      keep = keep & !drop;
    }
    const unsigned long long bal = __ballot(keep);
    if (bal == 0ull) { off = nr.skip; continue; }
    // Where many triangles are (nearly) equidistant — towards the medial axis, e.g. deep inside a round body — the brick-level
    // test keeps a large part of the tree however far it descends: a brick that has already opened `budget` nodes emits what
    // it meets next as it is and leaves the rest to the packet's per-voxel tests (which are 200 times sharper there).
    // (a saturated list — NMAX ranges — only grows its last range over every gap from here on: nothing finer can be said)
    const bool emit = keep & (nr.tri >= 0 || fmaxf(nr.R, nr.half) <= emit_radius || opened >= budget || n == NMAX || steps >= wave_cap);
    if (emit) {
      // keep this subtree: [off, skip).  Adjacent subtrees merge; past NMAX ranges the last one grows over the gap
      if (n > 0 && (last_end == off || n == NMAX)) last_end = nr.skip;
      else {
        if (n > 0) out[n] = cut_word(last_start, last_end);
        ++n; last_start = off; last_end = nr.skip;
      }
    }
    opened += (keep & !emit) ? 1u : 0u;
    if (active & (emit | !keep)) resume = nr.skip;            // done with this subtree either way
    off = (__ballot(keep & !emit) != 0ull) ? off + NB : nr.skip;   // some brick still has to look inside
  }
    if (LEVEL != 2) break;
  }
#ifdef M2S_STATS_BUILD
  if (mesh.stats != nullptr && lane == 0u) {
    unsigned long long* sc = mesh.stats + (LEVEL == 1 ? 112 : 104);
    atomicAdd(&sc[0], 1ull);
    atomicAdd(&sc[1], (unsigned long long)st_visits);
    atomicAdd(&sc[2], (unsigned long long)st_above1);
    atomicAdd(&sc[3], (unsigned long long)st_above4);
    atomicMax(&sc[4], (unsigned long long)st_visits);
  }
#endif
  if (!in_grid) return;
  if (LEVEL == 1) {                                           // an empty sub-list is fine: the other subtrees hold what the block needs
    if (n > 0) out[n] = cut_word(last_start, last_end);
    out[0] = n;
    return;
  }
  if (n == 0) { n = 1; last_start = 0; last_end = tree_end; }    // cannot happen with finite input; never walk nothing
  out[n] = cut_word(last_start, last_end);
  out[0] = n;
}

// ---- k_brute --------------------------------------------------------------------------------
template <bool GRID, int MODE, int SIGN>
__global__ __launch_bounds__(256) void k_brute(DeviceMesh mesh, GridParams g, const float* __restrict__ queries,
                                               uint32_t n_q, const uint32_t* __restrict__ plane,
                                               float* __restrict__ out, int* __restrict__ err, uint32_t n_packets, PeerOut peers) {
  __shared__ TriRec tile[TILE];
  const int lane = threadIdx.x & 63;
  const uint32_t packet = blockIdx.x * 4 + (threadIdx.x >> 6);
  const bool active = packet < n_packets;

  f3 p = {0, 0, 0};
  size_t out_index = 0;
  bool store = false;
  GridBrick vox{};
  if (active) {
    if (GRID) {
      vox = grid_lane_voxel(g, packet, lane);
      p = grid_point(g, vox);
      out_index = ((size_t)vox.x * g.n[1] + vox.y) * g.n[2] + vox.z - (size_t)g.out_off;
      store = vox.in_range;
    } else {
      const uint32_t i = min(packet * 64u + lane, n_q - 1);
      p = mk3(queries[3 * (size_t)i], queries[3 * (size_t)i + 1], queries[3 * (size_t)i + 2]);
      out_index = i;
      store = packet * 64u + lane < n_q;
    }
  }

  Best<MODE> best;
  uint32_t hits[3] = {0, 0, 0};
  for (uint32_t t0 = 0; t0 < mesh.n_tris; t0 += TILE) {
    const uint32_t nt = min((uint32_t)TILE, mesh.n_tris - t0);
    __syncthreads();
    {  // 128 records x 96 B = 768 float4; 256 threads x 3
      const float4* src = reinterpret_cast<const float4*>(mesh.tris + t0);
      float4* dst = reinterpret_cast<float4*>(tile);
      for (uint32_t i = threadIdx.x; i < nt * 6; i += 256) dst[i] = src[i];
    }
    __syncthreads();
    for (uint32_t k = 0; k < nt; ++k) {
      const TriRec& tr = tile[k];
      const f3 a = mk3(tr.ax, tr.ay, tr.az), b = mk3(tr.bx, tr.by, tr.bz), c = mk3(tr.cx, tr.cy, tr.cz);
      eval_triangle<MODE>(best, p, tr);
      if (MODE == MODE_UNSIGNED && SIGN == SIGN_XRAY_ALL) {
        float t;
        hits[0] += ray_triangle_aligned<0>(p, a, b, c, &t) ? 1u : 0u;   // default.rs:35-37: every triangle
      }
      if (MODE == MODE_UNSIGNED && SIGN == SIGN_RAYS3) {
        f3 mn, mx;
        triangle_bounding_box(a, b, c, &mn, &mx);
        float t;
        hits[0] += (ray_meets_box<0>(p, mn, mx) & ray_triangle_aligned<0>(p, a, b, c, &t)) ? 1u : 0u;
        hits[1] += (ray_meets_box<1>(p, mn, mx) & ray_triangle_aligned<1>(p, a, b, c, &t)) ? 1u : 0u;
        hits[2] += (ray_meets_box<2>(p, mn, mx) & ray_triangle_aligned<2>(p, a, b, c, &t)) ? 1u : 0u;
      }
    }
  }

  bool negate = false;
  if (MODE == MODE_UNSIGNED) {
    if (SIGN == SIGN_GRID_PLANE && active) {
      const size_t w = ((size_t)vox.x * g.n[1] + vox.y) * g.nzw + (vox.z >> 5);
      negate = (plane[w] >> (vox.z & 31u)) & 1u;
    } else if (SIGN == SIGN_XRAY_ALL) {
      negate = hits[0] & 1u;                                            // default.rs:65-72
    } else if (SIGN == SIGN_RAYS3) {
      negate = ((hits[0] & 1u) + (hits[1] & 1u) + (hits[2] & 1u)) > 1u;
    }
  }
  if (MODE == MODE_NORMAL_FOLD && best.nan && store) atomicOr(err, ERRF_NAN);
  const float result = finish<MODE>(best, negate);
  if (store) out[out_index] = result;
  if (GRID && store)
    for (uint32_t i = 0; i < peers.n; ++i) peers.p[i][out_index + (size_t)g.out_off] = result;
}

// ---- k_brute_split: tiny problems without a tree -------------------------------------------------------------------
// The reference's own criterion shapes include a 16^3 grid over an 11 k-triangle mesh (benches/generate_grid_sdf.rs:8-34): 4 096
// voxels are 64 waves, each lane walks the tree alone, and the call lasts as long as its slowest lane's chain of dependent loads
// (0.7 ms) on top of a 0.17 ms build.  All voxels x all triangles is only 46 M evaluations there — 0.1 ms if the whole chip takes
// part, and no tree is needed at all.  k_brute gives a block ALL triangles (16 blocks for 16^3); here the triangles are cut into
// chunks as well: block (x, y) evaluates voxel block x against triangle chunk y and folds its minima into per-voxel words with
// atomic minima (non-negative floats order like their bit patterns; min is associative and commutative: bit-identical to k_brute
// and to every walk), k_brute_finish turns them into signed distances.  Chosen for cells x triangles <= M2S_BRUTE_MAX.
template <int MODE>
__global__ __launch_bounds__(256) void k_brute_split(DeviceMesh mesh, GridParams g, uint32_t* __restrict__ acc, int* __restrict__ err,
                                                     uint32_t n_packets, uint32_t tiles_per_chunk) {
  __shared__ TriRec tile[TILE];
  const int lane = threadIdx.x & 63;
  const uint32_t packet = blockIdx.x * 4 + (threadIdx.x >> 6);
  const bool active = packet < n_packets;
  f3 p = {0, 0, 0};
  bool store = false;
  if (active) {
    const GridBrick vox = grid_lane_voxel_plain(g, packet, lane);
    p = grid_point(g, vox);
    store = vox.in_range;
  }
  Best<MODE> best;
  const uint32_t t_begin = blockIdx.y * tiles_per_chunk * TILE, t_end = min(mesh.n_tris, t_begin + tiles_per_chunk * TILE);
  for (uint32_t t0 = t_begin; t0 < t_end; t0 += TILE) {
    const uint32_t nt = min((uint32_t)TILE, t_end - t0);
    __syncthreads();
    {
      const float4* src = reinterpret_cast<const float4*>(mesh.tris + t0);
      float4* dst = reinterpret_cast<float4*>(tile);
      for (uint32_t i = threadIdx.x; i < nt * 6; i += 256) dst[i] = src[i];
    }
    __syncthreads();
    for (uint32_t k = 0; k < nt; ++k) eval_triangle<MODE>(best, p, tile[k]);
  }
  if (!active || !store) return;
  const size_t slot = ((size_t)packet * 64u + lane) * 2u;
  atomicMin(&acc[slot], __float_as_uint(best.d2));
  if (MODE == MODE_NORMAL_FOLD) {
    atomicMin(&acc[slot + 1], __float_as_uint(best.d2pos));
    if (best.nan) atomicOr(err, ERRF_NAN);
  }
}
template <int MODE, int SIGN>
__global__ __launch_bounds__(256) void k_brute_finish(GridParams g, const uint32_t* __restrict__ plane, const uint32_t* __restrict__ acc,
                                                      float* __restrict__ out, uint32_t n_packets, PeerOut peers) {
  const int lane = threadIdx.x & 63;
  const uint32_t packet = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (packet >= n_packets) return;
  const GridBrick vox = grid_lane_voxel_plain(g, packet, lane);
  if (!vox.in_range) return;
  const size_t slot = ((size_t)packet * 64u + lane) * 2u;
  Best<MODE> best;
  best.d2 = __uint_as_float(acc[slot]);
  best.d2pos = __uint_as_float(acc[slot + 1]);
  bool negate = false;
  if (MODE == MODE_UNSIGNED && SIGN == SIGN_GRID_PLANE) {
    const size_t w = ((size_t)vox.x * g.n[1] + vox.y) * g.nzw + (vox.z >> 5);
    negate = (plane[w] >> (vox.z & 31u)) & 1u;
  }
  const float result = finish<MODE>(best, negate);
  const size_t out_index = ((size_t)vox.x * g.n[1] + vox.y) * g.n[2] + vox.z - (size_t)g.out_off;
  out[out_index] = result;
  for (uint32_t i = 0; i < peers.n; ++i) peers.p[i][out_index + (size_t)g.out_off] = result;
}

// ---- k_brute_split_q: small query sets without a tree ---------------------------------------------------------------
// The crate's documented use is a handful of query points (lib.rs:13-31, examples/demo.rs:29-54): for those the LBVH build (0.16 -
// 0.24 ms), the query sort and a lane walk that lasts as long as its slowest lane's chain of dependent loads (~1 ms) are all
// overhead — queries x triangles is a few 10^7 evaluations, 0.1 - 0.3 ms if the whole chip takes part.  As k_brute_split: block
// (x, y) evaluates query block x against triangle chunk y and folds into per-query words with atomics — minima for the distances
// (non-negative floats order like their bits), a 64-bit (d2, index, !positive) key for the Rtree rule (lowest index on ties, as
// the walks and k_brute take it), XOR for the three ray parities (the parity of a sum is the XOR of the parities) — and
// k_brute_finish_q turns the words into signed distances.  Per query: [0] d2, [1] d2 of the positive side (Normal fold) or the
// hit parities (bits 0..2: +X, +Y, +Z), [2..3] the key.  Same arithmetic per pair as everywhere else: bit-identical.
__global__ __launch_bounds__(256) void k_brute_q_init(uint32_t* __restrict__ acc, uint32_t n_q, uint32_t second) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_q) return;
  reinterpret_cast<uint4*>(acc)[i] = make_uint4(0x7f800000u, second, 0xffffffffu, 0xffffffffu);   // +inf, +inf or no hits, "no triangle"
}
template <int MODE, int SIGN>
__global__ __launch_bounds__(256) void k_brute_split_q(DeviceMesh mesh, const float* __restrict__ queries, uint32_t n_q, uint32_t* __restrict__ acc,
                                                       int* __restrict__ err, uint32_t tiles_per_chunk) {
  __shared__ TriRec tile[TILE];
  const uint32_t i_raw = blockIdx.x * 256u + threadIdx.x, i = min(i_raw, n_q - 1u);
  const f3 p = mk3(queries[3 * (size_t)i], queries[3 * (size_t)i + 1], queries[3 * (size_t)i + 2]);
  Best<MODE> best;
  uint32_t hits[3] = {0, 0, 0};
  const uint32_t t_begin = blockIdx.y * tiles_per_chunk * TILE, t_end = min(mesh.n_tris, t_begin + tiles_per_chunk * TILE);
  for (uint32_t t0 = t_begin; t0 < t_end; t0 += TILE) {
    const uint32_t nt = min((uint32_t)TILE, t_end - t0);
    __syncthreads();
    {
      const float4* src = reinterpret_cast<const float4*>(mesh.tris + t0);
      float4* dst = reinterpret_cast<float4*>(tile);
      for (uint32_t k = threadIdx.x; k < nt * 6; k += 256) dst[k] = src[k];
    }
    __syncthreads();
    for (uint32_t k = 0; k < nt; ++k) {
      const TriRec& tr = tile[k];
      eval_triangle<MODE>(best, p, tr);
      if (MODE == MODE_UNSIGNED && SIGN == SIGN_RAYS3) {               // the candidate rule of bvh.rs:119 / rtree_bvh.rs:149: the triangle's own padded box
        const f3 a = mk3(tr.ax, tr.ay, tr.az), b = mk3(tr.bx, tr.by, tr.bz), c = mk3(tr.cx, tr.cy, tr.cz);
        f3 mn, mx;
        triangle_bounding_box(a, b, c, &mn, &mx);
        float t;
        hits[0] += (ray_meets_box<0>(p, mn, mx) & ray_triangle_aligned<0>(p, a, b, c, &t)) ? 1u : 0u;
        hits[1] += (ray_meets_box<1>(p, mn, mx) & ray_triangle_aligned<1>(p, a, b, c, &t)) ? 1u : 0u;
        hits[2] += (ray_meets_box<2>(p, mn, mx) & ray_triangle_aligned<2>(p, a, b, c, &t)) ? 1u : 0u;
      }
    }
  }
  if (i_raw >= n_q) return;
  uint32_t* w = acc + 4 * (size_t)i;
  if (MODE == MODE_NEAREST_NORMAL) {
    if (best.idx != 0xffffffffu)
      atomicMin(reinterpret_cast<unsigned long long*>(w + 2), ((unsigned long long)__float_as_uint(best.d2) << 32) | ((unsigned long long)best.idx << 1) | (best.pos ? 0ull : 1ull));
    return;
  }
  atomicMin(&w[0], __float_as_uint(best.d2));
  if (MODE == MODE_NORMAL_FOLD) {
    atomicMin(&w[1], __float_as_uint(best.d2pos));
    if (best.nan) atomicOr(err, ERRF_NAN);
  } else if (SIGN == SIGN_RAYS3) {
    const uint32_t par = (hits[0] & 1u) | ((hits[1] & 1u) << 1) | ((hits[2] & 1u) << 2);
    if (par) atomicXor(&w[1], par);
  }
}
template <int MODE, int SIGN>
__global__ __launch_bounds__(256) void k_brute_finish_q(const uint32_t* __restrict__ acc, uint32_t n_q, float* __restrict__ out) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n_q) return;
  const uint4 w = reinterpret_cast<const uint4*>(acc)[i];
  Best<MODE> best;
  bool negate = false;
  if (MODE == MODE_NEAREST_NORMAL) {
    best.d2 = __uint_as_float(w.w);                                    // high word of the key; "no triangle" reads as NaN, as a walk over nothing would not
    best.pos = (w.z & 1u) == 0u;
    if (w.w == 0xffffffffu && w.z == 0xffffffffu) { best.d2 = __builtin_inff(); best.pos = false; }
  } else {
    best.d2 = __uint_as_float(w.x);
    if (MODE == MODE_NORMAL_FOLD) best.d2pos = __uint_as_float(w.y);
    else if (SIGN == SIGN_RAYS3) negate = ((w.y & 1u) + ((w.y >> 1) & 1u) + ((w.y >> 2) & 1u)) > 1u;   // bvh.rs:131-141, rtree_bvh.rs:161-171
  }
  out[i] = finish<MODE>(best, negate);
}

// ---- query ordering (generic path): Morton sort so that a packet is spatially compact ---------
__device__ __forceinline__ int ordf(float f) {
  int i = __float_as_int(f);
  return i >= 0 ? i : i ^ 0x7fffffff;
}
__device__ __forceinline__ float unordf(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7fffffff); }

// Bounding box of the queries (order-encoded ints): grid-stride partials per block, folded by a second
// one-block launch — no atomics on six hot addresses.
constexpr unsigned QB_BLOCKS = 1024;
__device__ __forceinline__ void qb_block_reduce(int lo[3], int hi[3], int* __restrict__ dst) {
  __shared__ int part[6][4];
  const int wv = threadIdx.x >> 6;
  for (int k = 0; k < 3; ++k) {
    int l = lo[k], h = hi[k];
    for (int off = 32; off > 0; off >>= 1) { l = min(l, __shfl_xor(l, off)); h = max(h, __shfl_xor(h, off)); }
    if ((threadIdx.x & 63) == 0) { part[k][wv] = l; part[3 + k][wv] = h; }
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    const int k = threadIdx.x;
    int v = part[k][0];
    for (int w = 1; w < 4; ++w) v = k < 3 ? min(v, part[k][w]) : max(v, part[k][w]);
    dst[k] = v;
  }
}
__global__ __launch_bounds__(256) void k_qbounds(const float* __restrict__ q, uint32_t n_q, int* __restrict__ partial) {
  int lo[3] = {INT32_MAX, INT32_MAX, INT32_MAX}, hi[3] = {INT32_MIN, INT32_MIN, INT32_MIN};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_q; i += (size_t)gridDim.x * blockDim.x)
    for (int k = 0; k < 3; ++k) {
      const float v = q[3 * i + k];
      if (v == v && fabsf(v) < 3.0e38f) { const int o = ordf(v); lo[k] = min(lo[k], o); hi[k] = max(hi[k], o); }
    }
  qb_block_reduce(lo, hi, partial + 6 * blockIdx.x);
}
__global__ __launch_bounds__(256) void k_qbounds_final(const int* __restrict__ partial, uint32_t n_blocks, int* __restrict__ b) {
  int lo[3] = {INT32_MAX, INT32_MAX, INT32_MAX}, hi[3] = {INT32_MIN, INT32_MIN, INT32_MIN};
  for (uint32_t j = threadIdx.x; j < n_blocks; j += 256)
    for (int k = 0; k < 3; ++k) { lo[k] = min(lo[k], partial[6 * j + k]); hi[k] = max(hi[k], partial[6 * j + 3 + k]); }
  qb_block_reduce(lo, hi, b);
}
// 30-bit Morton key of a query in the query bounding box (10 bits per axis: 1024^3 cells — far finer than a packet of 64 of
// any realistic query count, and a 32-bit key sorts in four radix passes instead of the eight of the 63-bit key used before:
// 0.83 -> 0.45 ms for 10 M queries).  Queries of one cell keep their input order among themselves.
constexpr int QKEY_BITS = 30;
__device__ __forceinline__ uint32_t expand10q(uint32_t v) {
  uint32_t x = v & 0x3ffu;
  x = (x | x << 16) & 0x030000ffu;
  x = (x | x << 8) & 0x0300f00fu;
  x = (x | x << 4) & 0x030c30c3u;
  x = (x | x << 2) & 0x09249249u;
  return x;
}
// `drop`: low key bits cleared.  The sort then runs over the bits [drop, 30) only — 10 M queries need 21 bits (2 M cells) to form their
// packets, three radix passes instead of four; queries of one finest cell stay in input order, which k_qcells treats like identical keys.
__global__ __launch_bounds__(256) void k_qkeys(const float* __restrict__ q, uint32_t n_q, const int* __restrict__ b,
                                               uint32_t* __restrict__ keys, uint32_t* __restrict__ vals, uint32_t drop) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_q) return;
  uint32_t c[3];
  for (int k = 0; k < 3; ++k) {
    const float lo = unordf(b[k]), hi = unordf(b[3 + k]);
    float u = (q[3 * (size_t)i + k] - lo) / (hi - lo);
    u = (u == u) ? fminf(fmaxf(u, 0.0f), 1.0f) : 0.0f;
    c[k] = min((uint32_t)(u * 1024.0f), 1023u);
  }
  keys[i] = (((expand10q(c[0]) << 2) | (expand10q(c[1]) << 1) | expand10q(c[2])) >> drop) << drop;
  vals[i] = i;
}
// Seed lattice for generic queries: QL^3 cells over the query bounding box (description kept on the device).
constexpr uint32_t QL = 64;
__global__ void k_qlattice(const int* __restrict__ b, GridParams* __restrict__ L) {
  if (threadIdx.x != 0) return;
  GridParams g{};
  for (int k = 0; k < 3; ++k) {
    const float lo = unordf(b[k]), hi = unordf(b[3 + k]);
    float cs = (hi - lo) / (float)QL;
    if (!(cs > 0.0f) || !(cs < 3.0e38f)) cs = 1.0f;
    g.n[k] = QL;
    g.size[k] = cs;
    g.first[k] = ((lo == lo && fabsf(lo) < 3.0e38f) ? lo : 0.0f) + 0.5f * cs;
  }
  g.xb = 0; g.xe = QL; g.nzw = 0; g.out_off = 0; g.chunk_log = 31; g.period = 0;
  *L = g;
}

__global__ __launch_bounds__(256) void k_qgather(const float* __restrict__ q, const uint32_t* __restrict__ perm,
                                                 uint32_t n_q, float4* __restrict__ sorted) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_q) return;
  const size_t s = perm[i];
  sorted[i] = make_float4(q[3 * s], q[3 * s + 1], q[3 * s + 2], 0.0f);
}

// Packets of the generic path.  64 CONSECUTIVE queries of the Morton order are a loose group (the run straddles cell
// boundaries of every level: bounding radius 1.6 x that of a cube holding 64 uniform points, r^2 2.9 x) and the wave-uniform
// walk pays for the union of what its 64 lanes need.  The packets are therefore the LEAVES OF THE BUCKET K-D TREE over the
// keys, capacity 64: the largest key-prefix cells holding at most 64 queries — aligned boxes of aspect <= 2, 46 queries on
// average for uniform points (1.38 x the packets, radius 0.88, r^2 0.78 of that cube's).  No tree is built: with
// w[j] = common prefix length of keys j and j + 64, query i sits in an over-full cell of prefix length b iff some window
// j in [i - 64, i] has w[j] >= b, so its leaf has prefix length m(i) + 1, m(i) = max of w over those windows, the same for
// every query of the leaf; i starts a packet iff it differs from i - 1 within that prefix.  More than 64 queries with
// identical keys (m = QKEY_BITS) are cut at multiples of 64.
__global__ __launch_bounds__(256) void k_qcells(const uint32_t* __restrict__ keys, uint32_t n, uint8_t* __restrict__ head) {
  __shared__ uint32_t sk[256 + 128];   // keys[base - 64, base + 320)
  __shared__ int sw[256 + 64];         // w[j], j in [base - 64, base + 256)
  const long long base = (long long)blockIdx.x * 256;
  for (uint32_t t = threadIdx.x; t < 384u; t += 256u) {
    const long long idx = base - 64 + t;
    sk[t] = (idx >= 0 && idx < (long long)n) ? keys[idx] : 0u;
  }
  __syncthreads();
  for (uint32_t t = threadIdx.x; t < 320u; t += 256u) {
    const long long j = base - 64 + t;
    const uint32_t x = sk[t] ^ sk[t + 64];
    sw[t] = (j >= 0 && j + 64 < (long long)n) ? (x == 0u ? QKEY_BITS : __clz((int)x) - (32 - QKEY_BITS)) : -1;
  }
  __syncthreads();
  const long long i = base + threadIdx.x;
  if (i >= (long long)n) return;
  int m = -1;
  for (uint32_t t = 0; t <= 64u; ++t) m = max(m, sw[threadIdx.x + t]);
  const uint32_t plen = (uint32_t)min(m + 1, QKEY_BITS);
  const uint32_t key = sk[threadIdx.x + 64], prev = sk[threadIdx.x + 63];
  bool h = i == 0 || (plen != 0u && ((key ^ prev) >> ((uint32_t)QKEY_BITS - plen)) != 0u);
  if (m >= QKEY_BITS) h |= (i & 63) == 0;                   // more than 64 queries in one cell of the finest level
  head[i] = h ? 1 : 0;
}
__global__ void k_qtable_mode(uint32_t* __restrict__ table, uint32_t n, uint32_t launched) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const bool over = table[0] > launched;                    // cannot be ruled out (63 levels of 1 + 64 splits): consecutive packets then
  table[1] = over ? 1u : 0u;
  if (over) table[0] = (n + 63u) / 64u;
}

// (centre, radius) of the bounding box of every packet's queries: one wave per packet.  The radius is rounded up; a packet
// with a non-finite coordinate gets radius inf (its cut list then keeps the whole tree).
// `raw` != nullptr: the kernel also brings the packet's queries into sorted order (sorted[i] = raw[perm[i]]; the packets partition the
// sorted range, so every query is written once) — the gather that k_qgather does in a pass of its own otherwise.
__global__ __launch_bounds__(256) void k_qpacket_bounds(float4* __restrict__ sorted, const uint32_t* __restrict__ table,
                                                        uint32_t n_q, uint32_t launched, float4* __restrict__ centres,
                                                        const float* __restrict__ raw, const uint32_t* __restrict__ perm) {
  const uint32_t packet = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
  if (packet >= launched) return;
  uint32_t first, cnt;
  if (!query_packet_range(table, packet, n_q, &first, &cnt)) return;
  float4 v;
  if (raw != nullptr) {
    const size_t s = perm[first + min(lane, cnt - 1u)];
    v = make_float4(raw[3 * s], raw[3 * s + 1], raw[3 * s + 2], 0.0f);
    if (lane < cnt) sorted[first + lane] = v;
  } else {
    v = sorted[first + min(lane, cnt - 1u)];
  }
  float lo[3] = {v.x, v.y, v.z}, hi[3] = {v.x, v.y, v.z};
  bool bad = !(fabsf(v.x) < 3.0e37f) | !(fabsf(v.y) < 3.0e37f) | !(fabsf(v.z) < 3.0e37f);
  for (int o = 32; o >= 1; o >>= 1)
    for (int k = 0; k < 3; ++k) {
      lo[k] = fminf(lo[k], __shfl_xor(lo[k], o));
      hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], o));
    }
  bad = __ballot(bad) != 0ull;
  if (lane != 0u) return;
  float c[3], r2 = 0.0f;
  for (int k = 0; k < 3; ++k) {
    c[k] = 0.5f * lo[k] + 0.5f * hi[k];
    const float h = fmaxf(hi[k] - c[k], c[k] - lo[k]);
    r2 = __builtin_fmaf(h, h, r2);
  }
  float r = sqrtf(r2) * 1.0001f + 1.0e-30f;
  if (bad) { c[0] = c[1] = c[2] = 0.0f; r = __builtin_inff(); }
  centres[packet] = make_float4(c[0], c[1], c[2], r);
}

template <bool GRID, int MODE, int SIGN>
void launch_packet(hipStream_t st, const DeviceMesh& mesh, const GridParams& g, const float4* qs, const uint32_t* perm,
                   uint32_t n_q, const uint32_t* plane, float* out, int* err, uint32_t n_packets,
                   const uint32_t* seed_in = nullptr, uint32_t seed_shift = 0, uint32_t seed_ny = 0,
                   uint32_t seed_nz = 0, const GridParams* seed_lattice = nullptr, CutList cut = {nullptr, 0, 0, 0, 0, nullptr},
                   const PeerOut* peers_in = nullptr, const SplitCtl* split_in = nullptr, int defer = 0) {
  PeerOut peers{};
  if (peers_in) peers = *peers_in;
  SplitCtl split{};
  if (split_in) split = *split_in;
  const uint32_t per = 8u << XCD_RUN_LOG;                              // one run on each of the eight XCDs
  const uint32_t grid_blocks = ((n_packets + per - 1) / per) * per;    // a whole number of runs per XCD (xcd_remap)
#ifdef M2S_STATS_BUILD
  // M2S_STATS: the counting variant (a few SALU ops more per node); never suspended, so that a packet's counters are whole.  Only the
  // side library libm2s_stats.so (make stats; tools/exp_stats.py loads it through M2S_LIB) carries these instantiations: the product
  // library's code object is seven k_packet variants smaller.
  if (mesh.stats != nullptr)
    hipLaunchKernelGGL((k_packet<GRID, MODE, SIGN, true, false>), dim3(grid_blocks), dim3(64), 0, st, mesh, g,
                       qs, perm, n_q, plane, out, err, n_packets, seed_in, seed_shift, seed_ny, seed_nz, seed_lattice, cut, peers, split);
  else
#endif
  if (defer == 3 && GRID && MODE != MODE_NEAREST_NORMAL && split.cnt != nullptr)
    hipLaunchKernelGGL((k_packet<GRID, MODE == MODE_NEAREST_NORMAL ? MODE_UNSIGNED : MODE, SIGN, false, GRID, GRID ? 3 : 1>), dim3(grid_blocks), dim3(64), 0, st, mesh, g,
                       qs, perm, n_q, plane, out, err, n_packets, seed_in, seed_shift, seed_ny, seed_nz, seed_lattice, cut, peers, split);
  else if (defer == 3)
    hipLaunchKernelGGL((k_packet<GRID, MODE, SIGN, false, false, 3>), dim3(grid_blocks), dim3(64), 0, st, mesh, g,
                       qs, perm, n_q, plane, out, err, n_packets, seed_in, seed_shift, seed_ny, seed_nz, seed_lattice, cut, peers, split);
  else if (GRID && MODE != MODE_NEAREST_NORMAL && split.cnt != nullptr)   // (a split walk always queues its evaluations: the follow-up rounds do)
    hipLaunchKernelGGL((k_packet<GRID, MODE == MODE_NEAREST_NORMAL ? MODE_UNSIGNED : MODE, SIGN, false, GRID, 1>), dim3(grid_blocks), dim3(64), 0, st, mesh, g,
                       qs, perm, n_q, plane, out, err, n_packets, seed_in, seed_shift, seed_ny, seed_nz, seed_lattice, cut, peers, split);
  else if (defer == 2 && GRID)
    hipLaunchKernelGGL((k_packet<GRID, MODE, SIGN, false, false, GRID ? 2 : 1>), dim3(grid_blocks), dim3(64), 0, st, mesh, g,
                       qs, perm, n_q, plane, out, err, n_packets, seed_in, seed_shift, seed_ny, seed_nz, seed_lattice, cut, peers, split);
  else if (defer)
    hipLaunchKernelGGL((k_packet<GRID, MODE, SIGN, false, false, 1>), dim3(grid_blocks), dim3(64), 0, st, mesh, g,
                       qs, perm, n_q, plane, out, err, n_packets, seed_in, seed_shift, seed_ny, seed_nz, seed_lattice, cut, peers, split);
  else
    hipLaunchKernelGGL((k_packet<GRID, MODE, SIGN, false, false>), dim3(grid_blocks), dim3(64), 0, st, mesh, g,
                       qs, perm, n_q, plane, out, err, n_packets, seed_in, seed_shift, seed_ny, seed_nz, seed_lattice, cut, peers, split);
}
// The follow-up rounds and the finish of a split grid walk (after launch_packet on the same stream).
template <int MODE, int SIGN>
void launch_split_rounds(hipStream_t st, const DeviceMesh& mesh, const GridParams& g, const uint32_t* plane, float* out, int* err,
                         const SplitCtl& split, const CutList& cut, const PeerOut* peers_in) {
  PeerOut peers{};
  if (peers_in) peers = *peers_in;
  // as many single-wave workgroups as two rounds of the chip's wave slots: enough to fill it whatever the items' lengths, few enough
  // that a wave gets several items of a long list
  constexpr unsigned waves = 16384;
  const bool trace = tuning().split_report >= 2;   // debugging aid: which launch faults
  if (trace) { const hipError_t e = hipStreamSynchronize(st); fprintf(stderr, "[m2s split] packet launch: %s\n", hipGetErrorString(e)); }
  for (uint32_t r = 1; r <= split.rounds; ++r) {
    hipLaunchKernelGGL((k_split_round<MODE>), dim3(waves), dim3(64), 0, st, mesh, g, split, cut, r, r == split.rounds, err);
    if (trace) { const hipError_t e = hipStreamSynchronize(st); fprintf(stderr, "[m2s split] round %u: %s\n", r, hipGetErrorString(e)); }
  }
  hipLaunchKernelGGL((k_split_finish<MODE, SIGN>), dim3(2048), dim3(256), 0, st, g, plane, out, split, peers);
  if (trace) { const hipError_t e = hipStreamSynchronize(st); fprintf(stderr, "[m2s split] finish: %s\n", hipGetErrorString(e)); }
}
template <bool GRID, int MODE, int SIGN>
void launch_brute(hipStream_t st, const DeviceMesh& mesh, const GridParams& g, const float* q, uint32_t n_q,
                  const uint32_t* plane, float* out, int* err, uint32_t n_packets, const PeerOut* peers_in = nullptr) {
  PeerOut peers{};
  if (peers_in) peers = *peers_in;
  hipLaunchKernelGGL((k_brute<GRID, MODE, SIGN>), dim3((n_packets + 3) / 4), dim3(256), 0, st, mesh, g, q, n_q, plane,
                     out, err, n_packets, peers);
}

uint32_t host_brick_count(const GridParams& g) {   // padded to whole super-bricks
  const uint32_t nbx = bricks_along(g.xe - g.xb, g.bl[0]), nby = bricks_along(g.n[1], g.bl[1]), nbz = bricks_along(g.n[2], g.bl[2]);
  const uint32_t xl = super_brick_xlog(nbx, g.xl_cap);
  return ((nbx + (1u << xl) - 1u) >> xl) * ((nby + 7) >> 3) * ((nbz + 7) >> 3) * (64u << xl);
}

}  // namespace

// Coarse lattice whose points sit at the centres of the `stride`-sized blocks of `fine`.
static GridParams coarse_level(const GridParams& fine, const uint32_t log2_stride[3], uint32_t x_origin) {
  GridParams c = fine;
  for (int k = 0; k < 3; ++k) {
    const uint32_t span = k == 0 ? fine.xe - fine.xb : fine.n[k];
    const uint32_t stride = 1u << log2_stride[k];
    c.n[k] = (span + stride - 1) / stride;
    c.first[k] = fine.first[k] + ((float)(k == 0 ? x_origin : 0u) + 0.5f * (float)(stride - 1u)) * fine.size[k];   // brick centre
    c.size[k] = (float)stride * fine.size[k];
  }
  c.xb = 0;
  c.xe = c.n[0];
  c.out_off = 0;
  if (fine.chunk_log < 31u) {           // interleaved slab: chunk and period in lattice points (whole numbers: capi.hip checks)
    c.chunk_log = fine.chunk_log - log2_stride[0];
    c.period = fine.period >> log2_stride[0];
  }
  return c;
}

static size_t cut_blocks(const GridParams& g, uint32_t log) {
  const uint32_t nbx = bricks_along(g.xe - g.xb, g.bl[0]), nby = bricks_along(g.n[1], g.bl[1]), nbz = bricks_along(g.n[2], g.bl[2]);
  return (size_t)bricks_along(nbx, log) * bricks_along(nby, log) * bricks_along(nbz, log);
}

// Tiny problems take k_brute_split: at most 2^22 cells and cells x triangles <= 1e8 + 3000 x triangles (M2S_BRUTE_MAX overrides the
// product's limit).  Measured (tools/exp_tiny.py, whole calls, brute / build + walk): blob-11k 16^3
// 0.34 / 0.92 ms, 20^3 0.58 / 0.96, 24^3 0.92 / 0.83; blob-100k 8^3 0.41 / 2.21, 12^3 1.08 / 2.55, 16^3 2.21 / 2.22; blob-6k 16^3 0.20 / 0.74,
// 32^3 1.10 / 0.61 — brute force runs at 178 G point-triangle pairs per second (half the chip's fp32 issue rate), the walks of such
// grids as long as their slowest lane's chain of dependent loads, which grows with the mesh.
// Split walk: accumulator slots for up to SPLIT_CAP_SLOTS suspended packets, lists of SPLIT_ITEMS_PER_SLOT items per slot and round.
// A launch of more packets than SPLIT_MAX_PACKETS is not split at all: it is hundreds of rounds of the chip's wave slots deep, its tail a
// percent or two of it.  Below that every packet has a slot (a suspended packet can always hand over: no path back into the walk).
constexpr uint32_t SPLIT_MAX_PACKETS = 1u << 19, SPLIT_ITEMS_PER_SLOT = 8;
static uint32_t split_cap_slots(size_t packets) { return (uint32_t)std::min<size_t>(std::max<size_t>(packets, 64), SPLIT_MAX_PACKETS); }
// The same conditions prepare_grid_walk applies (those it adds — lane walk, tree-less path, counters — only switch the split walk off).
static bool split_may_run(const GridParams& g, size_t n_tris, size_t packets) {
  const Tuning& tn = tuning();
  if (tn.split == 0 || packets > SPLIT_MAX_PACKETS) return false;
  if (tn.split > 0) return true;
  const double real_bricks = (double)bricks_along(g.xe - g.xb, g.bl[0]) * bricks_along(g.n[1], g.bl[1]) * bricks_along(g.n[2], g.bl[2]);
  const double grid_bricks = (double)bricks_along(g.n[0], g.bl[0]) * bricks_along(g.n[1], g.bl[1]) * bricks_along(g.n[2], g.bl[2]);
  return real_bricks >= 10240.0 && n_tris >= 300000u && (double)n_tris >= 5.0 * grid_bricks;
}
// Nothing where the split walk cannot run (it was ~235 MB of every 256^3 call's block, ~370 MB from 2^19 packets on), the item lists by the
// rounds in use.
static size_t split_workspace_bytes(const GridParams& g, size_t n_tris, size_t packets) {
  if (!split_may_run(g, n_tris, packets)) return 0;
  const size_t cap = split_cap_slots(packets);
  const size_t items = std::max<size_t>(cap, std::min<size_t>(cap * SPLIT_ITEMS_PER_SLOT, 1u << 20));
  const size_t rounds = std::min(tuning().split_rounds, SPLIT_MAX_ROUNDS);
  return 256 + SPLIT_CNT_WORDS * 4 + cap * 4 + 256 + cap * 128 * 4 + 256 + rounds * items * 16 + 256;
}
// Round 6 (packet groups, one-workgroup seed flood: the walks of small problems got faster), whole one-shot calls, brute / build + walk
// (tools/exp_tiny.py, profiles/r06_tiny.txt): suzanne (968 triangles) 16^3 Raycast 0.120 / 0.125 ms, 24^3 0.151 / 0.127, Normal 24^3 0.097 / 0.132, 32^3
// 0.136 / 0.128; blob-11k 12^3 Raycast 0.190 / 0.276, 16^3 0.323 / 0.232, Normal 16^3 0.204 / 0.231, 20^3 0.347 / 0.252; blob-100k 8^3 0.40 / 1.17,
// 12^3 Raycast 1.05 / 0.94, Normal 0.66 / 0.93.  Brute force costs 0.08 ms + pairs / 1.9e11 per s with the Raycast planes (0.06 + pairs / 3e11 for
// Normal), the walk 0.12 ms + 1e-5 ms per triangle: the limits below are where they cross (rounds 2 - 5: 1e8 + 3 000 per triangle for both).
bool grid_is_tiny(const GridParams& g, size_t n_tris, int algorithm, bool raycast) {
  if (algorithm != 0 || n_tris == 0 || g.xe <= g.xb || g.n[1] == 0 || g.n[2] == 0 || g.chunk_log < 31u) return false;
  const double automatic = raycast ? 7.6e6 + 1.9e3 * (double)n_tris : 1.8e7 + 3.0e3 * (double)n_tris;
  const double limit = tuning().brute_max >= 0.0 ? tuning().brute_max : automatic;
  const double cells = (double)(g.xe - g.xb) * g.n[1] * g.n[2];
  return cells <= 4194304.0 && cells * (double)n_tris <= limit;
}
size_t grid_distance_workspace_bytes(const GridParams& g, size_t n_tris) {
  const size_t bricks = (size_t)host_brick_count(g);
  if ((double)(g.xe - g.xb) * g.n[1] * g.n[2] <= 4194304.0)   // room for k_brute_split's per-voxel words
    return bricks * 44 + bricks + 16384 + cut_blocks(g, 0) * CUT_WORDS * 4 + cut_blocks(g, 2) * CUTC_S * CUTC_WORDS * 4 + 512 + TOP_SUBTREES * 8 + 256 + bricks * 64 * 8 + 4096 + split_workspace_bytes(g, n_tris, bricks);
  const size_t trail_counters = (size_t)bricks_along(g.xe - g.xb, g.bl[0]) * (bricks_along(g.n[1], g.bl[1]) + 1) * 4;   // M2S_PEER_TRAIL progress
  return bricks * 44 + bricks + 16384 + cut_blocks(g, 0) * CUT_WORDS * 4 + cut_blocks(g, 2) * CUTC_S * CUTC_WORDS * 4 + 512 + TOP_SUBTREES * 8 + 256 + trail_counters + 1024 + split_workspace_bytes(g, n_tris, bricks);   // seeds + cut lists (one per brick) + split walk
}

__global__ __launch_bounds__(256) void k_seed_remap(uint32_t* __restrict__ ids, size_t n, const uint32_t* __restrict__ slot_of, uint32_t n_tris) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t t = ids[i];
  ids[i] = t < n_tris ? slot_of[t] : 0xffffffffu;
}

uint32_t host_packet_bricks(const GridParams& g) { return (g.xe <= g.xb || g.n[1] == 0 || g.n[2] == 0) ? 0u : host_brick_count(g); }

bool grid_walk_wants_seeds(const GridParams& g, size_t n_tris, int algorithm) {
  if (g.xe <= g.xb || g.n[1] == 0 || g.n[2] == 0) return false;
  return algorithm != 1 && n_tris && host_brick_count(g) >= 8;
}

// Seeding: every 4^3 brick starts its walk from a triangle near its own centre (jump flooding over the lattice of brick
// centres); that halves the nodes visited compared with a greedy descent.  `cen` are the triangle centroids the ids of
// the result refer to: the sorted array of the finished mesh, or the input-order array while the mesh is still being built.
int launch_grid_seeds(Arena& ws, hipStream_t st, const float4* cen, uint32_t n_tris, const GridParams& g, SeedLattice* out) {
  DeviceMesh mesh{};
  mesh.cen = cen;
  mesh.n_tris = n_tris;
  // one lattice point per packet brick, at its centre (one per 2 x 2 x 2 bricks — shift 1 — costs the headline walk 8.11 -> 9.23 ms)
  constexpr uint32_t seed_shift = 0u;
  const uint32_t stride_log[3] = {g.bl[0] + seed_shift, g.bl[1] + seed_shift, g.bl[2] + seed_shift};
  const GridParams g1 = coarse_level(g, stride_log, g.xb);
  const size_t points1 = (size_t)g1.n[0] * g1.n[1] * g1.n[2];
  uint32_t* ids = ws.take<uint32_t>(points1);
  float4* la = ws.take<float4>(points1);
  float4* lb = ws.take<float4>(points1);
  unsigned long long* keys = ws.take<unsigned long long>(points1);
  if (!ids || !la || !lb || !keys) { set_error("internal: seed workspace too small"); return M2S_ERR_HIP_INTERNAL; }
  const unsigned nb1 = (unsigned)((points1 + 255) / 256);
  if (points1 <= JFA_SMALL_MAX) {                               // the whole flood in one workgroup
    hipLaunchKernelGGL(k_jfa_small, dim3(1), dim3(JFA_SMALL_THREADS), 0, st, mesh, g1, ids);
    out->ids = ids;
    out->ny = g1.n[1];
    out->nz = g1.n[2];
    out->points = points1;
    out->shift = seed_shift;
    return 0;
  }
  M2S_HIP_CHECK(hipMemsetAsync(keys, 0xff, points1 * 8, st));
  hipLaunchKernelGGL(k_jfa_splat, dim3((mesh.n_tris + 255) / 256), dim3(256), 0, st, mesh, g1, nullptr, keys);
  // (Round 6, measured and not kept: the flood's long steps on a lattice of half the resolution + a refinement pass — the seeds get worse by
  // a few hundredths of a cell far from the surface, where a packet's candidate set grows with the square root of exactly that: the
  // headline walk 6.44 -> 7.00 ms with every step but the last two at half resolution, 6.76 -> 7.16 with only the steps >= 16 there;
  // profiles/r06_seed_coarse_*.txt.  A bound from the lanes' FINAL minima would take 1.5 % of the node tests, 12 % of the pre-tests and
  // 18 % of the exact evaluations: profiles/r06_stats2_headline.txt.)
  {
    const uint32_t maxdim = max(g1.n[0], max(g1.n[1], g1.n[2]));
    hipLaunchKernelGGL(k_jfa_load, dim3(nb1), dim3(256), 0, st, mesh, keys, points1, la);
    int step = 1;
    while ((uint32_t)step * 2 < maxdim) step *= 2;
    float4 *src = la, *dst = lb;
    for (; step >= 1; step /= 2) {
      launch_jfa_pass(st, g1, src, dst, step, nullptr);
      float4* t = src; src = dst; dst = t;
    }
    launch_jfa_pass(st, g1, src, dst, 1, ids);   // "JFA+1": one more unit pass; leaves the ids
  }
  uint32_t* dst_ids = ids;
  out->ids = dst_ids;
  out->ny = g1.n[1];
  out->nz = g1.n[2];
  out->points = points1;
  out->shift = seed_shift;
  return 0;
}

// Seeds and cut lists for the slab [g.xb, g.xe) (everything a walk needs besides the mesh); `launch_grid_walk` then
// walks the slab, or any x-piece of it that starts on a block boundary.  `raw_seeds` (optional): a seed lattice computed
// from the input-order centroids while the mesh was being built; its ids are translated to sorted slots here.
int prepare_grid_walk(Arena& ws, hipStream_t st, const DeviceMesh& mesh, const GridParams& g, int algorithm, bool pipelined,
                      GridWalkPlan* plan, const SeedLattice* raw_seeds) {
  *plan = GridWalkPlan{};
  if (g.xe <= g.xb || g.n[1] == 0 || g.n[2] == 0) return 0;
  const uint32_t packets = host_brick_count(g);
  // a one-shot call that found its problem tiny built the triangle records only (no tree: n_nodes == 0); a resident mesh decides here, by the
  // smaller (Raycast) limit: it has its tree already
  if ((mesh.n_nodes == 0 && mesh.n_tris != 0 && algorithm == 0) || grid_is_tiny(g, mesh.n_tris, algorithm, true)) {
    plan->brute_acc = ws.take<uint32_t>((size_t)packets * 64 * 2);
    if (!plan->brute_acc) { set_error("internal: brute-force workspace too small"); return M2S_ERR_HIP_INTERNAL; }
    return 0;
  }
  const bool brute = algorithm == 1;
  const uint32_t* seed1 = nullptr;
  uint32_t sh1 = 0, s1ny = 0, s1nz = 0;
  if (grid_walk_wants_seeds(g, mesh.n_tris, algorithm)) {
    SeedLattice lat;
    if (raw_seeds && raw_seeds->ids) {
      lat = *raw_seeds;
      hipLaunchKernelGGL(k_seed_remap, dim3((unsigned)((lat.points + 255) / 256)), dim3(256), 0, st, lat.ids, lat.points, mesh.slot_of, mesh.n_tris);
    } else {
      const int rc = launch_grid_seeds(ws, st, mesh.cen, mesh.n_tris, g, &lat);
      if (rc) return rc;
    }
    seed1 = lat.ids;
    s1ny = lat.ny;
    s1nz = lat.nz;
    sh1 = lat.shift;
  }
  // Walk flavour: bricks that each meet MANY triangles (triangles much smaller than voxels) are better served by
  // independent per-lane walks.  Estimate: triangles per surface brick ~ T / (6 * bricks^(2/3)).
  const int lane_env = tuning().lane_walk;   // -1 auto, 0 never, 1 always
  // measured crossover with the work-sharing lane walk (lane / packet walk, whole call, Raycast): blob-11k 32^3 0.77 / 0.86 ms, 48^3
  // 0.80 / 0.71; blob-100k 64^3 1.71 / 2.85, 96^3 2.19 / 2.12; blob-1M 128^3 8.8 / 12.1, 256^3 32.7 / 14.3: the lane walk wins while
  // there are more than ~8 triangles per brick.  (Round 4, with the packets' exact evaluations run densely — DeferQueue — the
  // crossover is ~18: blob-100k 64^3, 24 per brick, 1.60 / 2.04; 80^3, 12.5, 1.79 / 1.67; blob-11k 32^3, 22, 0.73 / 0.67.)  That is the
  // packet walk WITHOUT the split walk (below), whose launch lasts as long as its heaviest packets; where those can be split — the
  // launch must be deeper than the chip's wave slots for that — the packets win beyond 70 triangles per brick (lane walk / packets /
  // packets split, whole call, tools/exp_lane_vs_split.py, profiles/r04_lane_vs_split.txt): blob-100k 88^3 1.99 / 1.61 / 1.14 ms;
  // blob-1M 96^3 (72 per brick) 5.44 / 8.41 / 4.68, 128^3 7.53 / 9.43 / 4.95, 192^3 15.1 / 8.49 / 6.39.
  const Tuning& tn = tuning();
  const double real_bricks = (double)bricks_along(g.xe - g.xb, g.bl[0]) * bricks_along(g.n[1], g.bl[1]) * bricks_along(g.n[2], g.bl[2]);
  const double grid_bricks = (double)bricks_along(g.n[0], g.bl[0]) * bricks_along(g.n[1], g.bl[1]) * bricks_along(g.n[2], g.bl[2]);
  const bool split_possible = !brute && mesh.n_nodes != 0 && tn.split != 0 && mesh.stats == nullptr && packets <= SPLIT_MAX_PACKETS;
  // (automatic: a launch at least 1.25 x the chip's 8 192 wave slots deep, padding of the launch order not counted — the patience is measured from the time it takes to hand
  // the packets out, and a launch that is resident at once has none: blob-100k 80^3, 8 000 packets, 2.33 -> 3.77 ms, blob-11k 64^3
  // 0.54 -> 1.46)
  // With the leaf work queued and leaves of 4 - 8 triangles on coarse grids (end of round 4) the walks are two to three times shorter and
  // their tails with them, and the follow-up rounds' ~0.1 ms only pay on large meshes in coarse grids (walk without / with: blob-1M 112^3 2.80 /
  // 2.55 ms, 128^3 2.72 / 2.12, 160^3 2.54 / 2.40, 192^3 3.16 / 3.06, 256^3 4.79 / 4.98; blob-100k 96^3 0.57 / 0.55, 112^3 0.59 / 0.75, 128^3
  // 0.55 / 0.63, 160^3 0.86 / 0.80, 256^3 1.59 / 1.55; the 64-layer slabs of 512^3 x blob-1M 2.44 ... 2.99 / 2.57 ... 2.89): automatic from
  // 300 000 triangles and 5 per brick of the whole grid on.
  const bool split_auto = split_possible && tn.split < 0 && real_bricks >= 10240.0 && mesh.n_tris >= 300000u && (double)mesh.n_tris >= 5.0 * grid_bricks;
  const double lane_ratio = split_auto ? tn.lane_ratio_split : tn.lane_ratio;
  // (a tree with larger leaves — a one-shot call over a coarse grid, grid_leaf_max — is built for the packets: blob-100k 32^3, 195 triangles per
  // brick, lane walk over leaves of 2 / packets over leaves of 8: 1.68 / 1.67 ms, 48^3 1.74 / 1.18; blob-1M 48^3 5.23 / 4.83, 80^3 5.41 / 4.62)
  const bool lane_walk = !brute && mesh.n_tris && (lane_env >= 0 ? lane_env == 1 : (mesh.leaf_max <= 2u && (double)mesh.n_tris > lane_ratio * real_bricks));
  // cut lists: the top of the tree is walked once per block of 2^log bricks per axis (k_cut)
  CutList cut = {nullptr, 0, 0, 0, 0, nullptr};
  // k_cut costs about 0.25 us per brick plus a latency floor of ~0.1 ms; measured crossover (blob-100k / blob-6k,
  // tools/exp_cutmin.py): 192^3 = 110 592 packets loses 0.1-0.2 ms with the lists, 256^3 = 262 144 packets breaks even or
  // gains, 512^3 gains 1.3 ms.  The tests lower it to cover small grids.
  // (asynchronous calls are the pieces of a caller who pipelines them on two streams: k_cut then runs under the previous
  // piece's walk and pays from about half that size)
  const uint32_t cut_min_packets = tuning().cut_min_packets;   // 100 000 (round 2: 200 000 for synchronous calls; re-measured with the 64-byte lists: 224^3 2.88 -> 2.75 ms, 192^3 2.41 -> 2.38, 160^3 2.04 -> 2.07)
  if (!brute && !lane_walk && seed1 != nullptr && packets >= cut_min_packets) {
    // emission radius of a list entry: emit_near brick radii next to the surface, emit_far of the distance far from it
    const float emit_near = tuning().cut_near;
    const float emit_far = tuning().cut_far;   // 1/32: re-tuned at the end of round 3 (1/16 before): headline 9.19 -> 9.11 ms, 1024^3 x sheet-100k 92.95 -> 89.33 ms
    // A wave that has visited this many nodes lets its bricks emit whatever they meet next: the long union walks of the
    // regions with many near-ties (deep inside a round body) are the tail of the launch — on the 64-layer slab of an 8-GPU
    // rank, 4 waves per SIMD, they WERE its duration (0.39 -> 0.19 ms; 512^3: flat between 300 and 450, 200 costs the
    // packets 0.5 ms) — and what they still decide so deep in the tree the packets decide almost as cheaply.
    uint32_t depth = 1;
    while ((1ull << depth) < (unsigned long long)mesh.n_tris + 1ull) ++depth;
    const uint32_t wave_cap = tuning().cut_wave_cap ? tuning().cut_wave_cap : std::max(120u, 20u * depth);
    const uint32_t budget = 100000u;
    const uint32_t nbx = bricks_along(g.xe - g.xb, g.bl[0]), nby = bricks_along(g.n[1], g.bl[1]), nbz = bricks_along(g.n[2], g.bl[2]);
    const size_t bricks = (size_t)nbx * nby * nbz;
    uint32_t* lists = ws.take<uint32_t>(bricks * CUT_WORDS);
    if (!lists) { set_error("internal: cut-list workspace too small"); return M2S_ERR_HIP_INTERNAL; }
    const uint32_t cbx = bricks_along(nbx, 2), cby = bricks_along(nby, 2), cbz = bricks_along(nbz, 2);   // blocks of 4 x 4 x 4 bricks = fine waves
    const size_t waves = (size_t)cbx * cby * cbz;
    // two levels (k_cut LEVEL 1 + 2; M2S_CUT_COARSE: -1 automatic, 0 never, 1 always).  The coarse launch is waves / 8 waves of its own whose longest
    // chain (the subtree next to its region, up to the visit cap) is ~0.14 ms whatever the grid: 1024^3 (262 144 fine waves) seed + cut 5.98 -> 4.9 ms and
    // the call 83.2 -> 81.9 ms (sheet-100k, Normal) / 40.8 -> 39.9 (blob-100k); 512^3 (32 768) 0.72 -> 0.67 + 0.14 ms with a walk 0.08 ms shorter — a wash;
    // a 64-layer slab of 512^3 (4 096) 0.20 -> 0.34 ms.  Automatic from M2S_CUT_COARSE_MIN_WAVES = 40 000 fine waves on (profiles/r06_cut_coarse.txt; the
    // tests force it on small grids).  A block must lie inside one chunk of an interleaved slab.
    const int cc = tuning().cut_coarse;
    const bool blocks_ok = sh1 == 0u && (g.chunk_log >= 31u || g.chunk_log >= g.bl[0] + 2u);
    const bool two_level = blocks_ok && (cc > 0 || (cc < 0 && waves >= tuning().cut_coarse_min_waves));
    if (two_level) {
      uint32_t* coarse = ws.take<uint32_t>(waves * CUTC_S * CUTC_WORDS);
      if (!coarse) { set_error("internal: cut-list workspace too small"); return M2S_ERR_HIP_INTERNAL; }
      const size_t groups = (size_t)bricks_along(cbx, 2) * bricks_along(cby, 2) * bricks_along(cbz, 2);
      const uint32_t coarse_cap = tuning().cut_coarse_cap ? tuning().cut_coarse_cap : wave_cap;
      hipLaunchKernelGGL((k_cut<true, 1>), dim3((unsigned)(groups * CUTC_S)), dim3(64), 0, st, mesh, g, seed1, sh1, s1ny, s1nz, cbx, cby, cbz, coarse, 1.0f, emit_far, budget,
                         coarse_cap, (const float4*)nullptr, (const uint32_t*)nullptr, (const GridParams*)nullptr, (const uint32_t*)nullptr);
      hipLaunchKernelGGL((k_cut<true, 2>), dim3((unsigned)waves), dim3(64), 0, st, mesh, g, seed1, sh1, s1ny, s1nz, nbx, nby, nbz, lists, emit_near, emit_far, budget, wave_cap,
                         (const float4*)nullptr, (const uint32_t*)nullptr, (const GridParams*)nullptr, (const uint32_t*)coarse);
    } else
    hipLaunchKernelGGL((k_cut<true, 0>), dim3((unsigned)waves), dim3(64), 0, st, mesh, g, seed1, sh1, s1ny, s1nz, nbx, nby, nbz, lists, emit_near, emit_far, budget, wave_cap,
                       (const float4*)nullptr, (const uint32_t*)nullptr, (const GridParams*)nullptr, (const uint32_t*)nullptr);
    cut = {lists, 0, nby, nbz, 0, nullptr};
  }
  // packet groups (k_packet_group; M2S_GROUP: -1 automatic, 0 never, 1 always): launches of at most M2S_GROUP_MAX_PACKETS packets without
  // cut lists where the bricks meet several triangles each (the chains are long there: 64^3 ... 128^3 over 100 k triangles)
  {
    const int gk = tn.group;
    // (not where the split walk is the automatic choice — large meshes in launches deeper than the chip: blob-1M 96^3 Raycast 1.88 ms split, 2.48 in groups)
    const bool can = !brute && !lane_walk && cut.lists == nullptr && mesh.n_nodes != 0 && mesh.stats == nullptr && tn.split <= 0 && !split_auto && tn.defer < 0 && g.chunk_log >= 31u;
    const bool want = gk > 0 || (gk < 0 && (double)mesh.n_tris >= tn.group_min_ratio * real_bricks);
    if (can && want) {
      // as many waves per packet (a power of two, four at most) as keep the launch within M2S_GROUP_TARGET_WAVES waves
      uint32_t w = 1;
      while (w < GROUP_MAX_WAVES && (double)(2u * w) * real_bricks <= (double)tn.group_target_waves) w *= 2;
      if (gk > 0 && w < 2u) w = GROUP_MAX_WAVES;                      // forced (tests)
      if (w >= 2u) {
        uint2* top = ws.take<uint2>(TOP_SUBTREES);
        if (!top) { set_error("internal: workspace too small"); return M2S_ERR_HIP_INTERNAL; }
        hipLaunchKernelGGL(k_tree_top, dim3(1), dim3(TOP_SUBTREES), 0, st, mesh, top);
        plan->group_top = top;
        plan->group_waves = w;
      }
    }
  }
  plan->seeds = seed1; plan->seed_shift = sh1; plan->seed_ny = s1ny; plan->seed_nz = s1nz;
  plan->cut_lists = cut.lists; plan->cut_log = cut.log; plan->cut_ny = cut.ny; plan->cut_nz = cut.nz;
  plan->lane_walk = lane_walk;
  // split walk (packet walk only; M2S_SPLIT: -1 / 1 on, 0 off, 2 on with the flags raised from the start)
  // Where it pays (tools/exp_split.py, exp_split_rank.py; walk with / without): the stragglers are the packets deep inside a body
  // whose voxels see many triangles at (nearly) the same distance, and they weigh the more the finer the mesh is against the grid —
  // blob-100k in 96^3 ... 192^3 1.79 -> 1.07, 1.57 -> 1.28, 1.81 -> 1.63 ms, in 256^3 2.25 -> 2.34 (a wash), the 64-layer slabs of
  // 512^3 1.21 -> 1.26 (a loss: no tail to speak of, three more launches); blob-1M in 256^3 12.8 -> 9.5 ms, its slowest 8-GPU slab of
  // 512^3 5.05 -> 4.05, its fastest 3.14 -> 3.16.  Automatic: split_auto above (>= 300 000 triangles, >= 5 per packet brick of the WHOLE grid, >= 10 240 bricks).
  if (!lane_walk && split_possible && plan->group_top == nullptr && (tn.split > 0 || split_auto)) {
    SplitCtl sc;
    sc.cap_slots = split_cap_slots(packets);
    sc.cap_items = std::max(sc.cap_slots, std::min(sc.cap_slots * SPLIT_ITEMS_PER_SLOT, 1u << 20));
    sc.rounds = std::min(tn.split_rounds, SPLIT_MAX_ROUNDS);
    sc.emit_min = std::max(2u, tn.split_min_records) * (uint32_t)sizeof(NodeExt);
    sc.emit_max = std::max(std::max(2u, tn.split_min_records), tn.split_max_records) * (uint32_t)sizeof(NodeExt);
    sc.grace = tn.split_budget ? tn.split_budget : 128u;
    sc.idle_below = tn.split == 2 ? 1u : 0u;                               // forced (tests): every stamp is time 0 — all flags up, no patience
    // patience, in ordinary packet times: the launch is packets / slots rounds of the chip's 8 192 wave slots deep, the flag goes up
    // when all but the last round have been handed out
    const double rounds_before = std::max(1.0, (double)packets / 8192.0 - 1.0);
    sc.patience_q8 = (uint32_t)std::min(65535.0 * 256.0, 256.0 * tn.split_patience / rounds_before);
    sc.cnt = ws.take<uint32_t>(SPLIT_CNT_WORDS);
    sc.slot_packet = ws.take<uint32_t>(sc.cap_slots);
    sc.acc = ws.take<uint32_t>((size_t)sc.cap_slots * 128);
    sc.items = ws.take<uint4>((size_t)sc.rounds * sc.cap_items);
    if (!sc.cnt || !sc.slot_packet || !sc.acc || !sc.items) { set_error("internal: split-walk workspace too small"); return M2S_ERR_HIP_INTERNAL; }
    plan->split = sc;
    plan->split_forced = tn.split == 2;
  }
  // The packet walk's leaf work (DeferQueue; M2S_DEFER forces a form): 0 wave-wide at once, 1 the exact evaluations queued per (voxel, triangle) pair,
  // 2: queued, but at once where most of the wave is reached (grids much finer than the mesh: see DEFER_DIRECT_LANES),
  // 3: the pre-tests queued too (defer_pretest) — from 0.045 triangles per brick on: walk, queued evaluations / + queued pre-tests, blob-100k
  // 128^3 1.02 / 0.83 ms, 256^3 1.74 / 1.48, 512^3 (0.048 per brick) 6.85 / 6.54, 768^3 (0.014) 16.9 / 17.7; blob-1M 512^3 21.9 / 18.2; blob-11k
  // 256^3 (0.04) 0.60 / 0.63; sheet-100k 512^3 (0.05) 13.2 / 12.0.  (Both forms of pre-test in one kernel, chosen per leaf by the number of lanes that
  // want it, cost the dense regime what they gained the sparse one: 128^3 0.83 -> 0.92, 768^3 17.7 -> 16.9.)
  plan->defer = tn.defer == 0 ? 0 : tn.defer > 0 ? tn.defer : ((double)mesh.n_tris < 0.02 * grid_bricks ? 2 : (double)mesh.n_tris < 0.045 * grid_bricks ? 1 : 3);
  return 0;
}

int launch_push_cells(hipStream_t st, const float* src, const PeerOut& peers, uint64_t first, uint64_t count) {
  if (peers.n == 0 || count == 0) return 0;
  // a bandwidth-bound copy next to the walk of the following piece: enough workgroups to keep every xGMI link busy,
  // few enough to leave the CUs to the walk (M2S_PUSH_BLOCKS)
  const unsigned max_blocks = tuning().push_blocks ? tuning().push_blocks : 256u;
  const uint64_t want = (count / 4 + 255) / 256 + 1;
  const unsigned blocks = (unsigned)std::min<uint64_t>(max_blocks, want);
  hipLaunchKernelGGL(k_push_cells, dim3(blocks), dim3(256), 0, st, src, peers, first, count);
  M2S_HIP_CHECK(hipGetLastError());
  return 0;
}

uint32_t trail_unit_log(const GridParams& g) {
  (void)g;
  return 2u;   // 4 bricks = 16 layers of a 4^3-brick grid (16 MB per peer and unit at 512^2 rows): 2-brick units stream finer but
               // their packet order (super-bricks 2 bricks wide) costs the walk 20 % of its locality
}
uint32_t trail_units(const GridParams& g) {
  const uint32_t nbx = bricks_along(g.xe - g.xb, g.bl[0]), ul = trail_unit_log(g);
  return (nbx + (1u << ul) - 1u) >> ul;
}
uint32_t trail_rows(const GridParams& g) { return bricks_along(g.n[1], g.bl[1]); }
int launch_push_trailing(hipStream_t st, const float* src, const PeerOut& peers, const GridParams& g, int* d_err) {
  if (peers.n == 0 || g.xe <= g.xb) return 0;
  const unsigned blocks = tuning().push_blocks ? tuning().push_blocks : 64u;
  const uint64_t row = (uint64_t)g.n[1] * g.n[2];
  hipLaunchKernelGGL(k_push_trailing, dim3(blocks), dim3(256), 0, st, src, peers, (uint64_t)g.xb * row - g.out_off, row, g.xe - g.xb,
                     (1u << g.bl[0]) << peers.unit_log, trail_units(g), d_err);
  M2S_HIP_CHECK(hipGetLastError());
  return 0;
}

int launch_grid_walk(hipStream_t st, const DeviceMesh& mesh, const GridParams& g, int mode, const uint32_t* d_inside_plane,
                     int algorithm, const GridWalkPlan& plan, uint32_t bx_off, float* d_out, int* d_err, const PeerOut* peers) {
  if (g.xe <= g.xb || g.n[1] == 0 || g.n[2] == 0) return 0;
  PeerOut pz{};
  if (peers) pz = *peers;
  const uint32_t packets = host_brick_count(g);
  const bool brute = algorithm == 1;
  const uint32_t* seed1 = plan.seeds;
  const uint32_t sh1 = plan.seed_shift, s1ny = plan.seed_ny, s1nz = plan.seed_nz;
  const CutList cut = {plan.cut_lists, plan.cut_log, plan.cut_ny, plan.cut_nz, bx_off, nullptr};
  if (plan.brute_acc != nullptr) {
    // ~4 blocks per CU over (voxel blocks x triangle chunks); a chunk is a whole number of 128-triangle tiles
    const uint32_t real = bricks_along(g.xe - g.xb, g.bl[0]) * bricks_along(g.n[1], g.bl[1]) * bricks_along(g.n[2], g.bl[2]);   // no super-brick padding here
    const uint32_t vblocks = (real + 3) / 4, tiles = (mesh.n_tris + TILE - 1) / TILE;
    const uint32_t chunks = std::max(1u, std::min(tiles, (1024u + vblocks - 1) / vblocks));
    const uint32_t tiles_per_chunk = (tiles + chunks - 1) / chunks;
    const uint32_t ychunks = (tiles + tiles_per_chunk - 1) / tiles_per_chunk;
    M2S_HIP_CHECK(hipMemsetD32Async((hipDeviceptr_t)plan.brute_acc, 0x7f800000, (size_t)packets * 64 * 2, st));   // +inf: where every search starts (Best<>)
    if (mode == MODE_UNSIGNED) {
      hipLaunchKernelGGL((k_brute_split<MODE_UNSIGNED>), dim3(vblocks, ychunks), dim3(256), 0, st, mesh, g, plan.brute_acc, d_err, real, tiles_per_chunk);
      if (d_inside_plane) hipLaunchKernelGGL((k_brute_finish<MODE_UNSIGNED, SIGN_GRID_PLANE>), dim3(vblocks), dim3(256), 0, st, g, d_inside_plane, (const uint32_t*)plan.brute_acc, d_out, real, pz);
      else hipLaunchKernelGGL((k_brute_finish<MODE_UNSIGNED, SIGN_NONE>), dim3(vblocks), dim3(256), 0, st, g, (const uint32_t*)nullptr, (const uint32_t*)plan.brute_acc, d_out, real, pz);
    } else {
      hipLaunchKernelGGL((k_brute_split<MODE_NORMAL_FOLD>), dim3(vblocks, ychunks), dim3(256), 0, st, mesh, g, plan.brute_acc, d_err, real, tiles_per_chunk);
      hipLaunchKernelGGL((k_brute_finish<MODE_NORMAL_FOLD, SIGN_NONE>), dim3(vblocks), dim3(256), 0, st, g, (const uint32_t*)nullptr, (const uint32_t*)plan.brute_acc, d_out, real, pz);
    }
    M2S_HIP_CHECK(hipGetLastError());
    return 0;
  }
  if (plan.lane_walk) {
    const unsigned blocks = (packets + 3) / 4;
    if (mode == MODE_UNSIGNED && d_inside_plane)
      hipLaunchKernelGGL((k_lane<MODE_UNSIGNED, SIGN_GRID_PLANE>), dim3(blocks), dim3(256), 0, st, mesh, g, d_inside_plane, d_out, d_err, packets, seed1, sh1, s1ny, s1nz, bx_off, pz);
    else if (mode == MODE_UNSIGNED)
      hipLaunchKernelGGL((k_lane<MODE_UNSIGNED, SIGN_NONE>), dim3(blocks), dim3(256), 0, st, mesh, g, nullptr, d_out, d_err, packets, seed1, sh1, s1ny, s1nz, bx_off, pz);
    else
      hipLaunchKernelGGL((k_lane<MODE_NORMAL_FOLD, SIGN_NONE>), dim3(blocks), dim3(256), 0, st, mesh, g, nullptr, d_out, d_err, packets, seed1, sh1, s1ny, s1nz, bx_off, pz);
    M2S_HIP_CHECK(hipGetLastError());
    return 0;
  }
  if (plan.group_top != nullptr && !brute) {
    const uint32_t per = 8u << XCD_RUN_LOG;
    const uint32_t grid_blocks = ((packets + per - 1) / per) * per;        // as launch_packet: whole runs per XCD (xcd_remap)
    const uint32_t gw = plan.group_waves;
    const size_t lds_u = ((size_t)gw * 256u + DeferLayout<MODE_UNSIGNED>::SLOT_WORDS) * 4u, lds_n = ((size_t)gw * 256u + DeferLayout<MODE_NORMAL_FOLD>::SLOT_WORDS) * 4u;
    if (mode == MODE_UNSIGNED && d_inside_plane)
      hipLaunchKernelGGL((k_packet_group<MODE_UNSIGNED, SIGN_GRID_PLANE>), dim3(grid_blocks), dim3(64 * gw), lds_u, st, mesh, g, d_inside_plane, d_out, d_err, packets,
                         seed1, sh1, s1ny, s1nz, bx_off, plan.group_top, pz);
    else if (mode == MODE_UNSIGNED)
      hipLaunchKernelGGL((k_packet_group<MODE_UNSIGNED, SIGN_NONE>), dim3(grid_blocks), dim3(64 * gw), lds_u, st, mesh, g, (const uint32_t*)nullptr, d_out, d_err, packets,
                         seed1, sh1, s1ny, s1nz, bx_off, plan.group_top, pz);
    else
      hipLaunchKernelGGL((k_packet_group<MODE_NORMAL_FOLD, SIGN_NONE>), dim3(grid_blocks), dim3(64 * gw), lds_n, st, mesh, g, (const uint32_t*)nullptr, d_out, d_err, packets,
                         seed1, sh1, s1ny, s1nz, bx_off, plan.group_top, pz);
    M2S_HIP_CHECK(hipGetLastError());
    return 0;
  }
  // (the Normal fold's split variants — k_packet<GRID, NORMAL_FOLD, ..., SPLIT, {1, 3}>, k_split_round<NORMAL_FOLD> — need 9 - 11 registers
  // more than eight waves per SIMD leave and spill them to scratch, the very thing that costs this kernel 10 - 40 %: the automatic choice
  // leaves the Normal sign to the plain walk; M2S_SPLIT=1 / 2 still runs them — the tests do)
  const bool split_ok = mode != MODE_NORMAL_FOLD || tuning().split > 0;
  SplitCtl piece_split = plan.split;
  if (piece_split.cnt != nullptr && !plan.split_forced) {
    // the patience is measured in hand-out rounds of THIS launch (an x-piece of the slab is shallower than the slab the plan was made for)
    const double rounds_before = std::max(1.0, (double)packets / 8192.0 - 1.0);
    piece_split.patience_q8 = (uint32_t)std::min(65535.0 * 256.0, 256.0 * tuning().split_patience / rounds_before);
  }
  const SplitCtl* split = (!brute && plan.split.cnt != nullptr && mesh.stats == nullptr && split_ok) ? &piece_split : nullptr;
  if (split) hipLaunchKernelGGL(k_split_init, dim3(1), dim3(64), 0, st, split->cnt, plan.split_forced ? 1u : 0u);
  if (mode == MODE_UNSIGNED && d_inside_plane) {
    if (brute) launch_brute<true, MODE_UNSIGNED, SIGN_GRID_PLANE>(st, mesh, g, nullptr, 0, d_inside_plane, d_out, d_err, packets, peers);
    else launch_packet<true, MODE_UNSIGNED, SIGN_GRID_PLANE>(st, mesh, g, nullptr, nullptr, 0, d_inside_plane, d_out, d_err, packets, seed1, sh1, s1ny, s1nz, nullptr, cut, peers, split, plan.defer);
    if (split) launch_split_rounds<MODE_UNSIGNED, SIGN_GRID_PLANE>(st, mesh, g, d_inside_plane, d_out, d_err, *split, cut, peers);
  } else if (mode == MODE_UNSIGNED) {
    if (brute) launch_brute<true, MODE_UNSIGNED, SIGN_NONE>(st, mesh, g, nullptr, 0, nullptr, d_out, d_err, packets, peers);
    else launch_packet<true, MODE_UNSIGNED, SIGN_NONE>(st, mesh, g, nullptr, nullptr, 0, nullptr, d_out, d_err, packets, seed1, sh1, s1ny, s1nz, nullptr, cut, peers, split, plan.defer);
    if (split) launch_split_rounds<MODE_UNSIGNED, SIGN_NONE>(st, mesh, g, nullptr, d_out, d_err, *split, cut, peers);
  } else {
    if (brute) launch_brute<true, MODE_NORMAL_FOLD, SIGN_NONE>(st, mesh, g, nullptr, 0, nullptr, d_out, d_err, packets, peers);
    else launch_packet<true, MODE_NORMAL_FOLD, SIGN_NONE>(st, mesh, g, nullptr, nullptr, 0, nullptr, d_out, d_err, packets, seed1, sh1, s1ny, s1nz, nullptr, cut, peers, split, plan.defer);
    if (split) launch_split_rounds<MODE_NORMAL_FOLD, SIGN_NONE>(st, mesh, g, nullptr, d_out, d_err, *split, cut, peers);
  }
  M2S_HIP_CHECK(hipGetLastError());
  if (split && tuning().split_report) {
    uint32_t h[SPLIT_CNT_WORDS];
    M2S_HIP_CHECK(hipStreamSynchronize(st));
    M2S_HIP_CHECK(hipMemcpy(h, split->cnt, sizeof(h), hipMemcpyDeviceToHost));
    fprintf(stderr, "[m2s split] %u packets: %u suspended (room for %u); items per round:", packets, h[0], split->cap_slots);
    for (uint32_t r = 1; r <= split->rounds; ++r) fprintf(stderr, " %u", h[1 + r]);
    fprintf(stderr, " (room for %u each, reserved in blocks of 64); first look after %u units, patience %.2f of the time to the flag, subtrees of %u .. %u records; XCD 0: handed out in %.1f us\n",
            split->cap_items, split->grace, split->patience_q8 / 256.0, split->emit_min / (uint32_t)sizeof(NodeExt), split->emit_max / (uint32_t)sizeof(NodeExt),
            h[16] ? ((h[16] & ~1u) - h[24]) * 0.01 : 0.0);
  }
  return 0;
}

int launch_grid_distance(Arena& ws, hipStream_t st, const DeviceMesh& mesh, const GridParams& g, int mode,
                         const uint32_t* d_inside_plane, int algorithm, float* d_out, int* d_err,
                         hipEvent_t ev_before_final, hipEvent_t wait_before_final, bool pipelined,
                         const SeedLattice* raw_seeds, hipEvent_t wait_raw_seeds, const PeerOut* peers) {
  GridWalkPlan plan;
  if (raw_seeds && wait_raw_seeds) M2S_HIP_CHECK(hipStreamWaitEvent(st, wait_raw_seeds, 0));   // computed on another stream
  int rc = prepare_grid_walk(ws, st, mesh, g, algorithm, pipelined, &plan, raw_seeds);
  if (rc) return rc;
  // the sign planes may have been built beside the seed passes, on another stream (capi.hip): the walk needs them
  if (wait_before_final) M2S_HIP_CHECK(hipStreamWaitEvent(st, wait_before_final, 0));
  if (ev_before_final) M2S_HIP_CHECK(hipEventRecord(ev_before_final, st));
  return launch_grid_walk(st, mesh, g, mode, d_inside_plane, algorithm, plan, 0, d_out, d_err, peers);
}

// Test hook (capi.hip m2s_debug_cut_code): the list word k_cut writes for the range [start, start + len) of a tree of n_nodes records,
// and the (first, end) records k_packet reads back from it.
void cut_word_roundtrip(uint32_t n_nodes, uint32_t start, uint32_t len, uint32_t* word, uint32_t* first, uint32_t* end) {
  const uint32_t S = cut_start_bits(n_nodes);
  const uint32_t w = start | (cut_encode_len(len, 27u - S) << S);
  *word = w;
  const uint32_t f = w & ((1u << S) - 1u);
  const uint32_t l = ((w >> S) & ((1u << (27u - S)) - 1u)) << (w >> 27);                // as k_packet decodes it
  *first = f;
  *end = std::min(f + l, n_nodes);
}

// Small query sets take k_brute_split_q: queries x triangles <= 1.2e8, 6e7 with the three ray tests per pair (M2S_BRUTE_MAX overrides; 0:
// never).  Measured (tools/exp_small_queries.py, whole one-shot calls, all pairs / build + walk): 11 k triangles x 1 ... 1 000 queries
// 0.085 - 0.15 / 0.25 - 0.70 ms, x 10 000 0.47 (0.83 with rays) / 0.55 (0.66); 100 k triangles x 64 0.13 - 0.22 / 1.0 - 1.3 ms, x 1 000
// 0.46 (0.81) / 0.74 (0.94), x 10 000 3.2 (5.8) / 0.9 (1.0): 240 G pairs/s for the distance alone, 135 G with the rays.
// Sparse query sets take the lane walk (k_lane_q): fewer than M2S_QUERY_LANE_COEFF (2.5) queries per triangle.
bool query_walk_is_lane(size_t n_q, size_t n_tris, int sign_src) {
  const int lane_env = tuning().lane_walk;   // -1 auto, 0 never, 1 always
  return n_tris && sign_src != SIGN_XRAY_ALL && (lane_env >= 0 ? lane_env == 1 : (double)n_q < tuning().query_lane_coeff * (double)n_tris);
}
// The leaf size a query call wants (M2S_LEAF_MAX overrides): 2 for the lane walk (every lane pays for its own leaf), for the packets by
// queries per triangle — packets of few queries per triangle are large against the triangles, as bricks of a coarse grid are (packets, leaves of
// 2 / 4 / 8, RtreeBvh, whole call, tools/exp_query_walks.py): blob-100k 1 M queries 1.91 / 1.56 / 1.39 ms, 3 M 2.37 / 2.02 / 1.91, 10 M 3.93 / 3.60 / 3.74;
// blob-11k 300 k 0.66 / 0.63 / 0.63, 3 M 1.00 / 0.95 / 1.04, 10 M 2.19 / 2.30 / 2.77.
uint32_t query_leaf_max(size_t n_q, size_t n_tris, int sign_src) {
  const Tuning& tn = tuning();
  if (tn.leaf_max != 0) return tn.leaf_max;
  if (n_tris == 0 || query_walk_is_lane(n_q, n_tris, sign_src)) return 2u;
  const double per_tri = (double)n_q / (double)n_tris;
  return per_tri < 50.0 ? 8u : per_tri < 500.0 ? 4u : 2u;
}

// The leaf size of a one-shot grid call's tree (M2S_LEAF_MAX overrides).  A larger leaf trades node tests for leaf pre-tests; with
// those queued per (voxel, triangle) pair the optimum moved up wherever a brick meets more than a triangle or so (walk, leaves of 2 / 4 /
// 8, tools/exp_lane_vs_split.py with M2S_LEAF_MAX): blob-100k 64^3 (24 triangles per brick) 1.15 / 0.79 / 0.59 ms, 96^3 (7.2, split) 0.69 /
// 0.57 / 0.54, 128^3 (3.05) 0.87 / 0.74 / -; blob-1M 128^3 (30, split) 2.89 / 2.33 / 2.11, 256^3 (3.8) 6.10 / 5.22 / -; blob-11k 64^3 (2.7) 0.25 /
// 0.21 / 0.20, 96^3 (0.8) 0.22 / 0.20 / 0.21; blob-100k 256^3 (0.38) 1.55 / 1.52 / -, 512^3 (0.048) 6.48 / 7.88 / -.  A persistent mesh's tree is
// re-marked by the grid call that wants another size (set_leaf_size).
uint32_t grid_leaf_max(const GridParams& g, size_t n_tris) {
  const Tuning& tn = tuning();
  if (tn.leaf_max != 0) return tn.leaf_max;
  const double bricks = (double)bricks_along(g.n[0], g.bl[0]) * bricks_along(g.n[1], g.bl[1]) * bricks_along(g.n[2], g.bl[2]);
  const double per_brick = (double)n_tris / std::max(1.0, bricks);
  return per_brick >= 40.0 ? 16u : per_brick >= 3.0 ? 8u : per_brick >= 0.6 ? 4u : 2u;   // (16: blob-100k 32^3, 195 per brick, 1.31 -> 1.11 ms; blob-1M 80^3 3.14 -> 2.53)
}

bool query_is_tiny(size_t n_q, size_t n_tris, int algorithm, int sign_src) {
  if (algorithm != 0 || n_tris == 0 || n_q == 0 || sign_src == SIGN_XRAY_ALL) return false;
  const double limit = tuning().brute_max >= 0.0 ? tuning().brute_max : (sign_src == SIGN_RAYS3 ? 6.0e7 : 1.2e8);
  return (double)n_q * (double)n_tris <= limit;
}
int launch_query_brute_split(Arena& ws, hipStream_t st, const DeviceMesh& mesh, const float* d_queries, size_t n_q, int mode, int sign_src,
                             float* d_out, int* d_err) {
  const uint32_t nq = (uint32_t)n_q;
  uint32_t* acc = ws.take<uint32_t>(4 * n_q);
  if (!acc) { set_error("internal: query workspace too small"); return M2S_ERR_HIP_INTERNAL; }
  const uint32_t qblocks = (nq + 255u) / 256u, tiles = (mesh.n_tris + TILE - 1) / TILE;
  // ~4 blocks per CU over (query blocks x triangle chunks); a chunk is a whole number of 128-triangle tiles
  const uint32_t chunks = std::max(1u, std::min(tiles, (1024u + qblocks - 1) / qblocks));
  const uint32_t tiles_per_chunk = (tiles + chunks - 1) / chunks, ychunks = (tiles + tiles_per_chunk - 1) / tiles_per_chunk;
  const dim3 grid(qblocks, ychunks);
  hipLaunchKernelGGL(k_brute_q_init, dim3(qblocks), dim3(256), 0, st, acc, nq, mode == MODE_NORMAL_FOLD ? 0x7f800000u : 0u);
  if (mode == MODE_UNSIGNED && sign_src == SIGN_RAYS3) {
    hipLaunchKernelGGL((k_brute_split_q<MODE_UNSIGNED, SIGN_RAYS3>), grid, dim3(256), 0, st, mesh, d_queries, nq, acc, d_err, tiles_per_chunk);
    hipLaunchKernelGGL((k_brute_finish_q<MODE_UNSIGNED, SIGN_RAYS3>), dim3(qblocks), dim3(256), 0, st, (const uint32_t*)acc, nq, d_out);
  } else if (mode == MODE_UNSIGNED) {
    hipLaunchKernelGGL((k_brute_split_q<MODE_UNSIGNED, SIGN_NONE>), grid, dim3(256), 0, st, mesh, d_queries, nq, acc, d_err, tiles_per_chunk);
    hipLaunchKernelGGL((k_brute_finish_q<MODE_UNSIGNED, SIGN_NONE>), dim3(qblocks), dim3(256), 0, st, (const uint32_t*)acc, nq, d_out);
  } else if (mode == MODE_NORMAL_FOLD) {
    hipLaunchKernelGGL((k_brute_split_q<MODE_NORMAL_FOLD, SIGN_NONE>), grid, dim3(256), 0, st, mesh, d_queries, nq, acc, d_err, tiles_per_chunk);
    hipLaunchKernelGGL((k_brute_finish_q<MODE_NORMAL_FOLD, SIGN_NONE>), dim3(qblocks), dim3(256), 0, st, (const uint32_t*)acc, nq, d_out);
  } else {
    hipLaunchKernelGGL((k_brute_split_q<MODE_NEAREST_NORMAL, SIGN_NONE>), grid, dim3(256), 0, st, mesh, d_queries, nq, acc, d_err, tiles_per_chunk);
    hipLaunchKernelGGL((k_brute_finish_q<MODE_NEAREST_NORMAL, SIGN_NONE>), dim3(qblocks), dim3(256), 0, st, (const uint32_t*)acc, nq, d_out);
  }
  M2S_HIP_CHECK(hipGetLastError());
  return 0;
}

size_t query_workspace_bytes(size_t n_q) {
  size_t n = n_q ? n_q : 1, tmp = 0;
  (void)sort_pairs_u32(nullptr, tmp, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const uint32_t*)nullptr, (uint32_t*)nullptr,
                            n, 0, 30, (hipStream_t)0);
  size_t sel = 0;
  (void)select_flagged_indices(nullptr, sel, (const uint8_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, n, (hipStream_t)0);
  return n * (8 + 8 + 4 + 4 + 16 + 1 + 4) + n * 16 + 256 + (n / 32 + 64) * (16 + 4 * CUT_WORDS) + tmp + sel + 21 * 256 + (size_t)64 * 64 * 64 * 44 + 8192 + 24 * 1024 + 256;
}

// Generic queries in two parts.  prepare_query_walk needs the queries only — bounding box, Morton keys, sort, packet table, the
// packets' centres and the gather into sorted order — so a one-shot call runs it on a side stream BESIDE the LBVH build (capi.hip:
// 0.8 ms of bandwidth-bound passes next to 0.24 ms of latency-bound launches for 10 M queries x 100 k triangles); launch_query_walk
// needs the tree: seed lattice, cut lists, walk.  launch_query_distance is both on one stream (persistent meshes, asynchronous calls).
int prepare_query_walk(Arena& ws, hipStream_t st, const float* d_queries, size_t n_q, size_t n_tris, int sign_src, int algorithm, QueryPlan* plan,
                       hipEvent_t after_lattice) {
  *plan = QueryPlan{};
  plan->n_q = n_q;
  if (n_q == 0 || algorithm == 1) return 0;
  const uint32_t nq = (uint32_t)n_q;
  const uint32_t packets = (nq + 63) / 64;
  // Morton order
  int* qb = ws.take<int>(8 + 6 * QB_BLOCKS);
  uint32_t* keys = ws.take<uint32_t>(n_q);
  uint32_t* keys2 = ws.take<uint32_t>(n_q);
  uint32_t* vals = ws.take<uint32_t>(n_q);
  uint32_t* perm = ws.take<uint32_t>(n_q);
  float4* sorted = ws.take<float4>(n_q);
  size_t tmp_bytes = 0;
  (void)sort_pairs_u32(nullptr, tmp_bytes, keys, keys2, vals, perm, n_q, 0, QKEY_BITS, st);
  void* tmp = ws.take<char>(tmp_bytes ? tmp_bytes : 1);
  if (!qb || !keys || !keys2 || !vals || !perm || !sorted || !tmp) {
    set_error("internal: query workspace too small");
    return M2S_ERR_HIP_INTERNAL;
  }
  // key bits that matter: cells of ~8 queries at the finest level, whole Morton triples, 12 ... 30
  uint32_t bits = 12;
  while (bits < (uint32_t)QKEY_BITS && (1ull << bits) * 8ull < (unsigned long long)n_q) bits += 3;
  const uint32_t drop = (uint32_t)QKEY_BITS - bits;
  const unsigned B = 256, nb = (nq + B - 1) / B;
  const unsigned qblocks = nb < QB_BLOCKS ? nb : QB_BLOCKS;
  hipLaunchKernelGGL(k_qbounds, dim3(qblocks), dim3(B), 0, st, d_queries, nq, qb + 8);
  hipLaunchKernelGGL(k_qbounds_final, dim3(1), dim3(B), 0, st, qb + 8, qblocks, qb);
  const bool seeds = n_tris && packets >= 8;
  if (seeds) {                                   // the seed lattice's description: QL^3 cells over the queries' bounding box
    GridParams* lat = ws.take<GridParams>(1);
    if (!lat) { set_error("internal: query workspace too small"); return M2S_ERR_HIP_INTERNAL; }
    hipLaunchKernelGGL(k_qlattice, dim3(1), dim3(64), 0, st, qb, lat);
    plan->lat = lat;
  }
  if (after_lattice) M2S_HIP_CHECK(hipEventRecord(after_lattice, st));
  hipLaunchKernelGGL(k_qkeys, dim3(nb), dim3(B), 0, st, d_queries, nq, qb, keys, vals, drop);
  M2S_HIP_CHECK(sort_pairs_u32(tmp, tmp_bytes, keys, keys2, vals, perm, n_q, drop, QKEY_BITS, st));
  // Sparse query sets take the lane walk (k_lane_q).  Measured crossover, uniform queries in the extended box (lane / packet walk,
  // RtreeBvh): blob-100k 100 k queries 1.36 / 3.65 ms, 1 M 2.70 / 3.45, 3 M 5.00 / 4.38, 10 M 12.3 / 6.7 (crossover ~2 M);
  // blob-1M 1 M 6.3 / 12.8 ms, 10 M 26.4 / 22.4 (~7 M).  Below it the packet walk lasts as long as its worst packet's chain of
  // dependent loads (2.7 ms), above it the lane walk's divergence costs more than the packets' union.  n* ~ 3500 T^0.55 fits both.
  // (End of round 4, leaf work queued and leaves of 4 - 8 for the packets — query_leaf_max: lane / packets, whole call: blob-100k 100 k queries 1.12 /
  // 1.53 ms, 300 k 1.47 / 1.46, 1 M 2.13 / 1.39, 10 M 10.9 / 3.60; blob-11k 30 k 0.72 / 0.66, 300 k 0.80 / 0.63: the crossover is at ~2.5 queries
  // per triangle now.)
  const bool lane_walk = query_walk_is_lane(n_q, n_tris, sign_src);
  // packets = leaves of the bucket k-d tree over the sorted keys (k_qcells); the launch has room for twice the consecutive
  // count, and k_qtable_mode falls back to consecutive packets should there be more
  const uint32_t* table = nullptr;
  uint32_t launched = packets;
  if (!lane_walk) {
    launched = nq / 32u + 64u;
    if (tuning().query_launch_tight != 0) launched = packets + 1u;   // test hook: forces the consecutive-packet fallback
    uint8_t* head = ws.take<uint8_t>(n_q);
    uint32_t* tb = ws.take<uint32_t>(n_q + 2);               // [0] count, [1] mode, then one start per head (at most n_q)
    size_t sel_bytes = 0;
    (void)select_flagged_indices(nullptr, sel_bytes, head, tb + 2, tb, n_q, st);
    void* sel_tmp = ws.take<char>(sel_bytes ? sel_bytes : 1);
    if (!head || !tb || !sel_tmp) { set_error("internal: query workspace too small"); return M2S_ERR_HIP_INTERNAL; }
    hipLaunchKernelGGL(k_qcells, dim3(nb), dim3(B), 0, st, keys2, nq, head);
    M2S_HIP_CHECK(select_flagged_indices(sel_tmp, sel_bytes, head, tb + 2, tb, n_q, st));
    hipLaunchKernelGGL(k_qtable_mode, dim3(1), dim3(1), 0, st, tb, nq, launched);
    table = tb;
  }
  // cut lists, one per packet (k_cut<false>): they need the packets' centres, and the kernel that finds those gathers the queries too
  const uint32_t qcut_min = tuning().query_cut_min;
  float4* centres = nullptr;
  if (table != nullptr && seeds && packets >= qcut_min) {
    centres = ws.take<float4>(launched);
    if (!centres) { set_error("internal: query workspace too small"); return M2S_ERR_HIP_INTERNAL; }
    hipLaunchKernelGGL(k_qpacket_bounds, dim3((launched + 3) / 4), dim3(256), 0, st, sorted, table, nq, launched, centres, d_queries, (const uint32_t*)perm);
  } else {
    hipLaunchKernelGGL(k_qgather, dim3(nb), dim3(B), 0, st, d_queries, perm, nq, sorted);
  }
  M2S_HIP_CHECK(hipGetLastError());
  plan->qb = qb; plan->perm = perm; plan->sorted = sorted; plan->table = table; plan->centres = centres;
  plan->launched = launched; plan->lane_walk = lane_walk; plan->seeds = seeds;
  return 0;
}

// Seed lattice of a query set: jump flooding over the QL^3 cells of plan.lat from the centroids `cen` (the sorted array of the finished
// mesh, or the input-order array while the mesh is being built: `ids` then name input triangles and launch_query_walk translates them).
int launch_query_seeds(Arena& ws, hipStream_t st, const float4* cen, uint32_t n_tris, const QueryPlan& plan, bool raw, QuerySeeds* out) {
  *out = QuerySeeds{};
  if (!plan.seeds || plan.lat == nullptr || n_tris == 0) return 0;
  DeviceMesh mesh{};
  mesh.cen = cen;
  mesh.n_tris = n_tris;
  GridParams g{};
  const size_t cells = (size_t)QL * QL * QL;
  unsigned long long* k64 = ws.take<unsigned long long>(cells);
  uint32_t* ids = ws.take<uint32_t>(cells);
  float4* la = ws.take<float4>(cells);
  float4* lb = ws.take<float4>(cells);
  if (!k64 || !ids || !la || !lb) { set_error("internal: query workspace too small"); return M2S_ERR_HIP_INTERNAL; }
  const GridParams* lat = plan.lat;
  M2S_HIP_CHECK(hipMemsetAsync(k64, 0xff, cells * 8, st));
  hipLaunchKernelGGL(k_jfa_splat, dim3((n_tris + 255) / 256), dim3(256), 0, st, mesh, g, lat, k64);
  const unsigned nbl = (unsigned)((cells + 255) / 256);
  hipLaunchKernelGGL(k_jfa_load, dim3(nbl), dim3(256), 0, st, mesh, k64, cells, la);
  float4 *src = la, *dst = lb;
  for (int step = QL / 2; step >= 1; step /= 2) {
    hipLaunchKernelGGL(k_jfa_pass, dim3(nbl), dim3(256), 0, st, g, lat, src, dst, step, nullptr);
    float4* t = src; src = dst; dst = t;
  }
  hipLaunchKernelGGL(k_jfa_pass, dim3(nbl), dim3(256), 0, st, g, lat, src, dst, 1, ids);
  M2S_HIP_CHECK(hipGetLastError());
  out->ids = ids;
  out->raw = raw;
  return 0;
}

int launch_query_walk(Arena& ws, hipStream_t st, const DeviceMesh& mesh, const float* d_queries, const QueryPlan& plan,
                      int mode, int sign_src, int algorithm, float* d_out, int* d_err, const QuerySeeds* pre) {
  const size_t n_q = plan.n_q;
  if (n_q == 0) return 0;
  GridParams g{};
  const uint32_t nq = (uint32_t)n_q;
  const uint32_t packets = (nq + 63) / 64;
  if (algorithm == 1) {
    if (mode == MODE_UNSIGNED && sign_src == SIGN_XRAY_ALL) launch_brute<false, MODE_UNSIGNED, SIGN_XRAY_ALL>(st, mesh, g, d_queries, nq, nullptr, d_out, d_err, packets);
    else if (mode == MODE_UNSIGNED && sign_src == SIGN_RAYS3) launch_brute<false, MODE_UNSIGNED, SIGN_RAYS3>(st, mesh, g, d_queries, nq, nullptr, d_out, d_err, packets);
    else if (mode == MODE_UNSIGNED) launch_brute<false, MODE_UNSIGNED, SIGN_NONE>(st, mesh, g, d_queries, nq, nullptr, d_out, d_err, packets);
    else if (mode == MODE_NORMAL_FOLD) launch_brute<false, MODE_NORMAL_FOLD, SIGN_NONE>(st, mesh, g, d_queries, nq, nullptr, d_out, d_err, packets);
    else launch_brute<false, MODE_NEAREST_NORMAL, SIGN_NONE>(st, mesh, g, d_queries, nq, nullptr, d_out, d_err, packets);
    M2S_HIP_CHECK(hipGetLastError());
    return 0;
  }
  const uint32_t* perm = plan.perm;
  const float4* sorted = plan.sorted;
  const uint32_t* table = plan.table;
  const uint32_t launched = plan.launched;
  const bool lane_walk = plan.lane_walk;
  // seeds: jump flooding over a QL^3 lattice on the query bounding box (as for the grid path) — here, or beside the build (`pre`)
  const uint32_t* seeds = nullptr;
  const GridParams* d_lat = nullptr;
  if (plan.seeds && mesh.n_tris) {
    QuerySeeds own;
    if (pre == nullptr || pre->ids == nullptr) {
      const int rc = launch_query_seeds(ws, st, mesh.cen, mesh.n_tris, plan, false, &own);
      if (rc) return rc;
      pre = &own;
    } else if (pre->raw) {
      const size_t cells = (size_t)QL * QL * QL;
      hipLaunchKernelGGL(k_seed_remap, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, st, pre->ids, cells, mesh.slot_of, mesh.n_tris);
    }
    seeds = pre->ids;
    d_lat = plan.lat;
  }
  CutList cut = {nullptr, 0, 0, 0, 0, nullptr};
  if (plan.centres != nullptr && seeds != nullptr) {
    uint32_t* lists = ws.take<uint32_t>((size_t)launched * CUT_WORDS);
    if (!lists) { set_error("internal: query workspace too small"); return M2S_ERR_HIP_INTERNAL; }
    const float emit_near = tuning().cut_near;
    const float emit_far = tuning().cut_far;   // 1/32: re-tuned at the end of round 3 (1/16 before): headline 9.19 -> 9.11 ms, 1024^3 x sheet-100k 92.95 -> 89.33 ms
    uint32_t depth = 1;
    while ((1ull << depth) < (unsigned long long)mesh.n_tris + 1ull) ++depth;
    const uint32_t wave_cap = tuning().cut_wave_cap ? tuning().cut_wave_cap : std::max(120u, 20u * depth);
    hipLaunchKernelGGL((k_cut<false, 0>), dim3((launched + 63) / 64), dim3(64), 0, st, mesh, g, seeds, 0u, 0u, 0u, launched, 1u, 1u, lists,
                       emit_near, emit_far, 100000u, wave_cap, (const float4*)plan.centres, table, d_lat);
    cut = {lists, 0, 0, 0, 0, plan.centres};
  }
  if (lane_walk) {
    const unsigned lb = (nq + 255u) / 256u;
    if (mode == MODE_UNSIGNED && sign_src == SIGN_RAYS3) hipLaunchKernelGGL((k_lane_q<MODE_UNSIGNED, SIGN_RAYS3>), dim3(lb), dim3(256), 0, st, mesh, (const float4*)sorted, (const uint32_t*)perm, nq, d_out, d_err, seeds, d_lat);
    else if (mode == MODE_UNSIGNED) hipLaunchKernelGGL((k_lane_q<MODE_UNSIGNED, SIGN_NONE>), dim3(lb), dim3(256), 0, st, mesh, (const float4*)sorted, (const uint32_t*)perm, nq, d_out, d_err, seeds, d_lat);
    else if (mode == MODE_NORMAL_FOLD) hipLaunchKernelGGL((k_lane_q<MODE_NORMAL_FOLD, SIGN_NONE>), dim3(lb), dim3(256), 0, st, mesh, (const float4*)sorted, (const uint32_t*)perm, nq, d_out, d_err, seeds, d_lat);
    else hipLaunchKernelGGL((k_lane_q<MODE_NEAREST_NORMAL, SIGN_NONE>), dim3(lb), dim3(256), 0, st, mesh, (const float4*)sorted, (const uint32_t*)perm, nq, d_out, d_err, seeds, d_lat);
    M2S_HIP_CHECK(hipGetLastError());
    return 0;
  }
  if (mode == MODE_UNSIGNED && sign_src == SIGN_RAYS3) launch_packet<false, MODE_UNSIGNED, SIGN_RAYS3>(st, mesh, g, sorted, perm, nq, table, d_out, d_err, launched, seeds, 0, 0, 0, d_lat, cut, nullptr, nullptr, tuning().defer == 0 ? 0 : tuning().defer == 1 ? 1 : 3);
  else if (mode == MODE_UNSIGNED) launch_packet<false, MODE_UNSIGNED, SIGN_NONE>(st, mesh, g, sorted, perm, nq, table, d_out, d_err, launched, seeds, 0, 0, 0, d_lat, cut, nullptr, nullptr, tuning().defer == 0 ? 0 : tuning().defer == 1 ? 1 : 3);
  else if (mode == MODE_NORMAL_FOLD) launch_packet<false, MODE_NORMAL_FOLD, SIGN_NONE>(st, mesh, g, sorted, perm, nq, table, d_out, d_err, launched, seeds, 0, 0, 0, d_lat, cut, nullptr, nullptr, tuning().defer == 0 ? 0 : tuning().defer == 1 ? 1 : 3);
  else launch_packet<false, MODE_NEAREST_NORMAL, SIGN_NONE>(st, mesh, g, sorted, perm, nq, table, d_out, d_err, launched, seeds, 0, 0, 0, d_lat, cut, nullptr, nullptr, tuning().defer == 0 ? 0 : tuning().defer == 1 ? 1 : 3);
  M2S_HIP_CHECK(hipGetLastError());
  return 0;
}

int launch_query_distance(Arena& ws, hipStream_t st, const DeviceMesh& mesh, const float* d_queries, size_t n_q,
                          int mode, int sign_src, int algorithm, float* d_out, int* d_err) {
  if (query_is_tiny(n_q, mesh.n_tris, algorithm, sign_src) && mesh.stats == nullptr)
    return launch_query_brute_split(ws, st, mesh, d_queries, n_q, mode, sign_src, d_out, d_err);
  QueryPlan plan;
  const int rc = prepare_query_walk(ws, st, d_queries, n_q, mesh.n_tris, sign_src, algorithm, &plan, nullptr);
  if (rc) return rc;
  return launch_query_walk(ws, st, mesh, d_queries, plan, mode, sign_src, algorithm, d_out, d_err, nullptr);
}

// m2s_warmup: the first launch of a kernel of this translation unit makes the runtime load its code object (all its kernels).
__global__ void k_warm_distance() {}
void warm_distance(hipStream_t st) {
  hipLaunchKernelGGL(k_warm_distance, dim3(1), dim3(64), 0, st);
  // ... and resolves a kernel FUNCTION at its own first launch (~0.3 ms each): ask for the attributes of the ones a first call uses
  const void* fns[] = {
      (const void*)k_packet<true, MODE_UNSIGNED, SIGN_GRID_PLANE, false, false, 3>,
      (const void*)k_packet<true, MODE_UNSIGNED, SIGN_GRID_PLANE, false, false, 2>,
      (const void*)k_packet<true, MODE_NORMAL_FOLD, SIGN_NONE, false, false, 3>,
      (const void*)k_packet<true, MODE_NORMAL_FOLD, SIGN_NONE, false, false, 2>,
      (const void*)k_packet<false, MODE_UNSIGNED, SIGN_RAYS3, false, false, 3>,
      (const void*)k_packet<false, MODE_NEAREST_NORMAL, SIGN_NONE, false, false, 3>,
      (const void*)k_split_init,
      (const void*)k_split_round<MODE_UNSIGNED>,
      (const void*)k_split_round<MODE_NORMAL_FOLD>,
      (const void*)k_split_finish<MODE_UNSIGNED, SIGN_GRID_PLANE>,
      (const void*)k_split_finish<MODE_NORMAL_FOLD, SIGN_NONE>,
      (const void*)k_cut<true, 0>,
      (const void*)k_cut<true, 1>,
      (const void*)k_cut<true, 2>,
      (const void*)k_cut<false, 0>,
      (const void*)k_lane<MODE_UNSIGNED, SIGN_GRID_PLANE>,
      (const void*)k_lane_q<MODE_UNSIGNED, SIGN_RAYS3>,
      (const void*)k_jfa_splat,
      (const void*)k_jfa_load,
      (const void*)k_jfa_pass32,
      (const void*)k_seed_remap,
      (const void*)k_qbounds,
      (const void*)k_qbounds_final,
      (const void*)k_qkeys,
      (const void*)k_qgather,
      (const void*)k_qcells,
      (const void*)k_qtable_mode,
      (const void*)k_qpacket_bounds,
      (const void*)k_qlattice,
      (const void*)k_push_cells};
  hipFuncAttributes attr;
  for (const void* f : fns) (void)hipFuncGetAttributes(&attr, f);
  (void)hipGetLastError();
}

}  // namespace m2s
