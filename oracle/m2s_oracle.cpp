// m2s_oracle.cpp — CPU ORACLE. TEST INFRASTRUCTURE ONLY, NOT THE PRODUCT.
//
// A scalar C++17 restatement of the reference's per-query-point hot path
// (Azkellas/mesh_to_sdf 0.4.0, reference checked out at /root/reference), used only as
// the checker in tests/, in __graft_entry__.smoke() and as the `cpu_baseline` leg of
// bench.py.  Nothing under mesh_to_sdf_amd/ links, loads or calls this file.
//
// Every function cites the reference file:line it follows (paths relative to
// /root/reference/mesh_to_sdf/src).  Arithmetic is IEEE binary32 with the reference's
// operation order; compile with -ffp-contract=off and without -ffast-math / -march=native
// (Rust never contracts a*b+c into an FMA).
//
// PINNING.  The reference is Rust and no Rust toolchain exists in this image, so the
// reference itself cannot be executed here.  The oracle is pinned on every known-answer
// test the reference's own test-suite holds for this path (see tests/test_oracle_kat.py):
// lib.rs:13-31,58 / lib.rs:269-289 / generate/grid.rs:207-231 (exact 1.0 answers),
// generate/grid.rs:693-724 (grid == brute force on 125 cells, assert_eq), grid.rs:201-297
// (index / centre / snap), geo.rs:311-323 (segment), point/impl_array.rs (length, dist),
// generic/default.rs:83-109 (suzanne, three recorded external values, tol 0.1; the third
// matches pysdf's recorded 0.45411023 to the last digit), proptest-regressions/geo.txt.
//
// Third-party crates on the path (not vendored under /root/reference): bvh 0.10.0,
// rstar 0.12.0 (Cargo.lock:577-580, 3155-3158) — they only SELECT candidate triangles;
// every distance / ray hit that reaches the output is geo.rs arithmetic.  Where their
// tie-breaking is not observable from the reference tree the oracle documents its choice
// (search for "UNPINNED").

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <thread>
#include <vector>

namespace {

constexpr float F32_MAX = std::numeric_limits<float>::max();

struct V3 {
  float x, y, z;
};

// ---- point.rs:79-141 — default Point ops, exact operation order --------------------
inline V3 v_add(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }   // point.rs:81-87
inline V3 v_sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }   // point.rs:90-96
inline float v_dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }  // point.rs:99-101
inline V3 v_cross(V3 a, V3 b) {                                             // point.rs:104-110
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
inline float v_length(V3 a) { return std::sqrt(v_dot(a, a)); }               // point.rs:113-115
inline float v_dist(V3 a, V3 b) { return v_length(v_sub(a, b)); }            // point.rs:118-120
inline float v_dist2(V3 a, V3 b) {                                           // point.rs:123-126
  V3 d = v_sub(a, b);
  return v_dot(d, d);
}
inline V3 v_fmul(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }      // point.rs:130-132
inline V3 v_comp_div(V3 a, V3 b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }  // point.rs:135-141
inline bool v_eq(V3 a, V3 b) { return a.x == b.x && a.y == b.y && a.z == b.z; }  // derive(PartialEq)
inline float v_get(V3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }

inline V3 ld3(const float* p) { return {p[0], p[1], p[2]}; }
inline void st3(float* p, V3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }

// Rust f32::min / f32::max: if one argument is NaN the other is returned.
inline float rs_min(float a, float b) { return std::fmin(a, b); }
inline float rs_max(float a, float b) { return std::fmax(a, b); }

// Rust `f as usize` / `f as isize`: saturating, NaN -> 0.
inline uint64_t f32_as_usize(float f) {
  if (!(f == f)) return 0;
  if (f <= 0.0f) return 0;
  if (f >= 18446744073709551616.0f) return UINT64_MAX;
  return (uint64_t)f;
}
inline int64_t f32_as_isize(float f) {
  if (!(f == f)) return 0;
  if (f <= -9223372036854775808.0f) return INT64_MIN;
  if (f >= 9223372036854775808.0f) return INT64_MAX;
  return (int64_t)f;
}

// ---- geo.rs:4-22 — triangle AABB padded by 1e-4 -----------------------------------
inline void triangle_bounding_box(V3 a, V3 b, V3 c, V3* mn, V3* mx) {
  const float EPS = 0.0001f;
  V3 lo = {rs_min(a.x, rs_min(b.x, c.x)), rs_min(a.y, rs_min(b.y, c.y)), rs_min(a.z, rs_min(b.z, c.z))};
  V3 hi = {rs_max(a.x, rs_max(b.x, c.x)), rs_max(a.y, rs_max(b.y, c.y)), rs_max(a.z, rs_max(b.z, c.z))};
  V3 e = {EPS, EPS, EPS};
  *mn = v_sub(lo, e);
  *mx = v_add(hi, e);
}

// ---- geo.rs:141-151 — closest point on segment ------------------------------------
inline V3 closest_point_segment(V3 p, V3 a, V3 b) {
  V3 ab = v_sub(b, a);
  float m = v_dot(ab, ab);
  V3 ap = v_sub(p, a);
  float s12 = v_dot(ab, ap) / m;
  // f32::clamp(0.0, 1.0): NaN stays NaN.
  if (s12 < 0.0f) s12 = 0.0f;
  else if (s12 > 1.0f) s12 = 1.0f;
  return v_add(a, v_fmul(ab, s12));
}

// ---- geo.rs:70-138 — Embree-derived closest point on triangle ---------------------
inline V3 closest_point_triangle(V3 p, V3 a, V3 b, V3 c) {
  const bool ab_eq = v_eq(a, b), bc_eq = v_eq(b, c), ac_eq = v_eq(a, c);
  if (ab_eq && bc_eq && ac_eq) return a;                 // geo.rs:74-76
  if (ab_eq) return closest_point_segment(p, a, c);      // geo.rs:77-79
  if (bc_eq) return closest_point_segment(p, a, b);      // geo.rs:80-82
  if (ac_eq) return closest_point_segment(p, a, b);      // geo.rs:83-85

  V3 ab = v_sub(b, a);
  V3 ac = v_sub(c, a);
  V3 ap = v_sub(p, a);

  float d1 = v_dot(ab, ap);
  float d2 = v_dot(ac, ap);
  if (d1 <= 0.0f && d2 <= 0.0f) return a;                // geo.rs:97-99

  V3 bp = v_sub(p, b);
  float d3 = v_dot(ab, bp);
  float d4 = v_dot(ac, bp);
  if (d3 >= 0.0f && d4 <= d3) return b;                  // geo.rs:104-106

  V3 cp = v_sub(p, c);
  float d5 = v_dot(ab, cp);
  float d6 = v_dot(ac, cp);
  if (d6 >= 0.0f && d5 <= d6) return c;                  // geo.rs:111-113

  float vc = d1 * d4 - d3 * d2;
  if (vc <= 0.0f && d1 >= 0.0f && d3 <= 0.0f) {          // geo.rs:116-119
    float v = d1 / (d1 - d3);
    return v_add(a, v_fmul(ab, v));
  }

  float vb = d5 * d2 - d1 * d6;
  if (vb <= 0.0f && d2 >= 0.0f && d6 <= 0.0f) {          // geo.rs:122-125
    float v = d2 / (d2 - d6);
    return v_add(a, v_fmul(ac, v));
  }

  float va = d3 * d6 - d5 * d4;
  if (va <= 0.0f && d4 - d3 >= 0.0f && d5 - d6 >= 0.0f) {  // geo.rs:128-132
    float v = (d4 - d3) / ((d4 - d3) + (d5 - d6));
    V3 bc = v_sub(c, b);
    return v_add(b, v_fmul(bc, v));
  }

  float denom = 1.0f / (va + vb + vc);                   // geo.rs:134-137
  float v = vb * denom;
  float w = vc * denom;
  return v_add(v_add(a, v_fmul(ab, v)), v_fmul(ac, w));
}

// geo.rs:26-30
inline float point_triangle_distance(V3 x0, V3 x1, V3 x2, V3 x3) {
  return v_dist(x0, closest_point_triangle(x0, x1, x2, x3));
}
// geo.rs:33-37
inline float point_triangle_distance2(V3 x0, V3 x1, V3 x2, V3 x3) {
  return v_dist2(x0, closest_point_triangle(x0, x1, x2, x3));
}
// geo.rs:60-64 (normal NOT normalised) and geo.rs:43-56
inline float point_triangle_signed_distance(V3 x0, V3 x1, V3 x2, V3 x3) {
  V3 nearest = closest_point_triangle(x0, x1, x2, x3);
  V3 direction = v_sub(x0, nearest);
  V3 normal = v_cross(v_sub(x2, x1), v_sub(x3, x1));
  float distance = v_dist(x0, nearest);
  return (v_dot(direction, normal) > 0.0f) ? distance : -distance;
}

// ---- geo.rs:165-216 — axis aligned ray / triangle ---------------------------------
// axis: 0 = X (plane y,z), 1 = Y (plane z,x), 2 = Z (plane x,y)   geo.rs:181-195
inline bool ray_triangle_intersection_aligned(V3 o, V3 t0, V3 t1, V3 t2, int axis, float* t_out) {
  V3 edge01 = v_sub(t1, t0);
  V3 edge12 = v_sub(t2, t1);
  V3 edge20 = v_sub(t0, t2);
  V3 p0 = v_sub(o, t0);
  V3 p1 = v_sub(o, t1);
  V3 p2 = v_sub(o, t2);
  auto gy = [axis](V3 v) { return axis == 0 ? v.y : (axis == 1 ? v.z : v.x); };
  auto gz = [axis](V3 v) { return axis == 0 ? v.z : (axis == 1 ? v.x : v.y); };
  auto gx = [axis](V3 v) { return axis == 0 ? v.x : (axis == 1 ? v.y : v.z); };

  float w0 = gz(p1) * gy(edge12) - gy(p1) * gz(edge12);  // geo.rs:199
  float w1 = gz(p2) * gy(edge20) - gy(p2) * gz(edge20);  // geo.rs:200
  float w2 = gz(p0) * gy(edge01) - gy(p0) * gz(edge01);  // geo.rs:201

  if ((w0 < 0.0f && w1 < 0.0f && w2 < 0.0f) || (w0 > 0.0f && w1 > 0.0f && w2 > 0.0f)) {  // geo.rs:203
    float t = -(w0 * gx(p0) + w2 * gx(p2) + w1 * gx(p1)) / (w0 + w1 + w2);             // geo.rs:208
    if (t > 0.0f) {                                                                     // geo.rs:210
      *t_out = t;
      return true;
    }
  }
  return false;
}

// ---- float-cmp 0.9.0 approx_eq!(f32, a, b, ulps = 2, epsilon = 1e-6) ---------------
// (crate not vendored; semantics: a == b || |a-b| <= eps || |bits(a)-bits(b)| <= ulps,
// bit difference taken as wrapping i32 subtraction with saturating abs.)
inline bool approx_eq_f32(float a, float b, int32_t ulps, float eps) {
  if (a == b) return true;
  float d = std::fabs(a - b);
  if (d <= eps) return true;
  int32_t ai, bi;
  std::memcpy(&ai, &a, 4);
  std::memcpy(&bi, &b, 4);
  int32_t diff = (int32_t)((uint32_t)ai - (uint32_t)bi);
  int32_t ad = diff == INT32_MIN ? INT32_MAX : (diff < 0 ? -diff : diff);
  return ad <= ulps;
}

// ---- lib.rs:242-259 — compare_distances -------------------------------------------
// returns -1 Less, 0 Equal, +1 Greater, -2 = the reference would panic ("NaN distance").
inline int compare_distances(float a, float b) {
  float aa = std::fabs(a), bb = std::fabs(b);
  if (approx_eq_f32(aa, bb, 2, 1e-6f)) {
    bool an = std::signbit(a), bn = std::signbit(b);
    if (an && !bn) return 1;
    if (!an && bn) return -1;
    if (aa < bb) return -1;
    if (aa > bb) return 1;
    if (aa == bb) return 0;
    return -2;  // partial_cmp(..).unwrap() on NaN
  }
  if (aa < bb) return -1;
  if (aa > bb) return 1;
  if (aa == bb) return 0;
  return -2;  // .expect("NaN distance")
}

// ---- grid.rs ----------------------------------------------------------------------
struct Grid {
  V3 first_cell;
  V3 cell_size;
  uint64_t count[3];
};
// grid.rs:59-74
inline Grid grid_from_bounding_box(V3 bmin, V3 bmax, const uint64_t count[3]) {
  V3 fc = {(float)count[0], (float)count[1], (float)count[2]};
  V3 cell_size = v_comp_div(v_sub(bmax, bmin), fc);
  V3 first = v_add(bmin, v_fmul(cell_size, 0.5f));
  return {first, cell_size, {count[0], count[1], count[2]}};
}
// grid.rs:110-119 (min only; the box min is RECOMPUTED from first_cell)
inline V3 grid_bbox_min(const Grid& g) { return v_sub(g.first_cell, v_fmul(g.cell_size, 0.5f)); }
// grid.rs:122-124
inline uint64_t grid_cell_idx(const Grid& g, const uint64_t c[3]) {
  return c[2] + c[1] * g.count[2] + c[0] * g.count[1] * g.count[2];
}
// grid.rs:127-132
inline void grid_cell_coords(const Grid& g, uint64_t idx, uint64_t c[3]) {
  c[2] = idx % g.count[2];
  c[1] = (idx / g.count[2]) % g.count[1];
  c[0] = idx / (g.count[1] * g.count[2]);
}
// grid.rs:135-141
inline V3 grid_cell_center(const Grid& g, const uint64_t c[3]) {
  return {g.first_cell.x + (float)c[0] * g.cell_size.x, g.first_cell.y + (float)c[1] * g.cell_size.y,
          g.first_cell.z + (float)c[2] * g.cell_size.z};
}
// grid.rs:145-170 — returns true when Inside
inline bool grid_snap(const Grid& g, V3 p, uint64_t out[3]) {
  V3 cell = v_comp_div(v_sub(p, grid_bbox_min(g)), g.cell_size);
  int64_t ic[3] = {f32_as_isize(std::floor(cell.x)), f32_as_isize(std::floor(cell.y)),
                   f32_as_isize(std::floor(cell.z))};
  bool inside = true;
  for (int i = 0; i < 3; ++i) {
    int64_t hi = (int64_t)g.count[i] - 1;
    int64_t r = ic[i] < 0 ? 0 : (ic[i] > hi ? hi : ic[i]);
    if (r != ic[i]) inside = false;
    out[i] = (uint64_t)r;
  }
  return inside;
}

// ---- lib.rs:175-193 — Topology::get_triangles --------------------------------------
// topology 0 = TriangleList (tuples(): trailing partial dropped), 1 = TriangleStrip
// (tuple_windows(): sliding window, NO winding flip).  indices == nullptr => 0..n_verts.
// Returns -1 if an index is out of range (the reference would panic on vertices[i]).
int get_triangles(size_t n_verts, const uint32_t* indices, size_t n_indices, int topology,
                  std::vector<uint32_t>* tris) {
  size_t n = indices ? n_indices : n_verts;
  auto at = [&](size_t i) -> uint32_t { return indices ? indices[i] : (uint32_t)i; };
  tris->clear();
  if (topology == 0) {
    for (size_t i = 0; i + 2 < n; i += 3) {
      tris->push_back(at(i));
      tris->push_back(at(i + 1));
      tris->push_back(at(i + 2));
    }
  } else {
    for (size_t i = 0; i + 2 < n; ++i) {
      tris->push_back(at(i));
      tris->push_back(at(i + 1));
      tris->push_back(at(i + 2));
    }
  }
  for (uint32_t v : *tris)
    if (v >= n_verts) return -1;
  return 0;
}

struct Mesh {
  const float* verts;
  std::vector<uint32_t> tris;  // 3 per triangle
  size_t ntri() const { return tris.size() / 3; }
  V3 v(size_t t, int k) const { return ld3(verts + 3 * (size_t)tris[3 * t + k]); }
};

// Candidate rule of `bvh.traverse(&ray, shapes)` (bvh 0.10.0, not vendored): a triangle is
// a candidate iff the axis-aligned ray meets its 1e-4-padded AABB.  Restated as the closed
// interval test below.  UNPINNED corner: an origin coordinate exactly ON a padded face is
// 0*inf = NaN inside the crate's slab test; no reference test covers it, and no true hit
// (strictly inside the triangle, geo.rs:203) can lie on a padded face.
inline bool ray_meets_padded_aabb(V3 o, V3 mn, V3 mx, int axis) {
  if (axis == 0) return o.y >= mn.y && o.y <= mx.y && o.z >= mn.z && o.z <= mx.z && mx.x >= o.x;
  if (axis == 1) return o.z >= mn.z && o.z <= mx.z && o.x >= mn.x && o.x <= mx.x && mx.y >= o.y;
  return o.x >= mn.x && o.x <= mx.x && o.y >= mn.y && o.y <= mx.y && mx.z >= o.z;
}

// Count hits of the +axis ray from `o`; `bvh_filter` applies the candidate rule above.
inline uint32_t count_ray_hits(const Mesh& m, V3 o, int axis, bool bvh_filter) {
  uint32_t n = 0;
  for (size_t t = 0; t < m.ntri(); ++t) {
    V3 a = m.v(t, 0), b = m.v(t, 1), c = m.v(t, 2);
    if (bvh_filter) {
      V3 mn, mx;
      triangle_bounding_box(a, b, c, &mn, &mx);
      if (!ray_meets_padded_aabb(o, mn, mx, axis)) continue;
    }
    float tt;
    if (ray_triangle_intersection_aligned(o, a, b, c, axis, &tt)) ++n;
  }
  return n;
}

// One query of generate_sdf for each back-end.  accel: 0 None, 1 Bvh, 2 Rtree, 3 RtreeBvh;
// sign: 0 Raycast, 1 Normal.  Returns 0 ok, -2 NaN panic, -3 empty-mesh panic.
int query_one(const Mesh& m, V3 q, int accel, int sign, float* out) {
  const size_t T = m.ntri();
  if (accel == 0) {
    // generic/default.rs:27-73
    if (sign == 0) {
      float mind = F32_MAX;
      uint32_t cnt = 0;
      for (size_t t = 0; t < T; ++t) {
        V3 a = m.v(t, 0), b = m.v(t, 1), c = m.v(t, 2);
        mind = rs_min(mind, point_triangle_distance(q, a, b, c));
        float tt;
        cnt += ray_triangle_intersection_aligned(q, a, b, c, 0, &tt) ? 1u : 0u;
      }
      *out = (cnt % 2 == 0) ? mind : -mind;
      return 0;
    }
    float mind = F32_MAX;
    for (size_t t = 0; t < T; ++t) {
      float d = point_triangle_signed_distance(q, m.v(t, 0), m.v(t, 1), m.v(t, 2));
      int c = compare_distances(d, mind);  // default.rs:52-59
      if (c == -2) return -2;
      if (c == -1) mind = d;
    }
    *out = mind;
    return 0;
  }
  if (accel == 1) {
    // generic/bvh.rs:76-144.  nearest_candidates (bvh_ext.rs:59-78) returns a superset of
    // the true nearest triangles; the final min / fold only depends on the near-minimum
    // ones, so the candidate set is restated as "all triangles, index order"
    // (UNPINNED: the crate's candidate ORDER; see tests for the order-independence proof).
    if (sign == 1) {
      float mind = F32_MAX;
      for (size_t t = 0; t < T; ++t) {
        float d = point_triangle_signed_distance(q, m.v(t, 0), m.v(t, 1), m.v(t, 2));
        int c = compare_distances(mind, d);  // bvh.rs:90
        if (c == -2) return -2;
        if (c == 1) mind = d;
      }
      *out = mind;
      return 0;
    }
    float mind = F32_MAX;
    for (size_t t = 0; t < T; ++t)
      mind = rs_min(mind, point_triangle_distance(q, m.v(t, 0), m.v(t, 1), m.v(t, 2)));
    int insides = 0;
    for (int axis = 0; axis < 3; ++axis)
      if (count_ray_hits(m, q, axis, true) % 2 == 1) ++insides;  // bvh.rs:106-134
    *out = insides > 1 ? -mind : mind;                             // bvh.rs:137-141
    return 0;
  }
  // Rtree / RtreeBvh: rstar nearest_neighbor with distance_2 = point_triangle_distance2
  // (rtree.rs:64-77).  UNPINNED: rstar's choice among exactly tied distance_2 values;
  // restated as the lowest triangle index.
  if (T == 0) return -3;  // rtree.rs:117 unwrap on None (RtreeBvh handles empty before, see caller)
  size_t best = 0;
  float best_d2 = 0.0f;
  bool have = false;
  for (size_t t = 0; t < T; ++t) {
    float d2 = point_triangle_distance2(q, m.v(t, 0), m.v(t, 1), m.v(t, 2));
    if (d2 == d2 && (!have || d2 < best_d2)) { best = t; best_d2 = d2; have = true; }  // a NaN distance_2 is never nearest
  }
  V3 a = m.v(best, 0), b = m.v(best, 1), c = m.v(best, 2);
  if (accel == 2) {
    *out = point_triangle_signed_distance(q, a, b, c);  // rtree.rs:118-123
    return 0;
  }
  float dist = point_triangle_distance(q, a, b, c);     // rtree_bvh.rs:129-134
  int insides = 0;
  for (int axis = 0; axis < 3; ++axis)
    if (count_ray_hits(m, q, axis, true) % 2 == 1) ++insides;  // rtree_bvh.rs:136-164
  *out = insides > 1 ? -dist : dist;                            // rtree_bvh.rs:167-171
  return 0;
}

template <class F>
void parallel_for(size_t n, int threads, F f) {
  if (threads <= 1 || n < 2) {
    for (size_t i = 0; i < n; ++i) f(i, 0);
    return;
  }
  std::atomic<size_t> next{0};
  const size_t chunk = std::max<size_t>(1, n / ((size_t)threads * 16));
  std::vector<std::thread> pool;
  for (int t = 0; t < threads; ++t)
    pool.emplace_back([&, t] {
      for (;;) {
        size_t b = next.fetch_add(chunk);
        if (b >= n) break;
        size_t e = std::min(n, b + chunk);
        for (size_t i = b; i < e; ++i) f(i, t);
      }
    });
  for (auto& th : pool) th.join();
}

// ---- grid Raycast sign, generate/grid.rs:568-684 -----------------------------------
// Result: for each voxel and axis, parity of the number of hits of that voxel's grid line
// (origin = centre of cell 0 on the line, direction +axis) whose bucket
// k = min(floor(t / cell_size[axis]) as usize, n_axis - 1) is >= the voxel's coordinate
// (grid.rs:601-617).  The candidate enumeration is per TRIANGLE here (every line whose
// cell-0 centre satisfies ray_meets_padded_aabb) instead of per ray through the bvh
// crate; the hit set is identical by construction.
void grid_ray_parity(const Mesh& m, const Grid& g, int threads, std::vector<uint8_t>* inter /*3 per voxel, mod 2*/) {
  const uint64_t nx = g.count[0], ny = g.count[1], nz = g.count[2];
  const uint64_t total = nx * ny * nz;
  inter->assign(total * 3, 0);
  if (total == 0) return;
  // marker[axis][voxel] toggled at bucket k; suffix-xor along the axis gives the parity.
  std::vector<std::atomic<uint8_t>> marker(total * 3);
  for (auto& a : marker) a.store(0, std::memory_order_relaxed);
  const uint64_t n_ax[3] = {nx, ny, nz};
  parallel_for(m.ntri(), threads, [&](size_t t, int) {
    V3 a = m.v(t, 0), b = m.v(t, 1), c = m.v(t, 2);
    V3 mn, mx;
    triangle_bounding_box(a, b, c, &mn, &mx);
    for (int axis = 0; axis < 3; ++axis) {
      const int u = axis == 0 ? 1 : (axis == 1 ? 0 : 0);  // first free grid axis
      const int w = axis == 0 ? 2 : (axis == 1 ? 2 : 1);  // second free grid axis
      // Conservative index window of lines whose cell-0 centre can lie in [mn, mx] on (u, w);
      // every line in the window is then tested with the exact closed-interval rule.
      uint64_t lo[2], hi[2];
      bool empty = false;
      const int free_ax[2] = {u, w};
      for (int k = 0; k < 2 && !empty; ++k) {
        int ax = free_ax[k];
        float fc = v_get(g.first_cell, ax), cs = v_get(g.cell_size, ax);
        float bmn = v_get(mn, ax), bmx = v_get(mx, ax);
        uint64_t n = n_ax[ax];
        if (cs == 0.0f || !(cs == cs)) { lo[k] = 0; hi[k] = n - 1; continue; }
        double i0 = ((double)bmn - (double)fc) / (double)cs, i1 = ((double)bmx - (double)fc) / (double)cs;
        if (i0 > i1) std::swap(i0, i1);
        if (!(i0 == i0) || !(i1 == i1)) { lo[k] = 0; hi[k] = n - 1; continue; }
        double l = std::floor(i0) - 1.0, h = std::ceil(i1) + 1.0;
        if (h < 0.0 || l > (double)(n - 1)) { empty = true; break; }
        lo[k] = l < 0.0 ? 0 : (uint64_t)l;
        hi[k] = h > (double)(n - 1) ? n - 1 : (uint64_t)h;
      }
      if (empty) continue;
      for (uint64_t iu = lo[0]; iu <= hi[0]; ++iu)
        for (uint64_t iw = lo[1]; iw <= hi[1]; ++iw) {
          uint64_t cell[3] = {0, 0, 0};
          cell[u] = iu;
          cell[w] = iw;
          V3 o = grid_cell_center(g, cell);            // grid.rs:570
          if (!ray_meets_padded_aabb(o, mn, mx, axis)) continue;
          float tt;
          if (!ray_triangle_intersection_aligned(o, a, b, c, axis, &tt)) continue;  // grid.rs:601-603
          float fcnt = tt / v_get(g.cell_size, axis);                                // grid.rs:605
          uint64_t k = std::min<uint64_t>(f32_as_usize(std::floor(fcnt)), n_ax[axis] - 1);  // grid.rs:606-607
          cell[axis] = k;
          marker[grid_cell_idx(g, cell) * 3 + axis].fetch_xor(1, std::memory_order_relaxed);
        }
    }
  });
  inter->assign(total * 3, 0);
  // suffix parity along each axis == "fetch_add(1) on cells 0..=k" mod 2 (grid.rs:612-617)
  const uint64_t stride[3] = {ny * nz, nz, 1};
  for (int axis = 0; axis < 3; ++axis) {
    const int u = axis == 0 ? 1 : 0, w = axis == 2 ? 1 : 2;
    parallel_for(n_ax[u] * n_ax[w], threads, [&](size_t li, int) {
      uint64_t cell[3] = {0, 0, 0};
      cell[u] = li / n_ax[w];
      cell[w] = li % n_ax[w];
      uint64_t base = grid_cell_idx(g, cell);
      uint8_t run = 0;
      for (uint64_t i = n_ax[axis]; i-- > 0;) {
        uint64_t idx = base + i * stride[axis];
        run ^= marker[idx * 3 + axis].load(std::memory_order_relaxed);
        (*inter)[idx * 3 + axis] = run;
      }
    });
  }
}

// generate/grid.rs:622-639 — best of three
inline void apply_parity_sign(std::vector<uint8_t>& inter, float* dist, uint64_t total, int threads) {
  parallel_for(total, threads, [&](size_t i, int) {
    int odd = inter[i * 3] + inter[i * 3 + 1] + inter[i * 3 + 2];
    if (odd >= 2) dist[i] = -dist[i];
  });
}

// ---- reference-propagation semantics, generate/grid.rs:265-558 ---------------------
struct State {
  float distance;
  uint64_t cell[3];
  uint32_t tri[3];
};
// Ord for State, generate/grid.rs:27-35.  Returns <0, 0, >0 for self vs other.
inline int state_cmp(const State& s, const State& o) {
  int c = compare_distances(o.distance, s.distance);
  if (c != 0) return c;
  for (int i = 0; i < 3; ++i)
    if (s.cell[i] != o.cell[i]) return s.cell[i] < o.cell[i] ? -1 : 1;
  for (int i = 0; i < 3; ++i)
    if (s.tri[i] != o.tri[i]) return s.tri[i] < o.tri[i] ? -1 : 1;
  return 0;
}

// Max-heap following the structure of Rust's std::collections::BinaryHeap (rebuild by
// sift_down from n/2-1..0; pop = swap last into root, sift_down_to_bottom, sift_up;
// push = sift_up).  The comparator is not a total order (compare_distances is
// non-transitive), so pop order in near-ties depends on these details.
struct RsHeap {
  std::vector<State> d;
  static bool le(const State& a, const State& b) { return state_cmp(a, b) <= 0; }
  static bool lt(const State& a, const State& b) { return state_cmp(a, b) < 0; }
  void sift_up(size_t start, size_t pos) {
    State e = d[pos];
    while (pos > start) {
      size_t parent = (pos - 1) / 2;
      if (le(e, d[parent])) break;
      d[pos] = d[parent];
      pos = parent;
    }
    d[pos] = e;
  }
  void sift_down_range(size_t pos, size_t end) {
    State e = d[pos];
    size_t child = 2 * pos + 1;
    while (end >= 2 && child <= end - 2) {
      child += le(d[child], d[child + 1]) ? 1 : 0;
      if (!lt(e, d[child])) { d[pos] = e; return; }  // hole.element() >= child
      d[pos] = d[child];
      pos = child;
      child = 2 * pos + 1;
    }
    if (child == end - 1 && lt(e, d[child])) {
      d[pos] = d[child];
      pos = child;
    }
    d[pos] = e;
  }
  void sift_down_to_bottom(size_t pos) {
    size_t end = d.size(), start = pos;
    State e = d[pos];
    size_t child = 2 * pos + 1;
    while (end >= 2 && child <= end - 2) {
      child += le(d[child], d[child + 1]) ? 1 : 0;
      d[pos] = d[child];
      pos = child;
      child = 2 * pos + 1;
    }
    if (child == end - 1) {
      d[pos] = d[child];
      pos = child;
    }
    d[pos] = e;
    sift_up(start, pos);
  }
  void rebuild() {
    size_t n = d.size() / 2;
    while (n > 0) {
      --n;
      sift_down_range(n, d.size());
    }
  }
  void push(const State& s) {
    d.push_back(s);
    sift_up(0, d.size() - 1);
  }
  bool pop(State* out) {
    if (d.empty()) return false;
    State item = d.back();
    d.pop_back();
    if (!d.empty()) {
      std::swap(item, d[0]);
      sift_down_to_bottom(0);
    }
    *out = item;
    return true;
  }
};

inline float grid_tri_distance(const Mesh& m, const uint32_t tri[3], V3 p, int sign) {
  V3 a = ld3(m.verts + 3 * (size_t)tri[0]), b = ld3(m.verts + 3 * (size_t)tri[1]), c = ld3(m.verts + 3 * (size_t)tri[2]);
  return sign == 0 ? point_triangle_distance(p, a, b, c) : point_triangle_signed_distance(p, a, b, c);
}

struct PropStats {
  uint64_t init_steps, prop_steps, seeds;
};

// Lock-free equivalent of "take the cell's write lock, re-test, store" (grid.rs:447-454,
// 532-536): CAS on the f32 bit pattern, retried while compare_distances still says Less.
inline int relax_cell(std::atomic<uint32_t>* slot, float d, bool* stored) {
  uint32_t cur = slot->load(std::memory_order_relaxed);
  *stored = false;
  for (;;) {
    float curf;
    std::memcpy(&curf, &cur, 4);
    int c = compare_distances(d, curf);
    if (c == -2) return -2;
    if (c != -1) return 0;
    uint32_t nb;
    std::memcpy(&nb, &d, 4);
    if (slot->compare_exchange_weak(cur, nb, std::memory_order_relaxed)) {
      *stored = true;
      return 0;
    }
  }
}

int grid_propagate(const Mesh& m, const Grid& g, int sign, int heaps, int threads, float* out, PropStats* st) {
  const uint64_t total = g.count[0] * g.count[1] * g.count[2];
  const size_t T = m.ntri();
  std::atomic<int> err{0};
  std::atomic<uint64_t> init_steps{0}, prop_steps{0};

  // preheap: per cell (triangle, distance), grid.rs:118-124.  Packed as 64 bits
  // (tri index << 32 | f32 bits) so the lock-protected pair update becomes one CAS.
  std::vector<std::atomic<uint64_t>> preheap(total);
  {
    uint32_t mxb;
    std::memcpy(&mxb, &F32_MAX, 4);
    parallel_for(total, threads, [&](size_t i, int) { preheap[i].store((uint64_t)mxb, std::memory_order_relaxed); });
  }
  // PHASE 1 generate_preheap, grid.rs:383-457
  parallel_for(T, threads, [&](size_t t, int) {
    V3 a = m.v(t, 0), b = m.v(t, 1), c = m.v(t, 2);
    V3 mn, mx;
    triangle_bounding_box(a, b, c, &mn, &mx);
    uint64_t lo[3], hi[3];
    grid_snap(g, mn, lo);
    grid_snap(g, mx, hi);
    V3 lo_f = grid_cell_center(g, lo);
    for (int i = 0; i < 3; ++i)
      if (lo[i] > 0 && v_get(lo_f, i) > v_get(mn, i)) lo[i] -= 1;   // grid.rs:413-417
    V3 hi_f = grid_cell_center(g, hi);
    for (int i = 0; i < 3; ++i)
      if (hi[i] < g.count[i] - 1 && v_get(hi_f, i) < v_get(mx, i)) hi[i] += 1;  // grid.rs:420-426
    uint64_t steps = 0;
    for (uint64_t x = lo[0]; x <= hi[0]; ++x)
      for (uint64_t y = lo[1]; y <= hi[1]; ++y)
        for (uint64_t z = lo[2]; z <= hi[2]; ++z) {
          uint64_t cell[3] = {x, y, z};
          uint64_t idx = grid_cell_idx(g, cell);
          V3 p = grid_cell_center(g, cell);
          float d = sign == 0 ? point_triangle_distance(p, a, b, c) : point_triangle_signed_distance(p, a, b, c);
          uint64_t cur = preheap[idx].load(std::memory_order_relaxed);
          for (;;) {
            uint32_t cb = (uint32_t)cur;
            float curf;
            std::memcpy(&curf, &cb, 4);
            int cmp = compare_distances(d, curf);
            if (cmp == -2) { err.store(-2); break; }
            if (cmp != -1) break;
            uint32_t nb;
            std::memcpy(&nb, &d, 4);
            uint64_t nv = ((uint64_t)t << 32) | nb;
            if (preheap[idx].compare_exchange_weak(cur, nv, std::memory_order_relaxed)) { ++steps; break; }
          }
        }
    init_steps.fetch_add(steps, std::memory_order_relaxed);
  });
  if (err.load()) return err.load();

  // generate_heap, grid.rs:464-490 (serial scan, then sort)
  std::vector<std::atomic<uint32_t>> dist(total);
  std::vector<State> seeds;
  for (uint64_t i = 0; i < total; ++i) {
    uint64_t pv = preheap[i].load(std::memory_order_relaxed);
    uint32_t db = (uint32_t)pv;
    float d;
    std::memcpy(&d, &db, 4);
    uint32_t mxb;
    std::memcpy(&mxb, &F32_MAX, 4);
    dist[i].store(mxb, std::memory_order_relaxed);
    if (d < F32_MAX) {
      dist[i].store(db, std::memory_order_relaxed);
      State s;
      s.distance = d;
      grid_cell_coords(g, i, s.cell);
      size_t t = (size_t)(pv >> 32);
      s.tri[0] = m.tris[3 * t];
      s.tri[1] = m.tris[3 * t + 1];
      s.tri[2] = m.tris[3 * t + 2];
      seeds.push_back(s);
    }
  }
  { std::vector<std::atomic<uint64_t>>().swap(preheap); }
  std::sort(seeds.begin(), seeds.end(), [](const State& a, const State& b) { return state_cmp(a, b) < 0; });
  if (st) st->seeds = seeds.size();

  // PHASE 2: round-robin split into `heaps` heaps (grid.rs:322-330), one thread each.
  if (heaps < 1) heaps = 1;
  std::vector<RsHeap> hs((size_t)heaps);
  for (size_t i = 0; i < seeds.size(); ++i) hs[i % (size_t)heaps].d.push_back(seeds[i]);
  { std::vector<State>().swap(seeds); }
  // the reference pops the LAST vec first for the first spawned thread; irrelevant to results.
  auto run_heap = [&](RsHeap& h) {
    h.rebuild();
    State s;
    uint64_t steps = 0;
    while (h.pop(&s)) {                                    // propagate_heap, grid.rs:495-558
      ++steps;
      for (int dx = -1; dx <= 1; ++dx)
        for (int dy = -1; dy <= 1; ++dy)
          for (int dz = -1; dz <= 1; ++dz) {
            int64_t x = (int64_t)s.cell[0] + dx, y = (int64_t)s.cell[1] + dy, z = (int64_t)s.cell[2] + dz;
            if (x < 0 || y < 0 || z < 0 || x >= (int64_t)g.count[0] || y >= (int64_t)g.count[1] || z >= (int64_t)g.count[2])
              continue;
            uint64_t nc[3] = {(uint64_t)x, (uint64_t)y, (uint64_t)z};
            V3 p = grid_cell_center(g, nc);
            uint64_t idx = grid_cell_idx(g, nc);
            float d = grid_tri_distance(m, s.tri, p, sign);
            bool stored;
            int rc = relax_cell(&dist[idx], d, &stored);
            if (rc) { err.store(rc); return; }
            if (stored) {
              State ns;
              ns.distance = d;
              ns.cell[0] = nc[0]; ns.cell[1] = nc[1]; ns.cell[2] = nc[2];
              ns.tri[0] = s.tri[0]; ns.tri[1] = s.tri[1]; ns.tri[2] = s.tri[2];
              h.push(ns);
            }
          }
    }
    prop_steps.fetch_add(steps, std::memory_order_relaxed);
  };
  if (threads <= 1) {
    // deterministic: heaps processed one after another (a legal interleaving of grid.rs:318-339)
    for (auto& h : hs) run_heap(h);
  } else {
    std::vector<std::thread> pool;
    for (auto& h : hs) pool.emplace_back([&run_heap, &h] { run_heap(h); });
    for (auto& th : pool) th.join();
  }
  if (err.load()) return err.load();

  parallel_for(total, threads, [&](size_t i, int) {      // grid.rs:350
    uint32_t b = dist[i].load(std::memory_order_relaxed);
    std::memcpy(&out[i], &b, 4);
  });
  if (st) {
    st->init_steps = init_steps.load();
    st->prop_steps = prop_steps.load();
  }
  return 0;
}


// ---- oracle-side acceleration (NOT part of the reference) ---------------------------
// A plain median-split BVH used only to make the EXACT semantics affordable at 100k+
// triangles.  Pruning is conservative (box bounds evaluated in double, with slack), so the
// answers are those of the brute-force loops above; tests/test_oracle_fast.py checks that
// bit for bit.
struct CpuBvh {
  struct Node {
    float mn[3], mx[3];
    int32_t left, right;   // children, or -1
    uint32_t first, count; // leaf range in `order`
  };
  std::vector<Node> nodes;
  std::vector<uint32_t> order;
  std::vector<V3> tmn, tmx;  // padded triangle boxes (geo.rs:4-22)
  double scale = 1.0;

  void build(const Mesh& m) {
    const size_t T = m.ntri();
    order.resize(T);
    tmn.resize(T);
    tmx.resize(T);
    std::vector<V3> cen(T);
    double mxabs = 0.0;
    for (size_t t = 0; t < T; ++t) {
      order[t] = (uint32_t)t;
      triangle_bounding_box(m.v(t, 0), m.v(t, 1), m.v(t, 2), &tmn[t], &tmx[t]);
      cen[t] = {0.5f * (tmn[t].x + tmx[t].x), 0.5f * (tmn[t].y + tmx[t].y), 0.5f * (tmn[t].z + tmx[t].z)};
      for (int k = 0; k < 3; ++k) {
        mxabs = std::max(mxabs, (double)std::fabs(v_get(tmn[t], k)));
        mxabs = std::max(mxabs, (double)std::fabs(v_get(tmx[t], k)));
      }
    }
    scale = mxabs;
    nodes.clear();
    if (T == 0) return;
    nodes.reserve(2 * T);
    build_rec(0, (uint32_t)T, cen);
  }
  int32_t build_rec(uint32_t first, uint32_t count, const std::vector<V3>& cen) {
    Node n;
    for (int k = 0; k < 3; ++k) { n.mn[k] = F32_MAX; n.mx[k] = -F32_MAX; }
    float cmn[3] = {F32_MAX, F32_MAX, F32_MAX}, cmx[3] = {-F32_MAX, -F32_MAX, -F32_MAX};
    for (uint32_t i = first; i < first + count; ++i) {
      uint32_t t = order[i];
      for (int k = 0; k < 3; ++k) {
        n.mn[k] = std::min(n.mn[k], v_get(tmn[t], k));
        n.mx[k] = std::max(n.mx[k], v_get(tmx[t], k));
        cmn[k] = std::min(cmn[k], v_get(cen[t], k));
        cmx[k] = std::max(cmx[k], v_get(cen[t], k));
      }
    }
    n.left = n.right = -1;
    n.first = first;
    n.count = count;
    int32_t id = (int32_t)nodes.size();
    nodes.push_back(n);
    if (count > 4) {
      int ax = 0;
      if (cmx[1] - cmn[1] > cmx[ax] - cmn[ax]) ax = 1;
      if (cmx[2] - cmn[2] > cmx[ax] - cmn[ax]) ax = 2;
      uint32_t mid = first + count / 2;
      std::nth_element(order.begin() + first, order.begin() + mid, order.begin() + first + count,
                       [&](uint32_t a, uint32_t b) { return v_get(cen[a], ax) < v_get(cen[b], ax); });
      int32_t l = build_rec(first, mid - first, cen);
      int32_t r = build_rec(mid, first + count - mid, cen);
      nodes[id].left = l;
      nodes[id].right = r;
    }
    return id;
  }
  static double box_dist2(const float* mn, const float* mx, V3 p) {
    double d2 = 0.0;
    for (int k = 0; k < 3; ++k) {
      double v = v_get(p, k);
      double d = std::max(std::max((double)mn[k] - v, v - (double)mx[k]), 0.0);
      d2 += d * d;
    }
    return d2;
  }
  // Visit every triangle whose padded box is within `radius(best)` of p; `best` shrinks as f reports.
  template <class F>
  void nearest(V3 p, F f /* float f(uint32_t tri) -> |distance| */) const {
    if (nodes.empty()) return;
    double best = std::numeric_limits<double>::infinity();
    auto bound2 = [&](double b) {
      double pm = std::max(scale, (double)std::max(std::fabs(p.x), std::max(std::fabs(p.y), std::fabs(p.z))));
      double r = b * (1.0 + 1e-4) + 1e-4 * (1.0 + pm * 1e-2) + 2e-6;
      return r * r;
    };
    int32_t stack[128];
    int sp = 0;
    stack[sp++] = 0;
    while (sp) {
      const Node& n = nodes[stack[--sp]];
      if (box_dist2(n.mn, n.mx, p) > bound2(best)) continue;
      if (n.left < 0) {
        for (uint32_t i = n.first; i < n.first + n.count; ++i) {
          float d = f(order[i]);
          if (d == d && (double)d < best) best = d;
        }
      } else {
        double dl = box_dist2(nodes[n.left].mn, nodes[n.left].mx, p), dr = box_dist2(nodes[n.right].mn, nodes[n.right].mx, p);
        if (dl < dr) { stack[sp++] = n.right; stack[sp++] = n.left; }
        else { stack[sp++] = n.left; stack[sp++] = n.right; }
      }
    }
  }
  // Visit every triangle whose padded box meets the +axis ray from o (closed rule above).
  template <class F>
  void stab(V3 o, int axis, F f) const {
    if (nodes.empty()) return;
    int32_t stack[128];
    int sp = 0;
    stack[sp++] = 0;
    while (sp) {
      const Node& n = nodes[stack[--sp]];
      V3 mn = {n.mn[0], n.mn[1], n.mn[2]}, mx = {n.mx[0], n.mx[1], n.mx[2]};
      if (!ray_meets_padded_aabb(o, mn, mx, axis)) continue;
      if (n.left < 0) {
        for (uint32_t i = n.first; i < n.first + n.count; ++i) {
          uint32_t t = order[i];
          if (ray_meets_padded_aabb(o, tmn[t], tmx[t], axis)) f(t);
        }
      } else {
        stack[sp++] = n.left;
        stack[sp++] = n.right;
      }
    }
  }
};

// Same contract as query_one, evaluated through the CpuBvh.
int query_one_fast(const Mesh& m, const CpuBvh& bvh, V3 q, int accel, int sign, float* out) {
  const size_t T = m.ntri();
  auto tri_dist = [&](uint32_t t) { return point_triangle_distance(q, m.v(t, 0), m.v(t, 1), m.v(t, 2)); };
  auto rays3 = [&]() {
    int insides = 0;
    for (int axis = 0; axis < 3; ++axis) {
      uint32_t cnt = 0;
      bvh.stab(q, axis, [&](uint32_t t) {
        float tt;
        if (ray_triangle_intersection_aligned(q, m.v(t, 0), m.v(t, 1), m.v(t, 2), axis, &tt)) ++cnt;
      });
      if (cnt % 2 == 1) ++insides;
    }
    return insides;
  };
  if ((accel == 0 || accel == 1) && sign == 1) {
    // Normal fold: literal compare_distances fold, index order, restricted to the triangles
    // whose magnitude is within a generous window of the minimum (the fold's result only
    // depends on those; checked against the full fold in tests).
    std::vector<std::pair<uint32_t, float>> cand;
    bool nan = false;
    bvh.nearest(q, [&](uint32_t t) {
      float d = point_triangle_signed_distance(q, m.v(t, 0), m.v(t, 1), m.v(t, 2));
      if (!(d == d)) nan = true;
      cand.push_back({t, d});
      return std::fabs(d);
    });
    if (nan) return -2;
    std::sort(cand.begin(), cand.end());
    float mind = F32_MAX;
    for (auto& c : cand) {
      int r = accel == 0 ? compare_distances(c.second, mind) : -compare_distances(mind, c.second);
      if (r == -2 || r == 2) return -2;
      if (r == -1) mind = c.second;
    }
    *out = mind;
    return 0;
  }
  if (accel == 0) {
    // None(Raycast): +X parity over ALL triangles with no box filter (default.rs:32-38);
    // the unfiltered count is only affordable brute force.
    return query_one(m, q, accel, sign, out);
  }
  if (accel == 1) {
    float mind = F32_MAX;
    bvh.nearest(q, [&](uint32_t t) { float d = tri_dist(t); mind = rs_min(mind, d); return d; });
    *out = rays3() > 1 ? -mind : mind;
    return 0;
  }
  if (T == 0) return -3;
  // Rtree / RtreeBvh: lowest index among the minimal distance_2 (see query_one).
  bool have = false;
  uint32_t best = 0;
  float best_d2 = 0.0f;
  bvh.nearest(q, [&](uint32_t t) {
    V3 a = m.v(t, 0), b = m.v(t, 1), c = m.v(t, 2);
    float d2 = point_triangle_distance2(q, a, b, c);
    if (d2 == d2 && (!have || d2 < best_d2 || (d2 == best_d2 && t < best))) { have = true; best = t; best_d2 = d2; }
    return point_triangle_distance(q, a, b, c);
  });
  V3 a = m.v(best, 0), b = m.v(best, 1), c = m.v(best, 2);
  if (accel == 2) { *out = point_triangle_signed_distance(q, a, b, c); return 0; }
  float dist = point_triangle_distance(q, a, b, c);
  *out = rays3() > 1 ? -dist : dist;
  return 0;
}

}  // namespace

// =====================================================================================
extern "C" {

void orc_closest_point_triangle(const float* p, const float* a, const float* b, const float* c, float* out) {
  st3(out, closest_point_triangle(ld3(p), ld3(a), ld3(b), ld3(c)));
}
void orc_closest_point_segment(const float* p, const float* a, const float* b, float* out) {
  st3(out, closest_point_segment(ld3(p), ld3(a), ld3(b)));
}
float orc_point_triangle_distance(const float* p, const float* a, const float* b, const float* c) {
  return point_triangle_distance(ld3(p), ld3(a), ld3(b), ld3(c));
}
float orc_point_triangle_distance2(const float* p, const float* a, const float* b, const float* c) {
  return point_triangle_distance2(ld3(p), ld3(a), ld3(b), ld3(c));
}
float orc_point_triangle_signed_distance(const float* p, const float* a, const float* b, const float* c) {
  return point_triangle_signed_distance(ld3(p), ld3(a), ld3(b), ld3(c));
}
int orc_ray_triangle_intersection_aligned(const float* o, const float* a, const float* b, const float* c, int axis, float* t) {
  float tt = 0.0f;
  bool hit = ray_triangle_intersection_aligned(ld3(o), ld3(a), ld3(b), ld3(c), axis, &tt);
  if (hit && t) *t = tt;
  return hit ? 1 : 0;
}
void orc_triangle_bounding_box(const float* a, const float* b, const float* c, float* mn, float* mx) {
  V3 lo, hi;
  triangle_bounding_box(ld3(a), ld3(b), ld3(c), &lo, &hi);
  st3(mn, lo);
  st3(mx, hi);
}
int orc_compare_distances(float a, float b) { return compare_distances(a, b); }
int orc_approx_eq(float a, float b, int ulps, float eps) { return approx_eq_f32(a, b, ulps, eps) ? 1 : 0; }
float orc_length(const float* a) { return v_length(ld3(a)); }
float orc_dist(const float* a, const float* b) { return v_dist(ld3(a), ld3(b)); }
float orc_dot(const float* a, const float* b) { return v_dot(ld3(a), ld3(b)); }

void orc_grid_from_bounding_box(const float* bmin, const float* bmax, const uint64_t* count, float* first, float* size) {
  Grid g = grid_from_bounding_box(ld3(bmin), ld3(bmax), count);
  st3(first, g.first_cell);
  st3(size, g.cell_size);
}
static Grid mk_grid(const float* first, const float* size, const uint64_t* count) {
  return {ld3(first), ld3(size), {count[0], count[1], count[2]}};
}
void orc_grid_bounding_box(const float* first, const float* size, const uint64_t* count, float* mn, float* mx) {
  Grid g = mk_grid(first, size, count);
  V3 lo = grid_bbox_min(g);  // grid.rs:110-119
  st3(mn, lo);
  mx[0] = lo.x + (float)count[0] * g.cell_size.x;
  mx[1] = lo.y + (float)count[1] * g.cell_size.y;
  mx[2] = lo.z + (float)count[2] * g.cell_size.z;
}
uint64_t orc_grid_cell_idx(const uint64_t* count, const uint64_t* cell) {
  Grid g = {{0, 0, 0}, {0, 0, 0}, {count[0], count[1], count[2]}};
  return grid_cell_idx(g, cell);
}
void orc_grid_cell_coords(const uint64_t* count, uint64_t idx, uint64_t* cell) {
  Grid g = {{0, 0, 0}, {0, 0, 0}, {count[0], count[1], count[2]}};
  grid_cell_coords(g, idx, cell);
}
void orc_grid_cell_center(const float* first, const float* size, const uint64_t* count, const uint64_t* cell, float* out) {
  st3(out, grid_cell_center(mk_grid(first, size, count), cell));
}
int orc_grid_snap(const float* first, const float* size, const uint64_t* count, const float* p, uint64_t* cell) {
  return grid_snap(mk_grid(first, size, count), ld3(p), cell) ? 1 : 0;
}

// Flattened triangle count / list (lib.rs:175-193).  tris_out may be NULL to query the count.
int64_t orc_get_triangles(size_t n_verts, const uint32_t* indices, size_t n_indices, int topology, uint32_t* tris_out) {
  std::vector<uint32_t> t;
  int rc = get_triangles(n_verts, indices, n_indices, topology, &t);
  if (rc) return rc;
  if (tris_out) std::memcpy(tris_out, t.data(), t.size() * 4);
  return (int64_t)(t.size() / 3);
}

// generate_sdf (lib.rs:291-311).  Returns the number of distances written (n_q, or 0 for
// RtreeBvh on an empty mesh, rtree_bvh.rs:104-106) or a negative error:
// -1 index out of range, -2 NaN panic, -3 empty mesh panic (Rtree).
int64_t orc_generate_sdf(const float* verts, size_t n_verts, const uint32_t* indices, size_t n_indices, int topology,
                         const float* queries, size_t n_q, int accel, int sign, int threads, float* out) {
  Mesh m;
  m.verts = verts;
  if (get_triangles(n_verts, indices, n_indices, topology, &m.tris)) return -1;
  if (accel == 3 && m.ntri() == 0) return 0;
  if (accel == 2 && m.ntri() == 0 && n_q > 0) return -3;
  std::atomic<int> err{0};
  parallel_for(n_q, threads, [&](size_t i, int) {
    int rc = query_one(m, ld3(queries + 3 * i), accel, sign, &out[i]);
    if (rc) err.store(rc);
  });
  if (err.load()) return err.load();
  return (int64_t)n_q;
}

// generate_sdf through the oracle-side BVH: same contract and answers as orc_generate_sdf.
int64_t orc_generate_sdf_fast(const float* verts, size_t n_verts, const uint32_t* indices, size_t n_indices, int topology,
                              const float* queries, size_t n_q, int accel, int sign, int threads, float* out) {
  Mesh m;
  m.verts = verts;
  if (get_triangles(n_verts, indices, n_indices, topology, &m.tris)) return -1;
  if (accel == 3 && m.ntri() == 0) return 0;
  if (accel == 2 && m.ntri() == 0 && n_q > 0) return -3;
  CpuBvh bvh;
  bvh.build(m);
  std::atomic<int> err{0};
  parallel_for(n_q, threads, [&](size_t i, int) {
    int rc = m.ntri() ? query_one_fast(m, bvh, ld3(queries + 3 * i), accel, sign, &out[i])
                      : query_one(m, ld3(queries + 3 * i), accel, sign, &out[i]);
    if (rc) err.store(rc);
  });
  if (err.load()) return err.load();
  return (int64_t)n_q;
}

// generate_grid_sdf (generate/grid.rs:265-378).
// semantics 2 = EXACT through the oracle-side BVH (identical answers to 0, for big meshes).
// semantics 0 = EXACT   : every cell = generate_sdf(None(sign)) magnitude at the cell centre (what
//                          generate/grid.rs:693-724 asserts the grid path to be), Raycast sign by
//                          the grid-line rule of grid.rs:568-642.
// semantics 1 = PROPAGATE: the reference's three phases restated; `heaps` = rayon::current_num_threads().
// threads <= 1 runs everything sequentially and deterministically.
int orc_generate_grid_sdf(const float* verts, size_t n_verts, const uint32_t* indices, size_t n_indices, int topology,
                          const float* first, const float* size, const uint64_t* count, int sign, int semantics,
                          int heaps, int threads, float* out, uint64_t* stats /*3, may be NULL*/) {
  Mesh m;
  m.verts = verts;
  if (get_triangles(n_verts, indices, n_indices, topology, &m.tris)) return -1;
  Grid g = mk_grid(first, size, count);
  const uint64_t total = count[0] * count[1] * count[2];
  if (semantics == 1) {
    PropStats st{0, 0, 0};
    int rc = grid_propagate(m, g, sign, heaps, threads, out, &st);
    if (rc) return rc;
    if (stats) { stats[0] = st.init_steps; stats[1] = st.prop_steps; stats[2] = st.seeds; }
  } else if (semantics == 2) {
    // EXACT semantics through the oracle-side BVH (same answers as semantics 0).
    CpuBvh bvh;
    bvh.build(m);
    std::atomic<int> err{0};
    parallel_for(total, threads, [&](size_t i, int) {
      uint64_t cell[3];
      grid_cell_coords(g, i, cell);
      V3 p = grid_cell_center(g, cell);
      if (m.ntri() == 0) { out[i] = F32_MAX; return; }
      float d;
      // grid magnitude == generate_sdf(None(sign)) magnitude; sign 0 magnitude via accel 1 path
      // without rays: reuse Bvh(Raycast) min and drop its sign.
      int rc;
      if (sign == 0) {
        float mind = F32_MAX;
        bvh.nearest(p, [&](uint32_t t) {
          float dd = point_triangle_distance(p, m.v(t, 0), m.v(t, 1), m.v(t, 2));
          mind = rs_min(mind, dd);
          return dd;
        });
        d = mind;
        rc = 0;
      } else {
        rc = query_one_fast(m, bvh, p, 0, 1, &d);
      }
      if (rc) { err.store(rc); return; }
      out[i] = d;
    });
    if (err.load()) return err.load();
  } else {
    std::atomic<int> err{0};
    parallel_for(total, threads, [&](size_t i, int) {
      uint64_t cell[3];
      grid_cell_coords(g, i, cell);
      V3 p = grid_cell_center(g, cell);
      float mind = F32_MAX;
      if (sign == 0) {
        for (size_t t = 0; t < m.ntri(); ++t)
          mind = rs_min(mind, point_triangle_distance(p, m.v(t, 0), m.v(t, 1), m.v(t, 2)));
      } else {
        for (size_t t = 0; t < m.ntri(); ++t) {
          float d = point_triangle_signed_distance(p, m.v(t, 0), m.v(t, 1), m.v(t, 2));
          int c = compare_distances(d, mind);
          if (c == -2) { err.store(-2); return; }
          if (c == -1) mind = d;
        }
      }
      out[i] = mind;
    });
    if (err.load()) return err.load();
  }
  if (sign == 0) {
    std::vector<uint8_t> inter;
    grid_ray_parity(m, g, threads, &inter);
    apply_parity_sign(inter, out, total, threads);
  }
  return 0;
}

// Per-voxel, per-axis hit parity of the grid-line rule (for tests of the sign kernels).
int orc_grid_ray_parity(const float* verts, size_t n_verts, const uint32_t* indices, size_t n_indices, int topology,
                        const float* first, const float* size, const uint64_t* count, int threads, uint8_t* out3) {
  Mesh m;
  m.verts = verts;
  if (get_triangles(n_verts, indices, n_indices, topology, &m.tris)) return -1;
  Grid g = mk_grid(first, size, count);
  std::vector<uint8_t> inter;
  grid_ray_parity(m, g, threads, &inter);
  std::memcpy(out3, inter.data(), inter.size());
  return 0;
}

int orc_hardware_threads() { return (int)std::thread::hardware_concurrency(); }

}  // extern "C"
