// geo.hip.h — device restatement of the reference's f32 geometry for gfx950 (wave64).
//
// Numerics contract: every expression that reaches the output is IEEE
// binary32 in the reference's operation order with NO fused multiply-add: this TU is compiled
// with -ffp-contract=off and the pragma below; `/` and sqrtf are the correctly rounded forms
// (hipcc default).  Where an FMA is wanted (conservative bounds only) it is written explicitly
// as __builtin_fmaf, which -ffp-contract=off does not touch.
//
// The code is written select-style instead of the reference's early returns so that the 64
// lanes of a wave, which all test the SAME triangle against 64 DIFFERENT points, never diverge;
// the predicates and the arithmetic feeding each result are those of geo.rs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#pragma clang fp contract(off)

#define M2S_HD __host__ __device__ __forceinline__

namespace m2s {

struct f3 {
  float x, y, z;
};

M2S_HD f3 mk3(float x, float y, float z) { return {x, y, z}; }
// point.rs:81-141 — operation order matters (x*x' + y*y' + z*z', left to right)
M2S_HD f3 add3(f3 a, f3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
M2S_HD f3 sub3(f3 a, f3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
M2S_HD float dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
M2S_HD f3 cross3(f3 a, f3 b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
M2S_HD f3 fmul3(f3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
M2S_HD f3 sel3(bool c, f3 a, f3 b) { return {c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z}; }
M2S_HD bool eq3(f3 a, f3 b) { return a.x == b.x && a.y == b.y && a.z == b.z; }

// Triangle degeneracy class, decided once per triangle on the host side of the kernel
// (geo.rs:73-88: the match on (a==b, b==c, a==c)).
enum : uint32_t {
  TRI_REGULAR = 0,
  TRI_POINT = 1,   // a == b == c            -> closest point is a                     geo.rs:74-76
  TRI_SEG_AC = 2,  // a == b                 -> closest_point_segment(p, a, c)         geo.rs:77-79
  TRI_SEG_AB = 3,  // b == c  or  a == c     -> closest_point_segment(p, a, b)         geo.rs:80-85
};
M2S_HD uint32_t tri_class(f3 a, f3 b, f3 c) {
  const bool ab = eq3(a, b), bc = eq3(b, c), ac = eq3(a, c);
  if (ab && bc && ac) return TRI_POINT;
  if (ab) return TRI_SEG_AC;
  if (bc || ac) return TRI_SEG_AB;
  return TRI_REGULAR;
}

// geo.rs:141-151
M2S_HD f3 closest_point_segment(f3 p, f3 a, f3 ab /* = b.sub(a) */) {
  float m = dot3(ab, ab);
  f3 ap = sub3(p, a);
  float s12 = dot3(ab, ap) / m;
  s12 = (s12 < 0.0f) ? 0.0f : ((s12 > 1.0f) ? 1.0f : s12);  // f32::clamp, NaN passes through
  return add3(a, fmul3(ab, s12));
}

// geo.rs:90-137 for a non-degenerate-class triangle.  One IEEE division per call, as in the
// reference (each region divides once; the numerator / denominator pair is selected first).
// In two halves, so that the packet walk can stop after the first one when every lane that still matters lies in a
// vertex region (distance.hip eval_triangle_leaf): the head is the six dot products and the three vertex-region tests
// the reference makes first (geo.rs:97, 104, 111), the tail the edge / interior regions.
struct RegularHead {
  float d1, d2, d3, d4, d5, d6;
  bool rA, rB, rC;
};
M2S_HD RegularHead closest_point_regular_head(f3 p, f3 a, f3 b, f3 c, f3 ab, f3 ac) {
  RegularHead h;
  const f3 ap = sub3(p, a);
  h.d1 = dot3(ab, ap);
  h.d2 = dot3(ac, ap);
  const f3 bp = sub3(p, b);
  h.d3 = dot3(ab, bp);
  h.d4 = dot3(ac, bp);
  const f3 cp = sub3(p, c);
  h.d5 = dot3(ab, cp);
  h.d6 = dot3(ac, cp);
  h.rA = (h.d1 <= 0.0f) & (h.d2 <= 0.0f);                           // geo.rs:97
  h.rB = (h.d3 >= 0.0f) & (h.d4 <= h.d3);                           // geo.rs:104
  h.rC = (h.d6 >= 0.0f) & (h.d5 <= h.d6);                           // geo.rs:111
  return h;
}
// The vertex a lane's closest point is, for lanes with rA | rB | rC (same precedence as the full selection below).
M2S_HD f3 closest_point_regular_vertex(const RegularHead& h, f3 a, f3 b, f3 c) { return sel3(h.rA, a, sel3(h.rB, b, c)); }
M2S_HD f3 closest_point_regular_tail(const RegularHead& h, f3 a, f3 b, f3 c, f3 ab, f3 ac, f3 bc) {
  const float d1 = h.d1, d2 = h.d2, d3 = h.d3, d4 = h.d4, d5 = h.d5, d6 = h.d6;
  const bool rA = h.rA, rB = h.rB, rC = h.rC;
  const float vc = d1 * d4 - d3 * d2;                                // geo.rs:115
  const bool rAB = (vc <= 0.0f) & (d1 >= 0.0f) & (d3 <= 0.0f);       // geo.rs:116
  const float vb = d5 * d2 - d1 * d6;                                // geo.rs:121
  const bool rAC = (vb <= 0.0f) & (d2 >= 0.0f) & (d6 <= 0.0f);       // geo.rs:122
  const float va = d3 * d6 - d5 * d4;                                // geo.rs:127
  const float d43 = d4 - d3;
  const float d56 = d5 - d6;
  const bool rBC = (va <= 0.0f) & (d43 >= 0.0f) & (d56 >= 0.0f);     // geo.rs:128

  // numerator / denominator of the single division of the region that fires first
  //   AB: d1/(d1-d3)   AC: d2/(d2-d6)   BC: (d4-d3)/((d4-d3)+(d5-d6))   interior: 1/(va+vb+vc)
  float num = 1.0f, den = va + vb + vc;
  f3 base = a, dir = ab;
  if (rBC) { num = d43; den = d43 + d56; base = b; dir = bc; }
  if (rAC) { num = d2; den = d2 - d6; base = a; dir = ac; }
  if (rAB) { num = d1; den = d1 - d3; base = a; dir = ab; }
  const float r = num / den;
  const bool edge = rAB | rAC | rBC;
  const float s = edge ? r : vb * r;                                 // interior: v = vb * denom
  f3 q = add3(base, fmul3(dir, s));                                  // a + ab*v   (geo.rs:118/124/131/137)
  const float w = vc * r;                                            // interior: w = vc * denom
  const f3 qi = add3(q, fmul3(ac, w));                               // (a + ab*v) + ac*w
  q = sel3(edge, q, qi);
  q = sel3(rC, c, q);
  q = sel3(rB, b, q);
  q = sel3(rA, a, q);
  return q;
}
M2S_HD f3 closest_point_regular(f3 p, f3 a, f3 b, f3 c, f3 ab, f3 ac, f3 bc) {
  return closest_point_regular_tail(closest_point_regular_head(p, a, b, c, ab, ac), a, b, c, ab, ac, bc);
}

// Closest point for any triangle class (cls is wave-uniform: one triangle per wave step).
// ab = b.sub(a), ac = c.sub(a), bc = c.sub(b): the edge vectors geo.rs forms inside each call.
struct TriEdges {
  f3 ab, ac, bc;
};
M2S_HD TriEdges tri_edges(f3 a, f3 b, f3 c) { return {sub3(b, a), sub3(c, a), sub3(c, b)}; }

M2S_HD f3 closest_point_triangle(f3 p, f3 a, f3 b, f3 c, const TriEdges& e, uint32_t cls) {
  if (cls == TRI_REGULAR) return closest_point_regular(p, a, b, c, e.ab, e.ac, e.bc);
  if (cls == TRI_POINT) return a;
  if (cls == TRI_SEG_AC) return closest_point_segment(p, a, e.ac);
  return closest_point_segment(p, a, e.ab);
}

// geo.rs:33-37 — squared distance dot(p-n, p-n).  sqrt is monotone, so the minimum over
// triangles of geo.rs:26-30's sqrt(dot) is sqrt(min dot): kernels minimise d2 and take ONE
// correctly rounded sqrt at the end, which gives the identical f32.
M2S_HD float point_triangle_dist2(f3 p, f3 a, f3 b, f3 c, const TriEdges& e, uint32_t cls) {
  const f3 n = closest_point_triangle(p, a, b, c, e, cls);
  const f3 d = sub3(p, n);
  return dot3(d, d);
}

// geo.rs:43-56 — returns d2 and whether the reference's signed distance is positive
// (direction . ((b-a) x (c-a)) > 0, normal not normalised; == 0 counts as negative).
M2S_HD float point_triangle_dist2_signed(f3 p, f3 a, f3 b, f3 c, const TriEdges& e, uint32_t cls, bool* positive) {
  const f3 n = closest_point_triangle(p, a, b, c, e, cls);
  const f3 d = sub3(p, n);
  const f3 nrm = cross3(e.ab, e.ac);
  *positive = dot3(d, nrm) > 0.0f;
  return dot3(d, d);
}
// Same with the normal cross3(e.ab, e.ac) supplied by the caller (precomputed per triangle, bit-identical).
M2S_HD float point_triangle_dist2_signed_n(f3 p, f3 a, f3 b, f3 c, const TriEdges& e, uint32_t cls, f3 nrm, bool* positive) {
  const f3 n = closest_point_triangle(p, a, b, c, e, cls);
  const f3 d = sub3(p, n);
  *positive = dot3(d, nrm) > 0.0f;
  return dot3(d, d);
}

// geo.rs:165-216 — axis-aligned ray/triangle.  AXIS 0: ray +X, plane (y,z); 1: +Y, (z,x); 2: +Z, (x,y).
template <int AXIS>
M2S_HD float gx_(f3 v) { return AXIS == 0 ? v.x : (AXIS == 1 ? v.y : v.z); }
template <int AXIS>
M2S_HD float gy_(f3 v) { return AXIS == 0 ? v.y : (AXIS == 1 ? v.z : v.x); }
template <int AXIS>
M2S_HD float gz_(f3 v) { return AXIS == 0 ? v.z : (AXIS == 1 ? v.x : v.y); }

template <int AXIS>
M2S_HD bool ray_triangle_aligned(f3 o, f3 t0, f3 t1, f3 t2, float* t_out) {
  const f3 e01 = sub3(t1, t0), e12 = sub3(t2, t1), e20 = sub3(t0, t2);
  const f3 p0 = sub3(o, t0), p1 = sub3(o, t1), p2 = sub3(o, t2);
  const float w0 = gz_<AXIS>(p1) * gy_<AXIS>(e12) - gy_<AXIS>(p1) * gz_<AXIS>(e12);  // geo.rs:199
  const float w1 = gz_<AXIS>(p2) * gy_<AXIS>(e20) - gy_<AXIS>(p2) * gz_<AXIS>(e20);  // geo.rs:200
  const float w2 = gz_<AXIS>(p0) * gy_<AXIS>(e01) - gy_<AXIS>(p0) * gz_<AXIS>(e01);  // geo.rs:201
  const bool inside = ((w0 < 0.0f) & (w1 < 0.0f) & (w2 < 0.0f)) | ((w0 > 0.0f) & (w1 > 0.0f) & (w2 > 0.0f));  // :203
  const float t = -(w0 * gx_<AXIS>(p0) + w2 * gx_<AXIS>(p2) + w1 * gx_<AXIS>(p1)) / (w0 + w1 + w2);           // :208
  *t_out = t;
  return inside & (t > 0.0f);                                                                                 // :210
}
M2S_HD bool ray_triangle_aligned_rt(int axis, f3 o, f3 t0, f3 t1, f3 t2, float* t_out) {
  if (axis == 0) return ray_triangle_aligned<0>(o, t0, t1, t2, t_out);
  if (axis == 1) return ray_triangle_aligned<1>(o, t0, t1, t2, t_out);
  return ray_triangle_aligned<2>(o, t0, t1, t2, t_out);
}

// geo.rs:4-22 — triangle AABB padded by 1e-4 (f32::min/max: a NaN operand is dropped)
M2S_HD void triangle_bounding_box(f3 a, f3 b, f3 c, f3* mn, f3* mx) {
  const float e = 0.0001f;
  *mn = {fminf(a.x, fminf(b.x, c.x)) - e, fminf(a.y, fminf(b.y, c.y)) - e, fminf(a.z, fminf(b.z, c.z)) - e};
  *mx = {fmaxf(a.x, fmaxf(b.x, c.x)) + e, fmaxf(a.y, fmaxf(b.y, c.y)) + e, fmaxf(a.z, fmaxf(b.z, c.z)) + e};
}

// Candidate rule of bvh::traverse for an axis-aligned ray (closed padded box, see oracle).
template <int AXIS>
M2S_HD bool ray_meets_box(f3 o, f3 mn, f3 mx) {
  return (gy_<AXIS>(o) >= gy_<AXIS>(mn)) & (gy_<AXIS>(o) <= gy_<AXIS>(mx)) & (gz_<AXIS>(o) >= gz_<AXIS>(mn)) &
         (gz_<AXIS>(o) <= gz_<AXIS>(mx)) & (gx_<AXIS>(mx) >= gx_<AXIS>(o));
}

M2S_HD int32_t f32_bits(float f) {
#ifdef __HIP_DEVICE_COMPILE__
  return __float_as_int(f);
#else
  int32_t i;
  __builtin_memcpy(&i, &f, 4);
  return i;
#endif
}

// float-cmp approx_eq!(f32, a, b, ulps = 2, epsilon = 1e-6) for non-negative a, b (lib.rs:248)
M2S_HD bool approx_eq_abs(float a, float b) {
  const int32_t d = (int32_t)((uint32_t)f32_bits(a) - (uint32_t)f32_bits(b));
  const int32_t ad = d == INT32_MIN ? INT32_MAX : (d < 0 ? -d : d);
  return (a == b) | (fabsf(a - b) <= 1e-6f) | (ad <= 2);   // select-style on purpose, see normal_fold_result
}

// Final value of the compare_distances fold (lib.rs:242-259 applied as in default.rs:52-59)
// from the two running minima the kernels keep: min d2 over all triangles and min d2 over the
// triangles whose signed distance is positive.  With dmin = min|d| and P = {positive d with
// approx_eq(|d|, dmin)} the fold returns min(P) if P is non-empty, else -dmin
// (tests/test_oracle_fast.py::test_normal_fold_closed_form).
// Written with selects only: hipcc 7.2 (clang 22) dropped the `return -dall` path of the
// equivalent early-return form for lanes that failed approx_eq (caught by the parity tests).
M2S_HD float normal_fold_result(float d2_all, float d2_pos) {
  const float f32max = 3.402823466e+38f;
  const float inf = __builtin_inff();
  const float dall = sqrtf(d2_all);
  const float dpos = sqrtf(d2_pos);
  const bool use_pos = (d2_pos < inf) & approx_eq_abs(dpos, dall);
  float r = use_pos ? dpos : -dall;
  r = (d2_all < inf) ? r : f32max;   // nothing finite seen: the fold keeps f32::MAX
  return r;
}

// Grid::get_cell_center for one axis (grid.rs:135-141): first + (i as f32) * size, mul then add.
M2S_HD float cell_center(float first, float size, uint32_t i) { return first + (float)i * size; }

}  // namespace m2s
