// geo_probe.hip — HOST build of the device math in geo.hip.h, for CPU unit tests only
// (tests/test_device_math_host.py).  Not linked into libm2s_hip.so.
#include "geo.hip.h"

using namespace m2s;

extern "C" {
float probe_dist2(const float* p, const float* a, const float* b, const float* c) {
  f3 A = mk3(a[0], a[1], a[2]), B = mk3(b[0], b[1], b[2]), Cc = mk3(c[0], c[1], c[2]);
  return point_triangle_dist2(mk3(p[0], p[1], p[2]), A, B, Cc, tri_edges(A, B, Cc), tri_class(A, B, Cc));
}
float probe_dist2_signed(const float* p, const float* a, const float* b, const float* c, int* positive) {
  f3 A = mk3(a[0], a[1], a[2]), B = mk3(b[0], b[1], b[2]), Cc = mk3(c[0], c[1], c[2]);
  bool pos;
  float d2 = point_triangle_dist2_signed(mk3(p[0], p[1], p[2]), A, B, Cc, tri_edges(A, B, Cc), tri_class(A, B, Cc), &pos);
  *positive = pos ? 1 : 0;
  return d2;
}
int probe_ray(int axis, const float* o, const float* a, const float* b, const float* c, float* t) {
  return ray_triangle_aligned_rt(axis, mk3(o[0], o[1], o[2]), mk3(a[0], a[1], a[2]), mk3(b[0], b[1], b[2]),
                                 mk3(c[0], c[1], c[2]), t) ? 1 : 0;
}
float probe_normal_fold_result(float d2_all, float d2_pos) { return normal_fold_result(d2_all, d2_pos); }
int probe_approx_eq_abs(float a, float b) { return approx_eq_abs(a, b) ? 1 : 0; }
void probe_tri_box(const float* a, const float* b, const float* c, float* mn, float* mx) {
  f3 lo, hi;
  triangle_bounding_box(mk3(a[0], a[1], a[2]), mk3(b[0], b[1], b[2]), mk3(c[0], c[1], c[2]), &lo, &hi);
  mn[0] = lo.x; mn[1] = lo.y; mn[2] = lo.z; mx[0] = hi.x; mx[1] = hi.y; mx[2] = hi.z;
}
}
