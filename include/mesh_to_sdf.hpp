// mesh_to_sdf.hpp — C++17 host-side mirror of the reference crate's public interface over the C ABI (m2s.h).
//
// The reference is compiled code (Rust, crate mesh_to_sdf 0.4.0); its toolchain is absent from the build image,
// so this header is the host side a C++ caller — or a maintainer porting call sites — uses: same names, same
// argument meaning, same error behaviour (a reference panic is a mesh_to_sdf::Panic exception here):
//
//   Topology<I>            mesh_to_sdf/src/lib.rs:151-193     TriangleList / TriangleStrip, optional indices
//   SignMethod             lib.rs:204-216                      Raycast (default) | Normal
//   AccelerationMethod     lib.rs:224-239                      None(sign) | Bvh(sign) | Rtree | RtreeBvh (default)
//   generate_sdf           lib.rs:291-311
//   Grid<V>                grid.rs:30-170                      new_ / from_bounding_box / getters / snap_point_to_grid
//   generate_grid_sdf      generate/grid.rs:265-378
//   serde::*               serde.rs:75-221                     SerializeSdf / DeserializeSdf / save_to_file / read_from_file
//
// V is any point type with x(), y(), z() | .x .y .z | operator[] (the reference's `Point` trait adapters,
// point/impl_*.rs: [f32;3], glam, cgmath, nalgebra, mint all reduce to three f32).  Header only; link -lm2s_hip.
#pragma once
#include <array>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>
#include <variant>
#include <vector>

#include "m2s.h"

namespace mesh_to_sdf {

// ---- errors ------------------------------------------------------------------------------------------
struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string& m) : std::runtime_error("m2s error " + std::to_string(c) + ": " + m), code(c) {}
};
// The reference would have panicked here: index out of range, "NaN distance" (lib.rs:257), Rtree on an empty mesh.
struct Panic : Error { using Error::Error; };

namespace detail {
inline void check(int rc) {
  if (rc == M2S_OK) return;
  const std::string msg = m2s_last_error();
  if (rc == M2S_ERR_BAD_ARG || rc == M2S_ERR_NAN || rc == M2S_ERR_EMPTY_MESH) throw Panic(rc, msg);
  throw Error(rc, msg);
}

// ---- Point access (point.rs:11-76: the trait needs new / x / y / z) -------------------------------------
template <class V, class = void> struct has_xyz_fn : std::false_type {};
template <class V> struct has_xyz_fn<V, std::void_t<decltype(std::declval<const V&>().x()), decltype(std::declval<const V&>().z())>> : std::true_type {};
template <class V, class = void> struct has_xyz_mem : std::false_type {};
template <class V> struct has_xyz_mem<V, std::void_t<decltype(std::declval<const V&>().x + std::declval<const V&>().z)>> : std::true_type {};

template <class V> float px(const V& v) { if constexpr (has_xyz_fn<V>::value) return v.x(); else if constexpr (has_xyz_mem<V>::value) return v.x; else return v[0]; }
template <class V> float py(const V& v) { if constexpr (has_xyz_fn<V>::value) return v.y(); else if constexpr (has_xyz_mem<V>::value) return v.y; else return v[1]; }
template <class V> float pz(const V& v) { if constexpr (has_xyz_fn<V>::value) return v.z(); else if constexpr (has_xyz_mem<V>::value) return v.z; else return v[2]; }
template <class V> V make_point(float x, float y, float z) {
  if constexpr (std::is_constructible_v<V, float, float, float>) return V(x, y, z);
  else return V{x, y, z};
}

// Packed xyz view of a point array: zero-copy when V already is three consecutive floats.
template <class V>
struct Packed {
  const float* ptr = nullptr;
  std::vector<float> copy;
  Packed(const V* p, size_t n) {
    if constexpr (sizeof(V) == 12 && std::is_trivially_copyable_v<V> && !has_xyz_fn<V>::value) {
      ptr = reinterpret_cast<const float*>(p);
    } else {
      copy.resize(3 * n);
      for (size_t i = 0; i < n; ++i) { copy[3 * i] = px(p[i]); copy[3 * i + 1] = py(p[i]); copy[3 * i + 2] = pz(p[i]); }
      ptr = copy.data();
    }
  }
};
}  // namespace detail

// ---- enums ---------------------------------------------------------------------------------------------
enum class SignMethod : int { Raycast = M2S_SIGN_RAYCAST, Normal = M2S_SIGN_NORMAL };   // default Raycast

struct AccelerationMethod {
  int kind = M2S_ACCEL_RTREE_BVH;                 // default RtreeBvh (lib.rs:233)
  SignMethod sign = SignMethod::Raycast;
  static AccelerationMethod None(SignMethod s = SignMethod::Raycast) { return {M2S_ACCEL_NONE, s}; }
  static AccelerationMethod Bvh(SignMethod s = SignMethod::Raycast) { return {M2S_ACCEL_BVH, s}; }
  static AccelerationMethod Rtree() { return {M2S_ACCEL_RTREE, SignMethod::Normal}; }
  static AccelerationMethod RtreeBvh() { return {M2S_ACCEL_RTREE_BVH, SignMethod::Raycast}; }
};

template <class I = uint32_t>
struct Topology {
  int kind = M2S_TRIANGLE_LIST;
  const I* indices = nullptr;   // nullptr == None: 0..vertices.len()
  size_t count = 0;
  static Topology TriangleList() { return {M2S_TRIANGLE_LIST, nullptr, 0}; }
  static Topology TriangleList(const I* idx, size_t n) { return {M2S_TRIANGLE_LIST, idx, n}; }
  static Topology TriangleList(const std::vector<I>& idx) { return {M2S_TRIANGLE_LIST, idx.data(), idx.size()}; }
  static Topology TriangleStrip() { return {M2S_TRIANGLE_STRIP, nullptr, 0}; }
  static Topology TriangleStrip(const I* idx, size_t n) { return {M2S_TRIANGLE_STRIP, idx, n}; }
  static Topology TriangleStrip(const std::vector<I>& idx) { return {M2S_TRIANGLE_STRIP, idx.data(), idx.size()}; }
  bool has_indices() const { return indices != nullptr || count != 0; }
};

namespace detail {
// I: Copy + Into<u32> — u16 and u32 pass through, anything else is widened once.
template <class I>
struct IndexArg {
  const void* ptr = nullptr;
  int bytes = 4;
  std::vector<uint32_t> widened;
  static uint32_t dummy() { return 0; }
  explicit IndexArg(const Topology<I>& t) {
    static const uint32_t kEmpty = 0;
    if (!t.has_indices()) return;                                  // None
    if (t.count == 0) { ptr = &kEmpty; return; }                   // Some(&[])
    if constexpr (std::is_integral_v<I> && (sizeof(I) == 2 || sizeof(I) == 4) && std::is_unsigned_v<I>) {
      ptr = t.indices;
      bytes = (int)sizeof(I);
    } else {
      widened.resize(t.count);
      for (size_t i = 0; i < t.count; ++i) widened[i] = static_cast<uint32_t>(t.indices[i]);
      ptr = widened.data();
    }
  }
};
}  // namespace detail

// ---- generate_sdf (lib.rs:291-311) ----------------------------------------------------------------------------
template <class V, class I = uint32_t>
std::vector<float> generate_sdf(const V* vertices, size_t n_vertices, const Topology<I>& indices, const V* query_points,
                                size_t n_queries, AccelerationMethod acceleration_method = AccelerationMethod()) {
  detail::Packed<V> v(vertices, n_vertices), q(query_points, n_queries);
  detail::IndexArg<I> ia(indices);
  std::vector<float> out(n_queries);
  size_t n_out = 0;
  detail::check(m2s_generate_sdf(v.ptr, n_vertices, ia.ptr, indices.count, ia.bytes, indices.kind, q.ptr, n_queries,
                                 acceleration_method.kind, (int)acceleration_method.sign, out.data(), &n_out, nullptr));
  out.resize(n_out);   // RtreeBvh on a mesh without triangles returns an empty Vec (generic/rtree_bvh.rs:104-106)
  return out;
}
template <class V, class I = uint32_t>
std::vector<float> generate_sdf(const std::vector<V>& vertices, const Topology<I>& indices, const std::vector<V>& query_points,
                                AccelerationMethod acceleration_method = AccelerationMethod()) {
  return generate_sdf(vertices.data(), vertices.size(), indices, query_points.data(), query_points.size(), acceleration_method);
}

// ---- Grid (grid.rs:30-170) ----------------------------------------------------------------------------------------
enum class SnapKind { Inside, Outside };   // SnapResult, grid.rs:10-17
struct SnapResult {
  SnapKind kind;
  std::array<size_t, 3> cell;
  bool operator==(const SnapResult& o) const { return kind == o.kind && cell == o.cell; }
};

template <class V>
class Grid {
 public:
  Grid() = default;
  static Grid new_(const V& first_cell, const V& cell_size, std::array<size_t, 3> cell_count) {   // grid.rs:43-49
    Grid g;
    const float f[3] = {detail::px(first_cell), detail::py(first_cell), detail::pz(first_cell)};
    const float s[3] = {detail::px(cell_size), detail::py(cell_size), detail::pz(cell_size)};
    for (int k = 0; k < 3; ++k) { g.g_.first_cell[k] = f[k]; g.g_.cell_size[k] = s[k]; g.g_.cell_count[k] = cell_count[k]; }
    return g;
  }
  static Grid from_bounding_box(const V& bbox_min, const V& bbox_max, std::array<size_t, 3> cell_count) {   // grid.rs:59-74
    Grid g;
    const float mn[3] = {detail::px(bbox_min), detail::py(bbox_min), detail::pz(bbox_min)};
    const float mx[3] = {detail::px(bbox_max), detail::py(bbox_max), detail::pz(bbox_max)};
    const uint64_t c[3] = {cell_count[0], cell_count[1], cell_count[2]};
    m2s_grid_from_bounding_box(mn, mx, c, &g.g_);
    return g;
  }
  V get_first_cell() const { return detail::make_point<V>(g_.first_cell[0], g_.first_cell[1], g_.first_cell[2]); }
  V get_cell_size() const { return detail::make_point<V>(g_.cell_size[0], g_.cell_size[1], g_.cell_size[2]); }
  std::array<size_t, 3> get_cell_count() const { return {(size_t)g_.cell_count[0], (size_t)g_.cell_count[1], (size_t)g_.cell_count[2]}; }
  size_t get_total_cell_count() const { return (size_t)(g_.cell_count[0] * g_.cell_count[1] * g_.cell_count[2]); }
  V get_last_cell() const {   // grid.rs:82-88, as written there: first + count * size
    float o[3];
    for (int k = 0; k < 3; ++k) { const float prod = (float)g_.cell_count[k] * g_.cell_size[k]; o[k] = g_.first_cell[k] + prod; }
    return detail::make_point<V>(o[0], o[1], o[2]);
  }
  std::pair<V, V> get_bounding_box() const {   // grid.rs:110-119
    float mn[3], mx[3];
    for (int k = 0; k < 3; ++k) {
      const float half = g_.cell_size[k] * 0.5f;
      mn[k] = g_.first_cell[k] - half;
      const float prod = (float)g_.cell_count[k] * g_.cell_size[k];
      mx[k] = mn[k] + prod;
    }
    return {detail::make_point<V>(mn[0], mn[1], mn[2]), detail::make_point<V>(mx[0], mx[1], mx[2])};
  }
  size_t get_cell_idx(std::array<size_t, 3> cell) const {   // grid.rs:122-124
    const uint64_t c[3] = {cell[0], cell[1], cell[2]};
    return (size_t)m2s_grid_cell_idx(&g_, c);
  }
  std::array<size_t, 3> get_cell_integer_coordinates(size_t cell_idx) const {   // grid.rs:127-132
    const size_t ny = g_.cell_count[1], nz = g_.cell_count[2];
    return {cell_idx / (ny * nz), (cell_idx / nz) % ny, cell_idx % nz};
  }
  V get_cell_center(std::array<size_t, 3> cell) const {   // grid.rs:135-141
    const uint64_t c[3] = {cell[0], cell[1], cell[2]};
    float o[3];
    m2s_grid_cell_center(&g_, c, o);
    return detail::make_point<V>(o[0], o[1], o[2]);
  }
  SnapResult snap_point_to_grid(const V& point) const {   // grid.rs:145-170
    const float p[3] = {detail::px(point), detail::py(point), detail::pz(point)};
    SnapResult r{SnapKind::Inside, {0, 0, 0}};
    for (int k = 0; k < 3; ++k) {
      const float half = g_.cell_size[k] * 0.5f;
      const float mn = g_.first_cell[k] - half;
      const float cell = std::floor((p[k] - mn) / g_.cell_size[k]);
      const long long n = (long long)g_.cell_count[k];
      long long ic = std::isnan(cell) ? 0 : (cell >= 9.2e18f ? n : cell <= -9.2e18f ? -1 : (long long)cell);   // `as isize` saturates
      if (ic < 0 || ic >= n) r.kind = SnapKind::Outside;
      r.cell[k] = (size_t)(ic < 0 ? 0 : ic >= n ? n - 1 : ic);
    }
    return r;
  }
  bool operator==(const Grid& o) const { return std::memcmp(&g_, &o.g_, sizeof(g_)) == 0; }   // #[derive(PartialEq)]
  const m2s_grid& raw() const { return g_; }
  static Grid from_raw(const m2s_grid& g) { Grid r; r.g_ = g; return r; }

 private:
  m2s_grid g_{};
};

// ---- generate_grid_sdf (generate/grid.rs:265-378) --------------------------------------------------------------------
template <class V, class I = uint32_t>
std::vector<float> generate_grid_sdf(const V* vertices, size_t n_vertices, const Topology<I>& indices, const Grid<V>& grid,
                                     SignMethod sign_method = SignMethod::Raycast) {
  detail::Packed<V> v(vertices, n_vertices);
  detail::IndexArg<I> ia(indices);
  std::vector<float> out(grid.get_total_cell_count());
  detail::check(m2s_generate_grid_sdf(v.ptr, n_vertices, ia.ptr, indices.count, ia.bytes, indices.kind, &grid.raw(), (int)sign_method,
                                      out.data(), nullptr));
  return out;
}
template <class V, class I = uint32_t>
std::vector<float> generate_grid_sdf(const std::vector<V>& vertices, const Topology<I>& indices, const Grid<V>& grid,
                                     SignMethod sign_method = SignMethod::Raycast) {
  return generate_grid_sdf(vertices.data(), vertices.size(), indices, grid, sign_method);
}

// ---- serde (serde.rs:75-221) ----------------------------------------------------------------------------------------------
namespace serde {
enum class SerdeErrorKind { SerializationFailed, DeserializationFailed, IoError };
struct SerdeError : Error {
  SerdeErrorKind kind;
  SerdeError(int c, const std::string& m)
      : Error(c, m), kind(c == M2S_ERR_IO ? SerdeErrorKind::IoError
                          : m.rfind("Serialization", 0) == 0 ? SerdeErrorKind::SerializationFailed : SerdeErrorKind::DeserializationFailed) {}
};
inline void check(int rc) { if (rc != M2S_OK) throw SerdeError(rc, m2s_last_error()); }

template <class V> struct SerializeGeneric { const V* query_points; size_t n_queries; const float* distances; size_t n_distances; };
template <class V> struct SerializeGrid { const Grid<V>* grid; const float* distances; size_t n_distances; };
template <class V> using SerializeSdf = std::variant<SerializeGeneric<V>, SerializeGrid<V>>;

template <class V> struct DeserializeGeneric { std::vector<V> query_points; std::vector<float> distances; };
template <class V> struct DeserializeGrid { Grid<V> grid; std::vector<float> distances; };
template <class V> using DeserializeSdf = std::variant<DeserializeGeneric<V>, DeserializeGrid<V>>;

template <class V>
std::vector<uint8_t> serialize(const SerializeSdf<V>& sdf) {   // serde.rs:161-166
  std::vector<uint8_t> out;
  size_t written = 0;
  if (const auto* g = std::get_if<SerializeGrid<V>>(&sdf)) {
    out.resize(m2s_sdf_grid_encoded_size(&g->grid->raw(), g->n_distances));
    if (out.empty()) throw SerdeError(M2S_ERR_BAD_ARG, "SerializationFailed");
    check(m2s_sdf_encode_grid(&g->grid->raw(), g->distances, g->n_distances, out.data(), out.size(), &written, nullptr));
  } else {
    const auto& q = std::get<SerializeGeneric<V>>(sdf);
    detail::Packed<V> pts(q.query_points, q.n_queries);
    out.resize(m2s_sdf_generic_encoded_size(q.n_queries, q.n_distances));
    if (out.empty()) throw SerdeError(M2S_ERR_BAD_ARG, "SerializationFailed");
    check(m2s_sdf_encode_generic(pts.ptr, q.n_queries, q.distances, q.n_distances, out.data(), out.size(), &written, nullptr));
  }
  out.resize(written);
  return out;
}

namespace detail_serde {
template <class V>
DeserializeSdf<V> finish(const m2s_sdf_info& info, std::vector<float>&& q, std::vector<float>&& d) {
  if (info.kind == M2S_SDF_GRID) return DeserializeGrid<V>{Grid<V>::from_raw(info.grid), std::move(d)};
  DeserializeGeneric<V> g;
  g.query_points.reserve(info.n_queries);
  for (uint64_t i = 0; i < info.n_queries; ++i) g.query_points.push_back(mesh_to_sdf::detail::make_point<V>(q[3 * i], q[3 * i + 1], q[3 * i + 2]));
  g.distances = std::move(d);
  return g;
}
}  // namespace detail_serde

template <class V>
DeserializeSdf<V> deserialize(const uint8_t* data, size_t n) {   // serde.rs:169-176
  m2s_sdf_info info;
  check(m2s_sdf_probe(data, n, &info, nullptr));
  std::vector<float> q(3 * info.n_queries), d(info.n_distances);
  check(m2s_sdf_decode(data, n, q.data(), d.data(), nullptr));
  return detail_serde::finish<V>(info, std::move(q), std::move(d));
}
template <class V> DeserializeSdf<V> deserialize(const std::vector<uint8_t>& data) { return deserialize<V>(data.data(), data.size()); }

template <class V>
void save_to_file(const SerializeSdf<V>& sdf, const std::string& path) {   // serde.rs:192-198
  if (const auto* g = std::get_if<SerializeGrid<V>>(&sdf)) {
    check(m2s_sdf_save_grid(path.c_str(), &g->grid->raw(), g->distances, g->n_distances, nullptr));
  } else {
    const auto& q = std::get<SerializeGeneric<V>>(sdf);
    mesh_to_sdf::detail::Packed<V> pts(q.query_points, q.n_queries);
    check(m2s_sdf_save_generic(path.c_str(), pts.ptr, q.n_queries, q.distances, q.n_distances, nullptr));
  }
}

template <class V>
DeserializeSdf<V> read_from_file(const std::string& path) {   // serde.rs:216-220
  m2s_sdf_info info;
  check(m2s_sdf_probe_file(path.c_str(), &info));
  std::vector<float> q(3 * info.n_queries), d(info.n_distances);
  check(m2s_sdf_read_file(path.c_str(), q.data(), d.data(), nullptr));
  return detail_serde::finish<V>(info, std::move(q), std::move(d));
}
}  // namespace serde

}  // namespace mesh_to_sdf
