// wire.h — host side of the V1 container (mesh_to_sdf/src/serde.rs:75-221): the MessagePack envelope
// that rmp-serde's compact encoder puts around the two payload arrays, and a tolerant reader for it.
// Pure host code (no HIP): the payload arrays themselves are produced / consumed by serde.hip's kernels.
#pragma once
#include <cstdint>
#include <cstring>

#include "../../include/m2s.h"

namespace m2s {
namespace wire {

constexpr uint64_t kMaxArray = 0xffffffffull;  // MessagePack array32

// ---- writer ------------------------------------------------------------------------------------
struct Writer {
  uint8_t buf[128];
  uint32_t n = 0;
  void u8(uint8_t b) { buf[n++] = b; }
  void be(uint64_t v, int bytes) {
    for (int i = bytes - 1; i >= 0; --i) u8((uint8_t)(v >> (8 * i)));
  }
  void str(const char* s) {  // fixstr (variant names are short)
    const size_t l = strlen(s);
    u8((uint8_t)(0xa0 | l));
    for (size_t i = 0; i < l; ++i) u8((uint8_t)s[i]);
  }
  void f32(float f) {
    uint32_t b;
    memcpy(&b, &f, 4);
    u8(0xca);
    be(b, 4);
  }
  void uint(uint64_t v) {  // shortest form, as rmp's write_uint
    if (v < 128) u8((uint8_t)v);
    else if (v <= 0xff) { u8(0xcc); be(v, 1); }
    else if (v <= 0xffff) { u8(0xcd); be(v, 2); }
    else if (v <= 0xffffffffull) { u8(0xce); be(v, 4); }
    else { u8(0xcf); be(v, 8); }
  }
  void array(uint64_t len) {
    if (len < 16) u8((uint8_t)(0x90 | len));
    else if (len <= 0xffff) { u8(0xdc); be(len, 2); }
    else { u8(0xdd); be(len, 4); }
  }
  void map1() { u8(0x81); }
};

inline uint32_t array_header_bytes(uint64_t len) { return len < 16 ? 1u : len <= 0xffff ? 3u : 5u; }

// Bytes in front of the distance payload of a Grid container.
inline void grid_prefix(const m2s_grid& g, uint64_t n_distances, Writer* w) {
  w->map1(); w->str("V1");
  w->map1(); w->str("Grid");
  w->array(2);           // SerializeGrid { grid, distances }
  w->array(3);           // Grid { first_cell, cell_size, cell_count }
  w->array(3); for (int k = 0; k < 3; ++k) w->f32(g.first_cell[k]);
  w->array(3); for (int k = 0; k < 3; ++k) w->f32(g.cell_size[k]);
  w->array(3); for (int k = 0; k < 3; ++k) w->uint(g.cell_count[k]);
  w->array(n_distances);
}

// Bytes in front of the point payload of a Generic container (the distances header follows the points).
inline void generic_prefix(uint64_t n_queries, Writer* w) {
  w->map1(); w->str("V1");
  w->map1(); w->str("Generic");
  w->array(2);           // SerializeGeneric { query_points, distances }
  w->array(n_queries);
}

// ---- reader ------------------------------------------------------------------------------------
// Accepts every MessagePack form serde's visitors accept for the target type (f32 from f32/f64/ints,
// usize from any non-negative int form, variant by name or by index).
struct Reader {
  const uint8_t* p;
  size_t n;
  size_t pos = 0;
  bool ok = true;
  Reader(const uint8_t* p_, size_t n_) : p(p_), n(n_) {}
  bool need(size_t k) {
    if (!ok || pos + k > n) { ok = false; return false; }
    return true;
  }
  uint64_t be(int bytes) {
    if (!need(bytes)) return 0;
    uint64_t v = 0;
    for (int i = 0; i < bytes; ++i) v = (v << 8) | p[pos++];
    return v;
  }
  uint8_t tag() { return need(1) ? p[pos++] : 0; }
  bool map1() {
    const uint8_t t = tag();
    if (t == 0x81) return ok;
    if (t == 0xde) return be(2) == 1 && ok;
    if (t == 0xdf) return be(4) == 1 && ok;
    return ok = false;
  }
  // returns the index of the variant among `names`, or -1
  int variant(const char* const* names, int count) {
    const uint8_t t = tag();
    size_t l = 0;
    if ((t & 0xe0) == 0xa0) l = t & 0x1f;
    else if (t == 0xd9) l = be(1);
    else if (t == 0xda) l = be(2);
    else if (t < 0x80) return t < count ? t : (ok = false, -1);
    else if (t == 0xcc) { const uint64_t v = be(1); return v < (uint64_t)count ? (int)v : (ok = false, -1); }
    else if (t == 0xce) { const uint64_t v = be(4); return v < (uint64_t)count ? (int)v : (ok = false, -1); }
    else { ok = false; return -1; }
    if (!need(l)) return -1;
    for (int i = 0; i < count; ++i)
      if (strlen(names[i]) == l && memcmp(names[i], p + pos, l) == 0) { pos += l; return i; }
    ok = false;
    return -1;
  }
  uint64_t array() {
    const uint8_t t = tag();
    if ((t & 0xf0) == 0x90) return t & 0x0f;
    if (t == 0xdc) return be(2);
    if (t == 0xdd) return be(4);
    ok = false;
    return 0;
  }
  bool int_value(uint8_t t, int64_t* sv, uint64_t* uv, bool* neg) {
    *neg = false;
    if (t < 0x80) { *uv = t; return true; }
    if (t >= 0xe0) { *sv = (int8_t)t; *neg = true; return true; }
    switch (t) {
      case 0xcc: *uv = be(1); return ok;
      case 0xcd: *uv = be(2); return ok;
      case 0xce: *uv = be(4); return ok;
      case 0xcf: *uv = be(8); return ok;
      case 0xd0: *sv = (int8_t)be(1); break;
      case 0xd1: *sv = (int16_t)be(2); break;
      case 0xd2: *sv = (int32_t)be(4); break;
      case 0xd3: *sv = (int64_t)be(8); break;
      default: return false;
    }
    if (*sv < 0) *neg = true; else *uv = (uint64_t)*sv;
    return ok;
  }
  float f32() {
    const uint8_t t = tag();
    if (t == 0xca) { const uint32_t b = (uint32_t)be(4); float f; memcpy(&f, &b, 4); return f; }
    if (t == 0xcb) { const uint64_t b = be(8); double d; memcpy(&d, &b, 8); return (float)d; }
    int64_t sv = 0; uint64_t uv = 0; bool neg = false;
    if (int_value(t, &sv, &uv, &neg)) return neg ? (float)sv : (float)uv;
    ok = false;
    return 0.0f;
  }
  uint64_t uint() {
    const uint8_t t = tag();
    int64_t sv = 0; uint64_t uv = 0; bool neg = false;
    if (int_value(t, &sv, &uv, &neg) && !neg) return uv;
    ok = false;
    return 0;
  }
  void point(float out[3]) {
    if (array() != 3) { ok = false; return; }
    for (int k = 0; k < 3; ++k) out[k] = f32();
  }
};

// Reads the envelope up to (and including) the header of the first payload array.
// On return r->pos is the offset of that array's first element.
inline bool read_prefix(Reader* r, m2s_sdf_info* info) {
  memset(info, 0, sizeof(*info));
  static const char* const kVersions[] = {"V1"};
  static const char* const kKinds[] = {"Generic", "Grid"};
  if (!r->map1() || r->variant(kVersions, 1) != 0) return false;
  if (!r->map1()) return false;
  const int kind = r->variant(kKinds, 2);
  if (kind < 0) return false;
  info->kind = kind;
  if (r->array() != 2 || !r->ok) return false;
  if (kind == M2S_SDF_GRID) {
    if (r->array() != 3) return false;
    r->point(info->grid.first_cell);
    r->point(info->grid.cell_size);
    if (r->array() != 3) return false;
    for (int k = 0; k < 3; ++k) info->grid.cell_count[k] = r->uint();
    info->n_distances = r->array();
    info->distances_offset = r->pos;
  } else {
    info->n_queries = r->array();
    info->queries_offset = r->pos;
  }
  return r->ok;
}

}  // namespace wire
}  // namespace m2s
