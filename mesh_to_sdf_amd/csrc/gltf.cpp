// gltf.cpp — host-side ingestion of a glTF 2.0 / GLB file into the model instances the generator's
// callers merge.  Mirrors what the reference client extracts from a file:
//   mesh_to_sdf_client/src/gltf/mod.rs:56-89    load_scene: models keyed by MESH index + flatten_hierarchy
//   mesh_to_sdf_client/src/gltf/mod.rs:119-174  load: one model per (mesh, primitive), inserted under mesh.index()
//   mesh_to_sdf_client/src/gltf/scene/mod.rs:56-160  ModelNode tree, read_node, simplify_tree
//   mesh_to_sdf_client/src/gltf/scene/model/mod.rs:247-262  positions + indices of a primitive
//   mesh_to_sdf_client/src/pbr/model.rs:29-32   missing indices => 0..vertex_count
// Pure host code: JSON + byte shuffling.  The per-vertex work (transform, merge, bounding box) is
// m2s_merge_instances on the GPU (client.hip).
//
// Third-party arithmetic restated here (crates not in the reference tree; "parity unpinned" for it):
//   gltf 1.4.1 scene::Transform::matrix()  — T * R * S with cgmath-style from_quaternion / column products
//   glam 0.29 Mat4 * Mat4                  — per column ((x_axis*c.x + y_axis*c.y) + z_axis*c.z) + w_axis*c.w
// Nodes that carry a `matrix` are exact; TRS nodes depend on the restated formula.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "../../include/m2s.h"

namespace m2s {
int fail(int code, const char* fmt, ...);   // capi.hip
void clear_error();

namespace gltf {

// ---- minimal JSON DOM -------------------------------------------------------------------------
struct Json {
  enum Kind { Null, Bool, Num, Str, Arr, Obj } kind = Null;
  bool b = false;
  double num = 0.0;
  std::string str;
  std::vector<Json> arr;
  std::vector<std::pair<std::string, Json>> obj;

  const Json* get(const char* key) const {
    if (kind != Obj) return nullptr;
    for (const auto& kv : obj)
      if (kv.first == key) return &kv.second;
    return nullptr;
  }
  bool is_num() const { return kind == Num; }
};

struct JsonParser {
  const char* p;
  const char* e;
  bool ok = true;
  int depth = 0;
  void ws() { while (p < e && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) ++p; }
  bool lit(const char* s) {
    const size_t l = strlen(s);
    if ((size_t)(e - p) >= l && memcmp(p, s, l) == 0) { p += l; return true; }
    return ok = false;
  }
  static void utf8(std::string& o, uint32_t c) {
    if (c < 0x80) o += (char)c;
    else if (c < 0x800) { o += (char)(0xc0 | (c >> 6)); o += (char)(0x80 | (c & 0x3f)); }
    else if (c < 0x10000) { o += (char)(0xe0 | (c >> 12)); o += (char)(0x80 | ((c >> 6) & 0x3f)); o += (char)(0x80 | (c & 0x3f)); }
    else { o += (char)(0xf0 | (c >> 18)); o += (char)(0x80 | ((c >> 12) & 0x3f)); o += (char)(0x80 | ((c >> 6) & 0x3f)); o += (char)(0x80 | (c & 0x3f)); }
  }
  bool hex4(uint32_t* v) {
    if (e - p < 4) return ok = false;
    *v = 0;
    for (int i = 0; i < 4; ++i) {
      const char c = *p++;
      *v <<= 4;
      if (c >= '0' && c <= '9') *v |= c - '0';
      else if (c >= 'a' && c <= 'f') *v |= c - 'a' + 10;
      else if (c >= 'A' && c <= 'F') *v |= c - 'A' + 10;
      else return ok = false;
    }
    return true;
  }
  bool string(std::string* out) {
    if (p >= e || *p != '"') return ok = false;
    ++p;
    while (p < e && *p != '"') {
      if (*p == '\\') {
        if (++p >= e) return ok = false;
        const char c = *p++;
        switch (c) {
          case '"': *out += '"'; break;
          case '\\': *out += '\\'; break;
          case '/': *out += '/'; break;
          case 'b': *out += '\b'; break;
          case 'f': *out += '\f'; break;
          case 'n': *out += '\n'; break;
          case 'r': *out += '\r'; break;
          case 't': *out += '\t'; break;
          case 'u': {
            uint32_t c1;
            if (!hex4(&c1)) return false;
            if (c1 >= 0xd800 && c1 < 0xdc00 && e - p >= 6 && p[0] == '\\' && p[1] == 'u') {
              p += 2;
              uint32_t c2;
              if (!hex4(&c2)) return false;
              c1 = 0x10000 + ((c1 - 0xd800) << 10) + (c2 - 0xdc00);
            }
            utf8(*out, c1);
            break;
          }
          default: return ok = false;
        }
      } else {
        *out += *p++;
      }
    }
    if (p >= e) return ok = false;
    ++p;
    return true;
  }
  bool value(Json* v) {
    if (++depth > 256) return ok = false;
    ws();
    if (p >= e) return ok = false;
    bool r = true;
    if (*p == '{') {
      v->kind = Json::Obj;
      ++p;
      ws();
      if (p < e && *p == '}') { ++p; }
      else {
        for (;;) {
          ws();
          std::string k;
          if (!string(&k)) { r = false; break; }
          ws();
          if (p >= e || *p != ':') { r = ok = false; break; }
          ++p;
          v->obj.emplace_back(std::move(k), Json());
          if (!value(&v->obj.back().second)) { r = false; break; }
          ws();
          if (p < e && *p == ',') { ++p; continue; }
          if (p < e && *p == '}') { ++p; break; }
          r = ok = false;
          break;
        }
      }
    } else if (*p == '[') {
      v->kind = Json::Arr;
      ++p;
      ws();
      if (p < e && *p == ']') { ++p; }
      else {
        for (;;) {
          v->arr.emplace_back();
          if (!value(&v->arr.back())) { r = false; break; }
          ws();
          if (p < e && *p == ',') { ++p; continue; }
          if (p < e && *p == ']') { ++p; break; }
          r = ok = false;
          break;
        }
      }
    } else if (*p == '"') {
      v->kind = Json::Str;
      r = string(&v->str);
    } else if (*p == 't') { v->kind = Json::Bool; v->b = true; r = lit("true"); }
    else if (*p == 'f') { v->kind = Json::Bool; v->b = false; r = lit("false"); }
    else if (*p == 'n') { v->kind = Json::Null; r = lit("null"); }
    else {
      // number: serde_json parses to f64; an f32 field is that f64 cast to f32
      const char* s = p;
      if (p < e && *p == '-') ++p;
      while (p < e && ((*p >= '0' && *p <= '9') || *p == '.' || *p == 'e' || *p == 'E' || *p == '+' || *p == '-')) ++p;
      if (p == s) return ok = false;
      const std::string t(s, p);
      char* endp = nullptr;
      v->num = strtod(t.c_str(), &endp);
      v->kind = Json::Num;
      if (endp != t.c_str() + t.size()) r = ok = false;
    }
    --depth;
    return r && ok;
  }
};

// ---- glam / gltf-crate matrix arithmetic (f32, separate mul and add, column-major) --------------
struct Mat4 { float c[4][4]; };   // c[col][row]

Mat4 identity() {
  Mat4 m{};
  for (int i = 0; i < 4; ++i) m.c[i][i] = 1.0f;
  return m;
}

// glam Mat4::mul_mat4: column j = ((a.x_axis*b[j].x + a.y_axis*b[j].y) + a.z_axis*b[j].z) + a.w_axis*b[j].w
Mat4 mul_glam(const Mat4& a, const Mat4& b) {
  Mat4 r;
  for (int j = 0; j < 4; ++j)
    for (int i = 0; i < 4; ++i) {
      float acc = a.c[0][i] * b.c[j][0];
      float t = a.c[1][i] * b.c[j][1];
      acc = acc + t;
      t = a.c[2][i] * b.c[j][2];
      acc = acc + t;
      t = a.c[3][i] * b.c[j][3];
      acc = acc + t;
      r.c[j][i] = acc;
    }
  return r;
}

// gltf crate math.rs (cgmath): column j = a.x*b[j][0] + a.y*b[j][1] + a.z*b[j][2] + a.w*b[j][3], left to right —
// the same association as glam's, so one routine serves both.
Mat4 trs_matrix(const float t[3], const float q[4] /*x,y,z,w*/, const float s[3]) {
  const float qx = q[0], qy = q[1], qz = q[2], qs = q[3];
  float x2 = qx + qx, y2 = qy + qy, z2 = qz + qz;
  float xx2 = x2 * qx, xy2 = x2 * qy, xz2 = x2 * qz;
  float yy2 = y2 * qy, yz2 = y2 * qz, zz2 = z2 * qz;
  float sy2 = y2 * qs, sz2 = z2 * qs, sx2 = x2 * qs;
  Mat4 R = identity();
  float a;
  a = 1.0f - yy2; R.c[0][0] = a - zz2; R.c[0][1] = xy2 + sz2; R.c[0][2] = xz2 - sy2;
  R.c[1][0] = xy2 - sz2; a = 1.0f - xx2; R.c[1][1] = a - zz2; R.c[1][2] = yz2 + sx2;
  R.c[2][0] = xz2 + sy2; R.c[2][1] = yz2 - sx2; a = 1.0f - xx2; R.c[2][2] = a - yy2;
  Mat4 T = identity();
  T.c[3][0] = t[0]; T.c[3][1] = t[1]; T.c[3][2] = t[2];
  Mat4 S = identity();
  S.c[0][0] = s[0]; S.c[1][1] = s[1]; S.c[2][2] = s[2];
  return mul_glam(mul_glam(T, R), S);
}

// ---- document ------------------------------------------------------------------------------------
struct Model {
  std::vector<float> positions;   // packed xyz
  std::vector<uint32_t> indices;
};

struct Node {                      // scene/mod.rs:32-43 ModelNode
  int model_id = -1;
  Mat4 transform = identity();
  std::vector<Node> children;
};

struct Instance { int model_id; Mat4 transform; };

struct Doc {
  Json root;
  std::vector<std::vector<uint8_t>> buffers;
  std::map<int, Model> models;     // keyed by mesh index
  std::vector<Instance> instances;
  uint64_t n_scenes = 0;
  std::string error;
  bool err(const std::string& m) { if (error.empty()) error = m; return false; }
};

bool read_file(const std::string& path, std::vector<uint8_t>* out) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return false;
  fseek(f, 0, SEEK_END);
  const long sz = ftell(f);
  fseek(f, 0, SEEK_SET);
  if (sz < 0) { fclose(f); return false; }
  out->resize((size_t)sz);
  const size_t r = sz ? fread(out->data(), 1, (size_t)sz, f) : 0;
  fclose(f);
  return r == (size_t)sz;
}

bool base64(const std::string& s, std::vector<uint8_t>* out) {
  uint32_t acc = 0;
  int bits = 0;
  for (char ch : s) {
    int v;
    if (ch >= 'A' && ch <= 'Z') v = ch - 'A';
    else if (ch >= 'a' && ch <= 'z') v = ch - 'a' + 26;
    else if (ch >= '0' && ch <= '9') v = ch - '0' + 52;
    else if (ch == '+' || ch == '-') v = 62;
    else if (ch == '/' || ch == '_') v = 63;
    else if (ch == '=') break;
    else return false;
    acc = (acc << 6) | (uint32_t)v;
    bits += 6;
    if (bits >= 8) { bits -= 8; out->push_back((uint8_t)(acc >> bits)); }
  }
  return true;
}

std::string percent_decode(const std::string& s) {
  std::string o;
  for (size_t i = 0; i < s.size(); ++i) {
    if (s[i] == '%' && i + 2 < s.size() && isxdigit((unsigned char)s[i + 1]) && isxdigit((unsigned char)s[i + 2])) {
      o += (char)strtol(s.substr(i + 1, 2).c_str(), nullptr, 16);
      i += 2;
    } else {
      o += s[i];
    }
  }
  return o;
}

int64_t get_int(const Json& o, const char* key, int64_t dflt) {
  const Json* v = o.get(key);
  return v && v->is_num() ? (int64_t)v->num : dflt;
}

const Json* get_arr(const Json& o, const char* key) {
  const Json* v = o.get(key);
  return v && v->kind == Json::Arr ? v : nullptr;
}

// Reads accessor `idx` as rows of `ncomp` components converted by `conv`; honours byteStride and sparse.
template <class T, class F>
bool read_accessor(Doc& d, int64_t idx, const char* want_type, int ncomp, bool want_float, std::vector<T>* out, F conv) {
  const Json* accs = get_arr(d.root, "accessors");
  if (!accs || idx < 0 || (size_t)idx >= accs->arr.size()) return d.err("accessor index out of range");
  const Json& acc = accs->arr[(size_t)idx];
  const Json* ty = acc.get("type");
  if (!ty || ty->str != want_type) return d.err(std::string("accessor type is not ") + want_type);
  const int64_t ct = get_int(acc, "componentType", 0), count = get_int(acc, "count", -1);
  size_t csz;
  switch (ct) {
    case 5120: case 5121: csz = 1; break;
    case 5122: case 5123: csz = 2; break;
    case 5125: case 5126: csz = 4; break;
    default: return d.err("bad accessor componentType");
  }
  if (want_float != (ct == 5126)) return d.err("unexpected accessor componentType");
  if (count < 0) return d.err("accessor without count");
  out->assign((size_t)count * ncomp, T(0));   // no bufferView: zeros (then sparse substitution)
  auto view_span = [&](int64_t bv, int64_t extra_off, size_t elem, size_t n, const uint8_t** base, size_t* stride) -> bool {
    const Json* views = get_arr(d.root, "bufferViews");
    if (!views || bv < 0 || (size_t)bv >= views->arr.size()) return d.err("bufferView index out of range");
    const Json& v = views->arr[(size_t)bv];
    const int64_t buf = get_int(v, "buffer", -1), off = get_int(v, "byteOffset", 0), len = get_int(v, "byteLength", -1);
    if (buf < 0 || (size_t)buf >= d.buffers.size()) return d.err("buffer index out of range");
    *stride = (size_t)get_int(v, "byteStride", 0);
    if (*stride == 0) *stride = elem;
    const size_t start = (size_t)off + (size_t)extra_off;
    const size_t need = n ? (n - 1) * *stride + elem : 0;
    if (len < 0 || (size_t)extra_off + need > (size_t)len || start + need > d.buffers[(size_t)buf].size())
      return d.err("accessor exceeds its bufferView");
    *base = d.buffers[(size_t)buf].data() + start;
    return true;
  };
  const Json* bvj = acc.get("bufferView");
  if (bvj && bvj->is_num()) {
    const uint8_t* base;
    size_t stride;
    if (!view_span((int64_t)bvj->num, get_int(acc, "byteOffset", 0), csz * ncomp, (size_t)count, &base, &stride)) return false;
    for (int64_t i = 0; i < count; ++i)
      for (int k = 0; k < ncomp; ++k) (*out)[(size_t)i * ncomp + k] = conv(base + (size_t)i * stride + (size_t)k * csz, ct);
  }
  if (const Json* sp = acc.get("sparse")) {
    const int64_t sc = get_int(*sp, "count", 0);
    const Json* si = sp->get("indices");
    const Json* sv = sp->get("values");
    if (!si || !sv) return d.err("bad sparse accessor");
    const int64_t ict = get_int(*si, "componentType", 0);
    const size_t isz = ict == 5121 ? 1 : ict == 5123 ? 2 : ict == 5125 ? 4 : 0;
    if (!isz) return d.err("bad sparse index type");
    const uint8_t *ib, *vb;
    size_t is, vs;
    if (!view_span(get_int(*si, "bufferView", -1), get_int(*si, "byteOffset", 0), isz, (size_t)sc, &ib, &is)) return false;
    if (!view_span(get_int(*sv, "bufferView", -1), get_int(*sv, "byteOffset", 0), csz * ncomp, (size_t)sc, &vb, &vs)) return false;
    for (int64_t j = 0; j < sc; ++j) {
      uint32_t at = 0;
      memcpy(&at, ib + (size_t)j * is, isz);
      if ((int64_t)at >= count) return d.err("sparse index out of range");
      for (int k = 0; k < ncomp; ++k) (*out)[(size_t)at * ncomp + k] = conv(vb + (size_t)j * vs + (size_t)k * csz, ct);
    }
  }
  return true;
}

bool load_models(Doc& d) {
  const Json* meshes = get_arr(d.root, "meshes");
  if (!meshes) return true;
  for (size_t mi = 0; mi < meshes->arr.size(); ++mi) {
    const Json* prims = get_arr(meshes->arr[mi], "primitives");
    if (!prims) return d.err("mesh without primitives");
    // mod.rs:137-149 inserts every (mesh, primitive) under mesh.index(): one survives.  The reference iterates
    // in parallel, so which one is unspecified there; document order (the last primitive) is used here.
    for (const Json& prim : prims->arr) {
      Model m;
      const Json* attrs = prim.get("attributes");
      const Json* pos = attrs ? attrs->get("POSITION") : nullptr;
      if (!pos || !pos->is_num()) return d.err("The model primitive doesn't contain positions");   // model/mod.rs:257
      auto f32 = [](const uint8_t* p, int64_t) { float f; memcpy(&f, p, 4); return f; };
      if (!read_accessor<float>(d, (int64_t)pos->num, "VEC3", 3, true, &m.positions, f32)) return false;
      const Json* ind = prim.get("indices");
      if (ind && ind->is_num()) {
        auto u32 = [](const uint8_t* p, int64_t ct) -> uint32_t {
          if (ct == 5121) return *p;
          if (ct == 5123) { uint16_t v; memcpy(&v, p, 2); return v; }
          uint32_t v; memcpy(&v, p, 4); return v;
        };
        const Json* accs = get_arr(d.root, "accessors");
        const int64_t ai = (int64_t)ind->num;
        if (!accs || ai < 0 || (size_t)ai >= accs->arr.size()) return d.err("accessor index out of range");
        const int64_t ct = get_int(accs->arr[(size_t)ai], "componentType", 0);
        if (ct != 5121 && ct != 5123 && ct != 5125) return d.err("index accessor is not u8/u16/u32");
        if (!read_accessor<uint32_t>(d, ai, "SCALAR", 1, false, &m.indices, u32)) return false;
      } else {
        m.indices.resize(m.positions.size() / 3);                  // pbr/model.rs:29-32
        for (size_t i = 0; i < m.indices.size(); ++i) m.indices[i] = (uint32_t)i;
      }
      d.models[(int)mi] = std::move(m);
    }
  }
  return true;
}

bool node_transform(Doc& d, const Json& n, Mat4* out) {
  if (const Json* m = get_arr(n, "matrix")) {
    if (m->arr.size() != 16) return d.err("node matrix must have 16 elements");
    for (int j = 0; j < 4; ++j)
      for (int i = 0; i < 4; ++i) out->c[j][i] = (float)m->arr[(size_t)(4 * j + i)].num;
    return true;
  }
  float t[3] = {0, 0, 0}, q[4] = {0, 0, 0, 1}, s[3] = {1, 1, 1};
  auto fill = [&](const char* key, float* dst, size_t cnt) -> bool {
    if (const Json* a = get_arr(n, key)) {
      if (a->arr.size() != cnt) return d.err(std::string("node ") + key + " has the wrong length");
      for (size_t i = 0; i < cnt; ++i) dst[i] = (float)a->arr[i].num;
    }
    return true;
  };
  if (!fill("translation", t, 3) || !fill("rotation", q, 4) || !fill("scale", s, 3)) return false;
  *out = trs_matrix(t, q, s);
  return true;
}

// scene/mod.rs:113-159 read_node
bool read_node(Doc& d, Node* parent, int64_t idx, int depth) {
  const Json* nodes = get_arr(d.root, "nodes");
  if (!nodes || idx < 0 || (size_t)idx >= nodes->arr.size()) return d.err("node index out of range");
  if (depth > 512) return d.err("node hierarchy too deep (cycle?)");
  const Json& n = nodes->arr[(size_t)idx];
  Node nn;
  if (!node_transform(d, n, &nn.transform)) return false;
  if (const Json* ch = get_arr(n, "children"))
    for (const Json& c : ch->arr)
      if (!read_node(d, &nn, (int64_t)c.num, depth + 1)) return false;
  const Json* mesh = n.get("mesh");
  if (mesh && mesh->is_num()) {
    nn.model_id = (int)mesh->num;
    if (!d.models.count(nn.model_id)) return d.err("node references a missing mesh");
  }
  parent->children.push_back(std::move(nn));
  return true;
}

// scene/mod.rs:56-75 simplify_tree
void simplify(Node* n) {
  for (Node& c : n->children) simplify(&c);
  if (n->model_id < 0 && n->children.size() == 1) {
    Node child = std::move(n->children[0]);
    child.transform = mul_glam(n->transform, child.transform);
    *n = std::move(child);
  }
}

// mod.rs:91-106 flatten_hierarchy
void flatten(const Node& n, const Mat4& parent, std::vector<Instance>* out) {
  const Mat4 t = mul_glam(parent, n.transform);
  for (const Node& c : n.children) flatten(c, t, out);
  if (n.model_id >= 0) out->push_back({n.model_id, t});
}

bool load_scenes(Doc& d) {
  const Json* scenes = get_arr(d.root, "scenes");
  if (!scenes) return true;
  d.n_scenes = scenes->arr.size();
  for (const Json& sc : scenes->arr) {
    Node root;                                             // Scene::load, scene/mod.rs:87-107
    if (const Json* ns = get_arr(sc, "nodes"))
      for (const Json& ni : ns->arr) {
        Node new_root;
        if (!read_node(d, &new_root, (int64_t)ni.num, 0)) return false;
        root.children.push_back(std::move(new_root));
      }
    simplify(&root);
    flatten(root, identity(), &d.instances);               // load_scene, mod.rs:77-86
  }
  return true;
}

bool load_buffers(Doc& d, const std::string& path, std::vector<uint8_t>* glb_bin, bool have_bin) {
  const Json* bufs = get_arr(d.root, "buffers");
  if (!bufs) return true;
  std::string dir = path;
  const size_t slash = dir.find_last_of('/');
  dir = slash == std::string::npos ? std::string(".") : dir.substr(0, slash);
  for (size_t i = 0; i < bufs->arr.size(); ++i) {
    const Json& b = bufs->arr[i];
    const int64_t len = get_int(b, "byteLength", -1);
    std::vector<uint8_t> data;
    const Json* uri = b.get("uri");
    if (uri && uri->kind == Json::Str) {
      if (uri->str.rfind("data:", 0) == 0) {
        const size_t comma = uri->str.find(',');
        if (comma == std::string::npos || !base64(uri->str.substr(comma + 1), &data)) return d.err("bad data: uri");
      } else if (!read_file(dir + "/" + percent_decode(uri->str), &data)) {
        return d.err("cannot read buffer " + uri->str);
      }
    } else if (i == 0 && have_bin) {
      data = std::move(*glb_bin);
    } else {
      return d.err("buffer without uri and without a GLB BIN chunk");
    }
    if (len < 0 || data.size() < (size_t)len) return d.err("buffer shorter than its byteLength");
    d.buffers.push_back(std::move(data));
  }
  return true;
}

// gltf 1.4.1 built with features ["KHR_lights_punctual", "names", "extras"] (mesh_to_sdf_client/Cargo.toml:48-52)
// rejects any other REQUIRED extension at import (mod.rs:407-410: tests/dragon.glb must fail to load).
bool check_extensions(Doc& d) {
  if (const Json* req = get_arr(d.root, "extensionsRequired"))
    for (const Json& x : req->arr)
      if (x.kind != Json::Str || x.str != "KHR_lights_punctual") return d.err("unsupported required extension " + x.str);
  return true;
}

bool parse(Doc& d, const std::string& path, bool* io_error) {
  std::vector<uint8_t> file;
  if (!read_file(path, &file)) { *io_error = true; return d.err("cannot read " + path); }
  std::vector<uint8_t> bin;
  bool have_bin = false;
  const char *js = nullptr, *je = nullptr;
  if (file.size() >= 12 && memcmp(file.data(), "glTF", 4) == 0) {
    uint32_t ver, total;
    memcpy(&ver, file.data() + 4, 4);
    memcpy(&total, file.data() + 8, 4);
    if (ver != 2) return d.err("unsupported GLB version");
    if (total > file.size()) return d.err("GLB length exceeds the file");
    size_t off = 12;
    while (off + 8 <= total) {
      uint32_t clen, ctype;
      memcpy(&clen, file.data() + off, 4);
      memcpy(&ctype, file.data() + off + 4, 4);
      if (off + 8 + (size_t)clen > total) return d.err("GLB chunk exceeds the file");
      if (ctype == 0x4E4F534Au && !js) { js = (const char*)file.data() + off + 8; je = js + clen; }
      else if (ctype == 0x004E4942u && !have_bin) { bin.assign(file.data() + off + 8, file.data() + off + 8 + clen); have_bin = true; }
      off += 8 + (size_t)clen;
    }
    if (!js) return d.err("GLB without a JSON chunk");
  } else {
    js = (const char*)file.data();
    je = js + file.size();
  }
  JsonParser jp{js, je};
  if (!jp.value(&d.root) || d.root.kind != Json::Obj) return d.err("invalid glTF JSON");
  jp.ws();
  return check_extensions(d) && load_buffers(d, path, &bin, have_bin) && load_models(d) && load_scenes(d);
}

}  // namespace gltf
}  // namespace m2s

struct m2s_gltf {
  m2s::gltf::Doc doc;
};

extern "C" {

int m2s_gltf_open(const char* path, m2s_gltf** out, m2s_gltf_info* info) {
  m2s::clear_error();
  if (!path || !out) return m2s::fail(M2S_ERR_BAD_ARG, "path / out is NULL");
  *out = nullptr;
  std::unique_ptr<m2s_gltf> g(new m2s_gltf());
  bool io = false;
  if (!m2s::gltf::parse(g->doc, path, &io))
    return m2s::fail(io ? M2S_ERR_IO : M2S_ERR_BAD_ARG, "gltf: %s", g->doc.error.c_str());
  g->doc.root = m2s::gltf::Json();   // the DOM is no longer needed
  if (info) {
    memset(info, 0, sizeof(*info));
    info->n_scenes = g->doc.n_scenes;
    info->n_models = g->doc.models.size();
    info->n_instances = g->doc.instances.size();
    for (const auto& I : g->doc.instances) {
      const auto& m = g->doc.models.at(I.model_id);
      info->n_vertices += m.positions.size() / 3;
      info->n_indices += m.indices.size();
    }
  }
  *out = g.release();
  return M2S_OK;
}

int m2s_gltf_instances(const m2s_gltf* g, m2s_instance* out, size_t capacity) {
  m2s::clear_error();
  if (!g || (!out && !g->doc.instances.empty())) return m2s::fail(M2S_ERR_BAD_ARG, "gltf / out is NULL");
  if (capacity < g->doc.instances.size()) return m2s::fail(M2S_ERR_BAD_ARG, "capacity %zu < %zu instances", capacity, g->doc.instances.size());
  for (size_t k = 0; k < g->doc.instances.size(); ++k) {
    const auto& I = g->doc.instances[k];
    const auto& m = g->doc.models.at(I.model_id);
    m2s_instance& o = out[k];
    o.vertices = m.positions.data();
    o.n_vertices = m.positions.size() / 3;
    o.vertex_stride = 12;
    o.indices = m.indices.data();
    o.n_indices = m.indices.size();
    memcpy(o.transform, I.transform.c, sizeof(o.transform));
  }
  return M2S_OK;
}

void m2s_gltf_close(m2s_gltf* g) { delete g; }

}  // extern "C"
