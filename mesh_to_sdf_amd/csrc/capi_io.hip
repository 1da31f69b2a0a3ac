// capi_io.hip — C ABI of the SURVEY §8(f) rows: the V1 container (serde.rs:75-221).  The envelope is
// host work (wire.h); the payload arrays go through serde.hip's kernels whatever side the data is on.
#include <cstdio>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <stdexcept>
#include <vector>

#include "../../include/m2s.h"
#include "capi_internal.h"
#include "common.h"
#include "wire.h"

namespace m2s {
namespace {

struct Layout {           // byte layout of one container
  wire::Writer head;      // envelope up to the first payload array's first element
  wire::Writer mid;       // Generic only: header of the distances array (sits after the points)
  uint64_t points_off = 0, points_bytes = 0;
  uint64_t mid_off = 0;
  uint64_t dist_off = 0, dist_bytes = 0;
  uint64_t total = 0;
};

bool plan_grid(const m2s_grid& g, uint64_t nd, Layout* L) {
  if (nd > wire::kMaxArray) return false;
  wire::grid_prefix(g, nd, &L->head);
  L->dist_off = L->head.n;
  L->dist_bytes = 5 * nd;
  L->total = L->dist_off + L->dist_bytes;
  return true;
}

bool plan_generic(uint64_t nq, uint64_t nd, Layout* L) {
  if (nq > wire::kMaxArray || nd > wire::kMaxArray) return false;
  wire::generic_prefix(nq, &L->head);
  L->points_off = L->head.n;
  L->points_bytes = 16 * nq;
  L->mid_off = L->points_off + L->points_bytes;
  L->mid.array(nd);
  L->dist_off = L->mid_off + L->mid.n;
  L->dist_bytes = 5 * nd;
  L->total = L->dist_off + L->dist_bytes;
  return true;
}

// Encodes into device memory: `out` itself when out_kind == M2S_MEM_DEVICE, else an arena buffer returned in
// *d_encoded (the caller moves it to its host sink).  Inputs live where c.mem_kind says.
int encode_impl(const CallCtx& c, DeviceState& st, const Layout& L, const float* queries, uint64_t nq,
                const float* distances, uint64_t nd, uint8_t* out, int out_kind, bool sync, uint8_t** d_encoded) {
  size_t need = 4096;
  if (c.mem_kind == M2S_MEM_HOST) need += align_up(nd * 4) + align_up(nq * 12);
  if (out_kind == M2S_MEM_HOST) need += align_up(L.total + 64);
  int rc = ensure_capacity(st, need);
  if (rc) return rc;
  Arena ws{st.base, st.cap, 0};
  const float* d_dist = distances;
  const float* d_q = queries;
  if (c.mem_kind == M2S_MEM_HOST) {
    if (nd) {
      float* p = ws.take<float>(nd);
      if (!p) return fail(M2S_ERR_HIP, "internal: workspace");
      rc = staged_h2d(st, c.stream, reinterpret_cast<char*>(p), reinterpret_cast<const char*>(distances), nd * 4);
      if (rc) return rc;
      d_dist = p;
    }
    if (nq) {
      float* p = ws.take<float>(nq * 3);
      if (!p) return fail(M2S_ERR_HIP, "internal: workspace");
      rc = staged_h2d(st, c.stream, reinterpret_cast<char*>(p), reinterpret_cast<const char*>(queries), nq * 12);
      if (rc) return rc;
      d_q = p;
    }
  }
  uint8_t* d_out = out;
  if (out_kind == M2S_MEM_HOST) {
    d_out = ws.take<uint8_t>(L.total + 64);
    if (!d_out) return fail(M2S_ERR_HIP, "internal: workspace");
  }
  rc = launch_write_bytes(c.stream, d_out, L.head.buf, L.head.n);
  if (rc) return rc;
  if (L.points_bytes) {
    rc = launch_encode_points(c.stream, d_q, nq, d_out + L.points_off);
    if (rc) return rc;
  }
  rc = launch_write_bytes(c.stream, d_out + L.mid_off, L.mid.buf, L.mid.n);
  if (rc) return rc;
  rc = launch_encode_f32(c.stream, d_dist, nd, d_out + L.dist_off);
  if (rc) return rc;
  if (d_encoded) *d_encoded = d_out;
  if (out_kind == M2S_MEM_HOST && out) {
    rc = staged_d2h(st, c.stream, reinterpret_cast<char*>(out), reinterpret_cast<const char*>(d_out), L.total);
    if (rc) return rc;
  }
  if (sync || out_kind == M2S_MEM_HOST || c.mem_kind == M2S_MEM_HOST) M2S_HIP_CHECK(hipStreamSynchronize(c.stream));
  return M2S_OK;
}

// Copies [off, off+len) of the container to the host buffer `dst`.
int fetch(const uint8_t* bytes, int bytes_kind, hipStream_t st, uint64_t off, size_t len, uint8_t* dst) {
  if (len == 0) return 0;
  if (bytes_kind == M2S_MEM_HOST) {
    memcpy(dst, bytes + off, len);
    return 0;
  }
  M2S_HIP_CHECK(hipMemcpyAsync(dst, bytes + off, len, hipMemcpyDeviceToHost, st));
  M2S_HIP_CHECK(hipStreamSynchronize(st));
  return 0;
}

const char* kBadContainer = "DeserializationFailed: not a V1 mesh_to_sdf container (serde.rs:169-176)";

// Envelope + array headers.  When the length arithmetic of the fixed-width encoding does not work out,
// `canonical` is 0 and the counts come from a full host walk (host bytes only; device bytes are copied).
int probe_impl(const uint8_t* bytes, size_t n, int bytes_kind, hipStream_t st, m2s_sdf_info* info,
               std::vector<uint8_t>* host_copy) {
  uint8_t head[160];
  const size_t hn = n < sizeof(head) ? n : sizeof(head);
  int rc = fetch(bytes, bytes_kind, st, 0, hn, head);
  if (rc) return rc;
  wire::Reader r(head, hn);
  if (!wire::read_prefix(&r, info)) return fail(M2S_ERR_BAD_ARG, "%s", kBadContainer);
  if (info->kind == M2S_SDF_GRID) {
    info->canonical = info->distances_offset + 5 * info->n_distances == n ? 1 : 0;   // exact fit (trailing bytes: host walk)
    return M2S_OK;   // a non-canonical grid container still has its count in the envelope
  }
  const uint64_t off2 = info->queries_offset + 16 * info->n_queries;
  if (off2 < n) {
    uint8_t h2[8];
    const size_t l2 = n - off2 < sizeof(h2) ? (size_t)(n - off2) : sizeof(h2);
    rc = fetch(bytes, bytes_kind, st, off2, l2, h2);
    if (rc) return rc;
    wire::Reader r2(h2, l2);
    const uint64_t nd = r2.array();
    if (r2.ok && off2 + r2.pos + 5 * nd == n) {
      info->n_distances = nd;
      info->distances_offset = off2 + r2.pos;
      info->canonical = 1;
      return M2S_OK;
    }
  }
  // not the fixed-width layout: walk the points on the host to find the distances header
  const uint8_t* hb = bytes;
  if (bytes_kind != M2S_MEM_HOST) {
    host_copy->resize(n);
    rc = fetch(bytes, bytes_kind, st, 0, n, host_copy->data());
    if (rc) return rc;
    hb = host_copy->data();
  }
  wire::Reader w(hb, n);
  w.pos = info->queries_offset;
  float tmp[3];
  for (uint64_t i = 0; i < info->n_queries && w.ok; ++i) w.point(tmp);
  info->n_distances = w.array();
  if (!w.ok) return fail(M2S_ERR_BAD_ARG, "%s", kBadContainer);
  info->distances_offset = w.pos;
  info->canonical = 0;
  return M2S_OK;
}

// Scalar reader for containers whose numbers are not all `ca`-encoded.
int decode_on_host(const uint8_t* hb, size_t n, const m2s_sdf_info& info, std::vector<float>* q, std::vector<float>* d) {
  wire::Reader r(hb, n);
  // counts come from the untrusted envelope: an element takes at least 1 byte (a point at least 4), so anything
  // larger than the container is malformed — checked BEFORE sizing the output (rmp-serde caps preallocation likewise)
  if (info.n_distances > n || (info.kind == M2S_SDF_GENERIC && info.n_queries > n / 4)) return fail(M2S_ERR_BAD_ARG, "%s", kBadContainer);
  if (info.kind == M2S_SDF_GENERIC) {
    r.pos = info.queries_offset;
    q->resize(info.n_queries * 3);
    for (uint64_t i = 0; i < info.n_queries && r.ok; ++i) r.point(q->data() + 3 * i);
    if (r.array() != info.n_distances) r.ok = false;
  } else {
    r.pos = info.distances_offset;
  }
  d->resize(info.n_distances);
  for (uint64_t i = 0; i < info.n_distances && r.ok; ++i) (*d)[i] = r.f32();
  if (!r.ok) return fail(M2S_ERR_BAD_ARG, "%s", kBadContainer);
  return M2S_OK;
}

int decode_impl(const CallCtx& c, DeviceState& st, const uint8_t* bytes, size_t n, int bytes_kind, float* queries_out,
                float* distances_out) {
  m2s_sdf_info info;
  std::vector<uint8_t> host_copy;
  int rc = probe_impl(bytes, n, bytes_kind, c.stream, &info, &host_copy);
  if (rc) return rc;
  const uint64_t nq = info.kind == M2S_SDF_GENERIC ? info.n_queries : 0, nd = info.n_distances;
  if (nd && !distances_out) return fail(M2S_ERR_BAD_ARG, "distances_out is NULL");
  if (nq && !queries_out) return fail(M2S_ERR_BAD_ARG, "queries_out is NULL for a Generic container");
  if (info.canonical) {
    size_t need = 4096;
    if (bytes_kind == M2S_MEM_HOST) need += align_up(n + 64);
    if (c.mem_kind == M2S_MEM_HOST) need += align_up(nd * 4) + align_up(nq * 12);
    rc = ensure_capacity(st, need);
    if (rc) return rc;
    Arena ws{st.base, st.cap, 0};
    int* d_err = ws.take<int>(16);
    M2S_HIP_CHECK(hipMemsetAsync(d_err, 0, 64, c.stream));
    const uint8_t* d_bytes = bytes;
    if (bytes_kind == M2S_MEM_HOST) {
      uint8_t* p = ws.take<uint8_t>(n + 64);
      if (!p) return fail(M2S_ERR_HIP, "internal: workspace");
      rc = staged_h2d(st, c.stream, reinterpret_cast<char*>(p), reinterpret_cast<const char*>(bytes), n);
      if (rc) return rc;
      d_bytes = p;
    }
    float* d_q = queries_out;
    float* d_d = distances_out;
    if (c.mem_kind == M2S_MEM_HOST) {
      d_d = ws.take<float>(nd ? nd : 1);
      d_q = ws.take<float>(nq ? nq * 3 : 1);
      if (!d_d || !d_q) return fail(M2S_ERR_HIP, "internal: workspace");
    }
    if (nq) {
      rc = launch_decode_points(c.stream, d_bytes + info.queries_offset, nq, d_q, d_err);
      if (rc) return rc;
    }
    rc = launch_decode_f32(c.stream, d_bytes + info.distances_offset, nd, d_d, d_err);
    if (rc) return rc;
    M2S_HIP_CHECK(hipMemcpyAsync(st.h_err, d_err, 4, hipMemcpyDeviceToHost, c.stream));
    if (c.mem_kind == M2S_MEM_HOST) {
      if (nd) { rc = staged_d2h(st, c.stream, reinterpret_cast<char*>(distances_out), reinterpret_cast<const char*>(d_d), nd * 4); if (rc) return rc; }
      if (nq) { rc = staged_d2h(st, c.stream, reinterpret_cast<char*>(queries_out), reinterpret_cast<const char*>(d_q), nq * 12); if (rc) return rc; }
    }
    M2S_HIP_CHECK(hipStreamSynchronize(c.stream));
    if (st.h_err[0] == 0) return M2S_OK;
    // a tag byte was not `ca` / `93`: the lengths matched by coincidence — use the scalar reader
  }
  const uint8_t* hb = bytes;
  if (bytes_kind != M2S_MEM_HOST) {
    if (host_copy.size() != n) {
      host_copy.resize(n);
      rc = fetch(bytes, bytes_kind, c.stream, 0, n, host_copy.data());
      if (rc) return rc;
    }
    hb = host_copy.data();
  }
  if (info.canonical) {   // re-derive the counts by walking (the fixed-width guess was wrong)
    wire::Reader r(hb, n);
    m2s_sdf_info tmp;
    if (!wire::read_prefix(&r, &tmp)) return fail(M2S_ERR_BAD_ARG, "%s", kBadContainer);
    if (tmp.kind == M2S_SDF_GENERIC) {
      float t3[3];
      for (uint64_t i = 0; i < tmp.n_queries && r.ok; ++i) r.point(t3);
      tmp.n_distances = r.array();
      tmp.distances_offset = r.pos;
      if (!r.ok) return fail(M2S_ERR_BAD_ARG, "%s", kBadContainer);
    }
    if (tmp.n_distances != info.n_distances || tmp.n_queries != info.n_queries)
      return fail(M2S_ERR_BAD_ARG, "%s", kBadContainer);   // the caller sized its buffers from other counts
    info = tmp;
  }
  std::vector<float> q, d;
  rc = decode_on_host(hb, n, info, &q, &d);
  if (rc) return rc;
  const hipMemcpyKind kind = c.mem_kind == M2S_MEM_HOST ? hipMemcpyHostToHost : hipMemcpyHostToDevice;
  if (nd) M2S_HIP_CHECK(hipMemcpy(distances_out, d.data(), nd * 4, kind));
  if (nq) M2S_HIP_CHECK(hipMemcpy(queries_out, q.data(), nq * 12, kind));
  return M2S_OK;
}

// Read-only mapping of a container file: the probe touches two pages, the decode streams the mapping through
// the pinned ring (host threads fault the page cache in while earlier chunks cross PCIe) — no copy of the file.
struct MappedFile {
  const uint8_t* p = nullptr;
  size_t n = 0;
  int open_path(const char* path) {
    const int fd = ::open(path, O_RDONLY);
    if (fd < 0) return fail(M2S_ERR_IO, "IoError: cannot open %s", path);
    struct stat sb;
    if (fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode)) { ::close(fd); return fail(M2S_ERR_IO, "IoError: cannot stat %s", path); }
    n = (size_t)sb.st_size;
    if (n) {
      void* m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
      if (m == MAP_FAILED) { ::close(fd); n = 0; return fail(M2S_ERR_IO, "IoError: cannot map %s", path); }
      p = static_cast<const uint8_t*>(m);
    }
    ::close(fd);
    return M2S_OK;
  }
  ~MappedFile() { if (p) munmap(const_cast<uint8_t*>(p), n); }
};

int save_impl(const char* path, const Layout& L, const float* queries, uint64_t nq, const float* distances, uint64_t nd,
              const m2s_opts* opts) {
  CallCtx c;
  DeviceState* st = nullptr;
  int rc = resolve_ctx(opts, &c, &st);
  if (rc) return rc;
  uint8_t* d_enc = nullptr;   // encoded on the device, then streamed to the file through the pinned ring
  rc = encode_impl(c, *st, L, queries, nq, distances, nd, nullptr, M2S_MEM_HOST, false, &d_enc);
  if (rc) return rc;
  FILE* f = fopen(path, "wb");
  if (!f) return fail(M2S_ERR_IO, "IoError: cannot open %s for writing", path);
  rc = staged_d2h_to_file(*st, c.stream, f, reinterpret_cast<const char*>(d_enc), L.total);
  const int cl = fclose(f);
  if (rc == M2S_ERR_IO || (rc == 0 && cl != 0)) return fail(M2S_ERR_IO, "IoError: short write to %s", path);
  return rc;
}

}  // namespace
}  // namespace m2s

using namespace m2s;

extern "C" {

size_t m2s_sdf_grid_encoded_size(const m2s_grid* grid, size_t n_distances) {
  Layout L;
  if (!grid || !plan_grid(*grid, n_distances, &L)) return 0;
  return L.total;
}

size_t m2s_sdf_generic_encoded_size(size_t n_queries, size_t n_distances) {
  Layout L;
  if (!plan_generic(n_queries, n_distances, &L)) return 0;
  return L.total;
}

int m2s_sdf_encode_grid(const m2s_grid* grid, const float* distances, size_t n_distances, uint8_t* bytes,
                        size_t capacity, size_t* written, const m2s_opts* opts) {
  clear_error();
  if (!grid) return fail(M2S_ERR_BAD_ARG, "grid is NULL");
  if (n_distances && !distances) return fail(M2S_ERR_BAD_ARG, "distances is NULL");
  Layout L;
  if (!plan_grid(*grid, n_distances, &L)) return fail(M2S_ERR_BAD_ARG, "SerializationFailed: %zu elements exceed a MessagePack array", n_distances);
  if (!bytes || capacity < L.total) return fail(M2S_ERR_BAD_ARG, "output buffer too small: need %llu bytes", (unsigned long long)L.total);
  CallCtx c;
  DeviceState* st = nullptr;
  int rc = resolve_ctx(opts, &c, &st);
  if (rc) return rc;
  rc = encode_impl(c, *st, L, nullptr, 0, distances, n_distances, bytes, c.mem_kind, c.sync, nullptr);
  if (rc == M2S_OK && written) *written = L.total;
  return rc;
}

int m2s_sdf_encode_generic(const float* queries, size_t n_queries, const float* distances, size_t n_distances,
                           uint8_t* bytes, size_t capacity, size_t* written, const m2s_opts* opts) {
  clear_error();
  if (n_queries && !queries) return fail(M2S_ERR_BAD_ARG, "queries is NULL");
  if (n_distances && !distances) return fail(M2S_ERR_BAD_ARG, "distances is NULL");
  Layout L;
  if (!plan_generic(n_queries, n_distances, &L)) return fail(M2S_ERR_BAD_ARG, "SerializationFailed: element count exceeds a MessagePack array");
  if (!bytes || capacity < L.total) return fail(M2S_ERR_BAD_ARG, "output buffer too small: need %llu bytes", (unsigned long long)L.total);
  CallCtx c;
  DeviceState* st = nullptr;
  int rc = resolve_ctx(opts, &c, &st);
  if (rc) return rc;
  rc = encode_impl(c, *st, L, queries, n_queries, distances, n_distances, bytes, c.mem_kind, c.sync, nullptr);
  if (rc == M2S_OK && written) *written = L.total;
  return rc;
}

static int m2s_sdf_probe_body(const uint8_t* bytes, size_t n_bytes, m2s_sdf_info* info, const m2s_opts* opts) {
  clear_error();
  if (!bytes || !info) return fail(M2S_ERR_BAD_ARG, "bytes / info is NULL");
  std::vector<uint8_t> host_copy;
  if (!opts || opts->mem_kind == M2S_MEM_HOST) return probe_impl(bytes, n_bytes, M2S_MEM_HOST, nullptr, info, &host_copy);
  CallCtx c;
  DeviceState* st = nullptr;
  int rc = resolve_ctx(opts, &c, &st);
  if (rc) return rc;
  return probe_impl(bytes, n_bytes, c.mem_kind, c.stream, info, &host_copy);
}

static int m2s_sdf_decode_body(const uint8_t* bytes, size_t n_bytes, float* queries_out, float* distances_out, const m2s_opts* opts) {
  clear_error();
  if (!bytes) return fail(M2S_ERR_BAD_ARG, "bytes is NULL");
  CallCtx c;
  DeviceState* st = nullptr;
  int rc = resolve_ctx(opts, &c, &st);
  if (rc) return rc;
  return decode_impl(c, *st, bytes, n_bytes, c.mem_kind, queries_out, distances_out);
}

int m2s_sdf_save_grid(const char* path, const m2s_grid* grid, const float* distances, size_t n_distances,
                      const m2s_opts* opts) {
  clear_error();
  if (!path || !grid) return fail(M2S_ERR_BAD_ARG, "path / grid is NULL");
  if (n_distances && !distances) return fail(M2S_ERR_BAD_ARG, "distances is NULL");
  Layout L;
  if (!plan_grid(*grid, n_distances, &L)) return fail(M2S_ERR_BAD_ARG, "SerializationFailed: %zu elements exceed a MessagePack array", n_distances);
  return save_impl(path, L, nullptr, 0, distances, n_distances, opts);
}

int m2s_sdf_save_generic(const char* path, const float* queries, size_t n_queries, const float* distances,
                         size_t n_distances, const m2s_opts* opts) {
  clear_error();
  if (!path) return fail(M2S_ERR_BAD_ARG, "path is NULL");
  if ((n_queries && !queries) || (n_distances && !distances)) return fail(M2S_ERR_BAD_ARG, "queries / distances is NULL");
  Layout L;
  if (!plan_generic(n_queries, n_distances, &L)) return fail(M2S_ERR_BAD_ARG, "SerializationFailed: element count exceeds a MessagePack array");
  return save_impl(path, L, queries, n_queries, distances, n_distances, opts);
}

static int m2s_sdf_probe_file_body(const char* path, m2s_sdf_info* info) {
  clear_error();
  if (!path || !info) return fail(M2S_ERR_BAD_ARG, "path / info is NULL");
  MappedFile file;
  int rc = file.open_path(path);
  if (rc) return rc;
  static const uint8_t kEmpty = 0;
  std::vector<uint8_t> unused;
  return probe_impl(file.n ? file.p : &kEmpty, file.n, M2S_MEM_HOST, nullptr, info, &unused);
}

static int m2s_sdf_read_file_body(const char* path, float* queries_out, float* distances_out, const m2s_opts* opts) {
  clear_error();
  if (!path) return fail(M2S_ERR_BAD_ARG, "path is NULL");
  MappedFile file;
  int rc = file.open_path(path);
  if (rc) return rc;
  static const uint8_t kEmpty = 0;
  CallCtx c;
  DeviceState* st = nullptr;
  rc = resolve_ctx(opts, &c, &st);
  if (rc) return rc;
  return decode_impl(c, *st, file.n ? file.p : &kEmpty, file.n, M2S_MEM_HOST, queries_out, distances_out);
}

int m2s_order_cells_by_distance(const float* distances, size_t n, uint32_t* ordered_indices, float* iso_limits,
                                const m2s_opts* opts) {
  clear_error();
  if (n && (!distances || !ordered_indices)) return fail(M2S_ERR_BAD_ARG, "distances / ordered_indices is NULL");
  if (n >= (1ull << 32)) return fail(M2S_ERR_BAD_ARG, "%zu cells do not fit the u32 indices of the reference (sdf.rs:67)", n);
  CallCtx c;
  DeviceState* st = nullptr;
  int rc = resolve_ctx(opts, &c, &st);
  if (rc) return rc;
  size_t need = order_workspace_bytes(n) + 8192;
  if (c.mem_kind == M2S_MEM_HOST) need += 2 * align_up(n * 4);
  rc = ensure_capacity(*st, need);
  if (rc) return rc;
  Arena ws{st->base, st->cap, 0};
  float* d_limits = iso_limits ? ws.take<float>(16) : nullptr;
  const float* d_dist = distances;
  uint32_t* d_ord = ordered_indices;
  if (c.mem_kind == M2S_MEM_HOST && n) {
    float* p = ws.take<float>(n);
    d_ord = ws.take<uint32_t>(n);
    if (!p || !d_ord) return fail(M2S_ERR_HIP, "internal: workspace");
    M2S_HIP_CHECK(hipMemcpyAsync(p, distances, n * 4, hipMemcpyHostToDevice, c.stream));
    d_dist = p;
  }
  rc = launch_order_cells(ws, c.stream, d_dist, n, d_ord, d_limits);
  if (rc) return rc;
  if (c.mem_kind == M2S_MEM_HOST && n)
    M2S_HIP_CHECK(hipMemcpyAsync(ordered_indices, d_ord, n * 4, hipMemcpyDeviceToHost, c.stream));
  if (iso_limits) M2S_HIP_CHECK(hipMemcpyAsync(st->h_err + 4, d_limits, 8, hipMemcpyDeviceToHost, c.stream));
  if (c.sync || iso_limits || c.mem_kind == M2S_MEM_HOST) M2S_HIP_CHECK(hipStreamSynchronize(c.stream));
  if (iso_limits) memcpy(iso_limits, st->h_err + 4, 8);
  return M2S_OK;
}

int m2s_merge_instances(const m2s_instance* instances, size_t n_instances, float* vertices_out, uint32_t* indices_out,
                        float* bbox, const m2s_opts* opts) {
  clear_error();
  if (n_instances && !instances) return fail(M2S_ERR_BAD_ARG, "instances is NULL");
  if (n_instances >= (1ull << 31)) return fail(M2S_ERR_BAD_ARG, "too many instances");
  std::vector<uint64_t> first(2 * (n_instances + 1));
  uint64_t* vfirst = first.data();
  uint64_t* ifirst = first.data() + n_instances + 1;
  uint64_t nv = 0, ni = 0;
  size_t stage = 0;
  for (size_t k = 0; k < n_instances; ++k) {
    const m2s_instance& I = instances[k];
    if ((I.n_vertices && !I.vertices) || (I.n_indices && !I.indices)) return fail(M2S_ERR_BAD_ARG, "instance %zu: NULL buffer", k);
    if (I.vertex_stride != 0 && (I.vertex_stride < 12 || I.vertex_stride % 4 != 0)) return fail(M2S_ERR_BAD_ARG, "instance %zu: bad vertex_stride", k);
    vfirst[k] = nv;
    ifirst[k] = ni;
    nv += I.n_vertices;
    ni += I.n_indices;
    const size_t stride = I.vertex_stride ? I.vertex_stride : 12;
    stage += align_up(I.n_vertices ? (I.n_vertices - 1) * stride + 12 : 0) + align_up(I.n_indices * 4);
  }
  vfirst[n_instances] = nv;
  ifirst[n_instances] = ni;
  if ((nv && !vertices_out) || (ni && !indices_out)) return fail(M2S_ERR_BAD_ARG, "vertices_out / indices_out is NULL");
  CallCtx c;
  DeviceState* st = nullptr;
  int rc = resolve_ctx(opts, &c, &st);
  if (rc) return rc;
  size_t need = merge_workspace_bytes() + align_up(n_instances * sizeof(InstanceDev)) + align_up(first.size() * 8) + 8192;
  if (c.mem_kind == M2S_MEM_HOST) need += stage + align_up(nv * 12) + align_up(ni * 4);
  rc = ensure_capacity(*st, need);
  if (rc) return rc;
  Arena ws{st->base, st->cap, 0};
  std::vector<InstanceDev> table(n_instances);
  for (size_t k = 0; k < n_instances; ++k) {
    const m2s_instance& I = instances[k];
    InstanceDev& D = table[k];
    D.stride = I.vertex_stride ? I.vertex_stride : 12;
    memcpy(D.m, I.transform, sizeof(D.m));
    D.vertices = I.vertices;
    D.indices = I.indices;
    if (c.mem_kind == M2S_MEM_HOST) {
      const size_t vb = I.n_vertices ? (I.n_vertices - 1) * D.stride + 12 : 0, ib = I.n_indices * 4;
      char* dv = ws.take<char>(vb ? vb : 1);
      uint32_t* di = ws.take<uint32_t>(I.n_indices ? I.n_indices : 1);
      if (!dv || !di) return fail(M2S_ERR_HIP, "internal: workspace");
      if (vb) M2S_HIP_CHECK(hipMemcpyAsync(dv, I.vertices, vb, hipMemcpyHostToDevice, c.stream));
      if (ib) M2S_HIP_CHECK(hipMemcpyAsync(di, I.indices, ib, hipMemcpyHostToDevice, c.stream));
      D.vertices = dv;
      D.indices = di;
    }
  }
  InstanceDev* d_inst = ws.take<InstanceDev>(n_instances ? n_instances : 1);
  uint64_t* d_first = ws.take<uint64_t>(first.size());
  float* d_bbox = bbox ? ws.take<float>(16) : nullptr;
  float* d_v = vertices_out;
  uint32_t* d_i = indices_out;
  if (c.mem_kind == M2S_MEM_HOST) {
    d_v = ws.take<float>(nv ? nv * 3 : 1);
    d_i = ws.take<uint32_t>(ni ? ni : 1);
  }
  if (!d_inst || !d_first || !d_v || !d_i) return fail(M2S_ERR_HIP, "internal: workspace");
  // the tables are pageable host memory: these copies complete before the vectors go out of scope only
  // because the call synchronises below
  if (n_instances) M2S_HIP_CHECK(hipMemcpyAsync(d_inst, table.data(), n_instances * sizeof(InstanceDev), hipMemcpyHostToDevice, c.stream));
  M2S_HIP_CHECK(hipMemcpyAsync(d_first, first.data(), first.size() * 8, hipMemcpyHostToDevice, c.stream));
  rc = launch_merge_instances(ws, c.stream, d_inst, d_first, d_first + n_instances + 1, (uint32_t)n_instances, nv, ni, d_v, d_i, d_bbox);
  if (rc) return rc;
  if (c.mem_kind == M2S_MEM_HOST) {
    if (nv) M2S_HIP_CHECK(hipMemcpyAsync(vertices_out, d_v, nv * 12, hipMemcpyDeviceToHost, c.stream));
    if (ni) M2S_HIP_CHECK(hipMemcpyAsync(indices_out, d_i, ni * 4, hipMemcpyDeviceToHost, c.stream));
  }
  if (bbox) M2S_HIP_CHECK(hipMemcpyAsync(st->h_err + 4, d_bbox, 24, hipMemcpyDeviceToHost, c.stream));
  M2S_HIP_CHECK(hipStreamSynchronize(c.stream));   // always: the instance tables above are call-local
  if (bbox) memcpy(bbox, st->h_err + 4, 24);
  return M2S_OK;
}

// An exception must not cross the C ABI (a container whose header promises more than the process can allocate).
int m2s_sdf_probe(const uint8_t* bytes, size_t n_bytes, m2s_sdf_info* info, const m2s_opts* opts) {
  try {
    return m2s_sdf_probe_body(bytes, n_bytes, info, opts);
  } catch (const std::bad_alloc&) {
    return fail(M2S_ERR_BAD_ARG, "DeserializationFailed: the container announces more elements than can be allocated");
  } catch (const std::exception& e) {
    return fail(M2S_ERR_BAD_ARG, "DeserializationFailed: %s", e.what());
  }
}

// An exception must not cross the C ABI (a container whose header promises more than the process can allocate).
int m2s_sdf_decode(const uint8_t* bytes, size_t n_bytes, float* queries_out, float* distances_out, const m2s_opts* opts) {
  try {
    return m2s_sdf_decode_body(bytes, n_bytes, queries_out, distances_out, opts);
  } catch (const std::bad_alloc&) {
    return fail(M2S_ERR_BAD_ARG, "DeserializationFailed: the container announces more elements than can be allocated");
  } catch (const std::exception& e) {
    return fail(M2S_ERR_BAD_ARG, "DeserializationFailed: %s", e.what());
  }
}

// An exception must not cross the C ABI (a container whose header promises more than the process can allocate).
int m2s_sdf_probe_file(const char* path, m2s_sdf_info* info) {
  try {
    return m2s_sdf_probe_file_body(path, info);
  } catch (const std::bad_alloc&) {
    return fail(M2S_ERR_BAD_ARG, "DeserializationFailed: the container announces more elements than can be allocated");
  } catch (const std::exception& e) {
    return fail(M2S_ERR_BAD_ARG, "DeserializationFailed: %s", e.what());
  }
}

// An exception must not cross the C ABI (a container whose header promises more than the process can allocate).
int m2s_sdf_read_file(const char* path, float* queries_out, float* distances_out, const m2s_opts* opts) {
  try {
    return m2s_sdf_read_file_body(path, queries_out, distances_out, opts);
  } catch (const std::bad_alloc&) {
    return fail(M2S_ERR_BAD_ARG, "DeserializationFailed: the container announces more elements than can be allocated");
  } catch (const std::exception& e) {
    return fail(M2S_ERR_BAD_ARG, "DeserializationFailed: %s", e.what());
  }
}

}  // extern "C"
