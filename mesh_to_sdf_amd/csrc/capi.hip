// capi.hip — the C ABI of include/m2s.h: argument checks that mirror the reference's panics,
// per-device workspace, host<->device staging for the drop-in (host pointer) case, phase timing.
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>

#include "../../include/m2s.h"
#include "common.h"

namespace m2s {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

namespace {

struct DeviceState {
  char* base = nullptr;
  size_t cap = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  int* h_err = nullptr;  // pinned
};

std::mutex g_mu;
std::map<int, DeviceState> g_dev;

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

int get_state(int device, DeviceState** out) {
  DeviceState& s = g_dev[device];
  if (!s.stream) {
    M2S_HIP_CHECK(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
    for (auto& e : s.ev) M2S_HIP_CHECK(hipEventCreate(&e));
    M2S_HIP_CHECK(hipHostMalloc((void**)&s.h_err, 64, hipHostMallocDefault));
  }
  *out = &s;
  return 0;
}

int ensure_capacity(DeviceState& s, size_t bytes) {
  if (bytes <= s.cap) return 0;
  if (s.base) {
    M2S_HIP_CHECK(hipDeviceSynchronize());
    M2S_HIP_CHECK(hipFree(s.base));
    s.base = nullptr;
    s.cap = 0;
  }
  const size_t want = bytes + bytes / 8 + (1u << 20);
  M2S_HIP_CHECK(hipMalloc((void**)&s.base, want));
  s.cap = want;
  return 0;
}

struct CallCtx {
  int device = -1;
  int mem_kind = M2S_MEM_HOST;
  int algorithm = 0;
  bool sync = true;
  hipStream_t stream = nullptr;
  m2s_timings* timings = nullptr;
  uint64_t x_begin = 0, x_end = 0;
};

int resolve_ctx(const m2s_opts* opts, CallCtx* c, DeviceState** st) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(M2S_ERR_HIP, "no HIP device available: the MI355X kernels cannot run (there is no CPU fallback)");
  int dev = -1;
  if (opts) {
    if (opts->struct_size != 0 && opts->struct_size < sizeof(m2s_opts)) return fail(M2S_ERR_BAD_ARG, "m2s_opts.struct_size too small");
    dev = opts->device;
    c->mem_kind = opts->mem_kind;
    c->algorithm = opts->algorithm;
    c->timings = opts->timings;
    c->x_begin = opts->x_begin;
    c->x_end = opts->x_end;
    c->sync = opts->synchronous != 0 || opts->mem_kind == M2S_MEM_HOST || opts->timings != nullptr;
  }
  if (c->mem_kind != M2S_MEM_HOST && c->mem_kind != M2S_MEM_DEVICE) return fail(M2S_ERR_BAD_ARG, "bad mem_kind");
  if (c->algorithm != 0 && c->algorithm != 1) return fail(M2S_ERR_BAD_ARG, "bad algorithm");
  if (dev < 0) M2S_HIP_CHECK(hipGetDevice(&dev));
  if (dev >= ndev) return fail(M2S_ERR_BAD_ARG, "device %d out of range (%d devices)", dev, ndev);
  M2S_HIP_CHECK(hipSetDevice(dev));
  c->device = dev;
  int rc = get_state(dev, st);
  if (rc) return rc;
  c->stream = (opts && opts->stream) ? (hipStream_t)opts->stream : (*st)->stream;
  return 0;
}

int check_mesh_args(const float* vertices, size_t n_vertices, const void* indices, size_t n_indices, int index_bytes,
                    int topology) {
  if (topology != M2S_TRIANGLE_LIST && topology != M2S_TRIANGLE_STRIP) return fail(M2S_ERR_BAD_ARG, "bad topology %d", topology);
  if (n_vertices && !vertices) return fail(M2S_ERR_BAD_ARG, "vertices is NULL");
  if (indices && index_bytes != 2 && index_bytes != 4) return fail(M2S_ERR_BAD_ARG, "index_bytes must be 2 or 4");
  if (!indices && n_indices) return fail(M2S_ERR_BAD_ARG, "n_indices > 0 with NULL indices");
  if (n_vertices >= 0x7fffffffull || n_indices >= 0x17fffffffull) return fail(M2S_ERR_BAD_ARG, "mesh too large for 32-bit indexing");
  return 0;
}

size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

struct StagedMesh {
  const float* d_verts = nullptr;
  const void* d_indices = nullptr;
};

// Copies the mesh to the device when the caller passed host pointers.
int stage_mesh(Arena& ws, const CallCtx& c, const float* vertices, size_t n_vertices, const void* indices,
               size_t n_indices, int index_bytes, StagedMesh* out) {
  if (c.mem_kind == M2S_MEM_DEVICE) {
    out->d_verts = vertices;
    out->d_indices = indices;
    return 0;
  }
  if (n_vertices) {
    float* dv = ws.take<float>(n_vertices * 3);
    if (!dv) return fail(M2S_ERR_HIP, "internal: workspace");
    M2S_HIP_CHECK(hipMemcpyAsync(dv, vertices, n_vertices * 12, hipMemcpyHostToDevice, c.stream));
    out->d_verts = dv;
  }
  if (indices && n_indices) {
    char* di = ws.take<char>(n_indices * (size_t)index_bytes);
    if (!di) return fail(M2S_ERR_HIP, "internal: workspace");
    M2S_HIP_CHECK(hipMemcpyAsync(di, indices, n_indices * (size_t)index_bytes, hipMemcpyHostToDevice, c.stream));
    out->d_indices = di;
  } else if (indices) {
    out->d_indices = ws.take<char>(16);  // non-NULL marker: "indices given, but empty"
  }
  return 0;
}

int finish_call(const CallCtx& c, DeviceState& st, int* d_err, m2s_timings* t, size_t n_tris, size_t n_units,
                bool had_sign, bool had_seed_event = false) {
  if (!c.sync) return M2S_OK;
  M2S_HIP_CHECK(hipMemcpyAsync(st.h_err, d_err, sizeof(int), hipMemcpyDeviceToHost, c.stream));
  M2S_HIP_CHECK(hipStreamSynchronize(c.stream));
  if (t) {
    float a = 0, b = 0, d = 0, tot = 0;
    (void)hipEventElapsedTime(&a, st.ev[0], st.ev[1]);
    (void)hipEventElapsedTime(&b, st.ev[1], st.ev[2]);
    float sd = 0;
    if (had_seed_event) {
      (void)hipEventElapsedTime(&sd, st.ev[2], st.ev[4]);
      (void)hipEventElapsedTime(&d, st.ev[4], st.ev[3]);
    } else {
      (void)hipEventElapsedTime(&d, st.ev[2], st.ev[3]);
    }
    (void)hipEventElapsedTime(&tot, st.ev[0], st.ev[3]);
    t->accel_build_ms = a;
    t->sign_ms = had_sign ? b : 0.0f;
    t->distance_ms = d;
    t->seed_ms = sd;
    t->total_ms = tot;
    t->n_triangles = n_tris;
    t->n_units = n_units;
    t->distance_launches = 1;
    t->reserved = 0;
  }
  const int e = *st.h_err;
  if (e & ERRF_INDEX_OOB) return fail(M2S_ERR_BAD_ARG, "vertex index out of range (the reference panics indexing `vertices`)");
  if (e & ERRF_NAN) return fail(M2S_ERR_NAN, "NaN distance (lib.rs:257)");
  return M2S_OK;
}

}  // namespace
}  // namespace m2s

using namespace m2s;

extern "C" {

size_t m2s_triangle_count(size_t n_vertices, size_t n_indices, int has_indices, int topology) {
  const size_t n = has_indices ? n_indices : n_vertices;
  if (topology == M2S_TRIANGLE_LIST) return n / 3;       // itertools::tuples drops a trailing partial triple
  return n >= 3 ? n - 2 : 0;                             // tuple_windows
}

int m2s_version(void) { return M2S_VERSION_MAJOR * 1000 + M2S_VERSION_MINOR; }

int m2s_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

const char* m2s_last_error(void) { return g_err; }

void m2s_release_workspace(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  for (auto& kv : g_dev) {
    if (hipSetDevice(kv.first) != hipSuccess) continue;
    (void)hipDeviceSynchronize();
    if (kv.second.base) (void)hipFree(kv.second.base);
    kv.second.base = nullptr;
    kv.second.cap = 0;
  }
}

void m2s_grid_from_bounding_box(const float bbox_min[3], const float bbox_max[3], const uint64_t cell_count[3],
                                m2s_grid* grid) {
  // grid.rs:59-74: cell_size = (max - min) / count;  first = min + cell_size * 0.5   (f32, no FMA)
  for (int k = 0; k < 3; ++k) {
    const float fc = (float)cell_count[k];
    const float cs = (bbox_max[k] - bbox_min[k]) / fc;
    const float half = cs * 0.5f;
    grid->cell_size[k] = cs;
    grid->first_cell[k] = bbox_min[k] + half;
    grid->cell_count[k] = cell_count[k];
  }
}

void m2s_grid_cell_center(const m2s_grid* grid, const uint64_t cell[3], float out[3]) {
  for (int k = 0; k < 3; ++k) {  // grid.rs:135-141
    const float prod = (float)cell[k] * grid->cell_size[k];
    out[k] = grid->first_cell[k] + prod;
  }
}

uint64_t m2s_grid_cell_idx(const m2s_grid* grid, const uint64_t cell[3]) {
  return cell[2] + cell[1] * grid->cell_count[2] + cell[0] * grid->cell_count[1] * grid->cell_count[2];  // grid.rs:122-124
}

int m2s_generate_grid_sdf(const float* vertices, size_t n_vertices, const void* indices, size_t n_indices,
                          int index_bytes, int topology, const m2s_grid* grid, int sign_method, float* out,
                          const m2s_opts* opts) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_err[0] = 0;
  if (!grid) return fail(M2S_ERR_BAD_ARG, "grid is NULL");
  if (sign_method != M2S_SIGN_RAYCAST && sign_method != M2S_SIGN_NORMAL) return fail(M2S_ERR_BAD_ARG, "bad sign_method %d", sign_method);
  int rc = check_mesh_args(vertices, n_vertices, indices, n_indices, index_bytes, topology);
  if (rc) return rc;
  const uint64_t nx = grid->cell_count[0], ny = grid->cell_count[1], nz = grid->cell_count[2];
  if (nx >= 0x7fffffffull || ny >= 0x7fffffffull || nz >= 0x7fffffffull) return fail(M2S_ERR_BAD_ARG, "cell_count too large");
  const size_t n_tris = m2s_triangle_count(n_vertices, n_indices, indices != nullptr, topology);

  const uint64_t xb = opts ? opts->x_begin : 0, xe = (opts && opts->x_end) ? opts->x_end : nx;
  if (xb > xe || xe > nx) return fail(M2S_ERR_BAD_ARG, "x-slab [%llu,%llu) outside [0,%llu)", (unsigned long long)xb, (unsigned long long)xe, (unsigned long long)nx);
  const size_t slab_cells = (size_t)(xe - xb) * ny * nz;
  if (slab_cells == 0) {  // a grid without cells: the reference returns an empty Vec
    if (opts && opts->timings) memset(opts->timings, 0, sizeof(*opts->timings));
    return M2S_OK;
  }
  if (!out) return fail(M2S_ERR_BAD_ARG, "out is NULL");

  CallCtx c;
  DeviceState* st = nullptr;
  rc = resolve_ctx(opts, &c, &st);
  if (rc) return rc;

  GridParams g;
  for (int k = 0; k < 3; ++k) {
    g.first[k] = grid->first_cell[k];
    g.size[k] = grid->cell_size[k];
    g.n[k] = (uint32_t)grid->cell_count[k];
  }
  g.xb = (uint32_t)xb;
  g.xe = (uint32_t)xe;
  g.nzw = (uint32_t)((nz + 31) / 32);
  g.out_off = 0;

  size_t need = bvh_workspace_bytes(n_tris) + 4096;
  if (sign_method == M2S_SIGN_RAYCAST) need += sign_workspace_bytes(g);
  need += grid_distance_workspace_bytes(g);
  if (c.mem_kind == M2S_MEM_HOST)
    need += align_up(n_vertices * 12) + align_up(n_indices * (size_t)(indices ? index_bytes : 0)) + align_up(slab_cells * 4) + 1024;
  rc = ensure_capacity(*st, need);
  if (rc) return rc;
  Arena ws{st->base, st->cap, 0};

  int* d_err = ws.take<int>(16);
  M2S_HIP_CHECK(hipMemsetAsync(d_err, 0, 64, c.stream));
  StagedMesh sm;
  rc = stage_mesh(ws, c, vertices, n_vertices, indices, n_indices, index_bytes, &sm);
  if (rc) return rc;
  float* d_out = out;
  float* d_slab = nullptr;
  if (c.mem_kind == M2S_MEM_HOST) {
    d_slab = ws.take<float>(slab_cells);
    if (!d_slab) return fail(M2S_ERR_HIP, "internal: workspace");
    d_out = d_slab;
    g.out_off = (uint64_t)xb * ny * nz;  // kernels index the whole grid; the staging buffer holds the slab only
  }

  M2S_HIP_CHECK(hipEventRecord(st->ev[0], c.stream));
  DeviceMesh mesh;
  rc = build_device_mesh(ws, c.stream, sm.d_verts, n_vertices, sm.d_indices, n_indices, index_bytes, topology, n_tris, d_err, &mesh);
  if (rc) return rc;
  unsigned long long* d_stats = nullptr;
  if (getenv("M2S_STATS")) {
    d_stats = ws.take<unsigned long long>(8);
    unsigned long long init[8] = {0, 0, 0, 0, 0, 0, 0, (unsigned long long)atoi(getenv("M2S_STATS"))};
    M2S_HIP_CHECK(hipMemcpyAsync(d_stats, init, 64, hipMemcpyHostToDevice, c.stream));
    M2S_HIP_CHECK(hipStreamSynchronize(c.stream));
    mesh.stats = d_stats;
  }
  M2S_HIP_CHECK(hipEventRecord(st->ev[1], c.stream));
  const uint32_t* plane = nullptr;
  if (sign_method == M2S_SIGN_RAYCAST) {
    rc = build_grid_sign_plane(ws, c.stream, mesh, g, &plane);
    if (rc) return rc;
  }
  M2S_HIP_CHECK(hipEventRecord(st->ev[2], c.stream));
  rc = launch_grid_distance(ws, c.stream, mesh, g, sign_method == M2S_SIGN_RAYCAST ? MODE_UNSIGNED : MODE_NORMAL_FOLD, plane,
                            c.algorithm, d_out, d_err, st->ev[4]);
  if (rc) return rc;
  M2S_HIP_CHECK(hipEventRecord(st->ev[3], c.stream));
  if (c.mem_kind == M2S_MEM_HOST)
    M2S_HIP_CHECK(hipMemcpyAsync(out + (size_t)xb * ny * nz, d_slab, slab_cells * 4, hipMemcpyDeviceToHost, c.stream));
  if (d_stats) {
    unsigned long long h[8];
    M2S_HIP_CHECK(hipMemcpyAsync(h, d_stats, 64, hipMemcpyDeviceToHost, c.stream));
    M2S_HIP_CHECK(hipStreamSynchronize(c.stream));
    const double w = h[3] ? (double)h[3] : 1.0;
    fprintf(stderr, "[m2s stats] packets %llu: per packet box tests %.1f, oriented-bound tests %.1f, exact triangle tests %.1f\n",
            h[3], h[0] / w, h[1] / w, h[2] / w);
  }
  return finish_call(c, *st, d_err, c.timings, n_tris, slab_cells, sign_method == M2S_SIGN_RAYCAST, true);
}

int m2s_generate_sdf(const float* vertices, size_t n_vertices, const void* indices, size_t n_indices, int index_bytes,
                     int topology, const float* queries, size_t n_queries, int accel, int sign_method, float* out,
                     size_t* n_out, const m2s_opts* opts) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_err[0] = 0;
  if (n_out) *n_out = 0;
  if (accel < M2S_ACCEL_NONE || accel > M2S_ACCEL_RTREE_BVH) return fail(M2S_ERR_BAD_ARG, "bad accel %d", accel);
  if (sign_method != M2S_SIGN_RAYCAST && sign_method != M2S_SIGN_NORMAL) return fail(M2S_ERR_BAD_ARG, "bad sign_method %d", sign_method);
  int rc = check_mesh_args(vertices, n_vertices, indices, n_indices, index_bytes, topology);
  if (rc) return rc;
  if (n_queries && !queries) return fail(M2S_ERR_BAD_ARG, "queries is NULL");
  if (n_queries >= 0xffffffc0ull) return fail(M2S_ERR_BAD_ARG, "too many queries for one call");
  const size_t n_tris = m2s_triangle_count(n_vertices, n_indices, indices != nullptr, topology);
  if (n_tris == 0 && accel == M2S_ACCEL_RTREE_BVH) return M2S_OK;  // rtree_bvh.rs:104-106: vec![]
  if (n_tris == 0 && accel == M2S_ACCEL_RTREE && n_queries) return fail(M2S_ERR_EMPTY_MESH, "Rtree on a mesh without triangles (rtree.rs:117 unwrap on None)");
  if (n_queries == 0) return M2S_OK;
  if (!out) return fail(M2S_ERR_BAD_ARG, "out is NULL");

  CallCtx c;
  DeviceState* st = nullptr;
  rc = resolve_ctx(opts, &c, &st);
  if (rc) return rc;

  int mode = MODE_UNSIGNED, sign_src = SIGN_NONE, algorithm = c.algorithm;
  switch (accel) {
    case M2S_ACCEL_NONE:   // no acceleration structure in the reference either: literal brute force
      algorithm = 1;
      if (sign_method == M2S_SIGN_RAYCAST) { mode = MODE_UNSIGNED; sign_src = SIGN_XRAY_ALL; }
      else mode = MODE_NORMAL_FOLD;
      break;
    case M2S_ACCEL_BVH:
      if (sign_method == M2S_SIGN_RAYCAST) { mode = MODE_UNSIGNED; sign_src = SIGN_RAYS3; }
      else mode = MODE_NORMAL_FOLD;
      break;
    case M2S_ACCEL_RTREE: mode = MODE_NEAREST_NORMAL; break;
    default: mode = MODE_UNSIGNED; sign_src = SIGN_RAYS3; break;
  }

  size_t need = bvh_workspace_bytes(n_tris) + query_workspace_bytes(n_queries) + 4096;
  if (c.mem_kind == M2S_MEM_HOST)
    need += align_up(n_vertices * 12) + align_up(n_indices * (size_t)(indices ? index_bytes : 0)) + align_up(n_queries * 12) + align_up(n_queries * 4) + 1024;
  rc = ensure_capacity(*st, need);
  if (rc) return rc;
  Arena ws{st->base, st->cap, 0};
  int* d_err = ws.take<int>(16);
  M2S_HIP_CHECK(hipMemsetAsync(d_err, 0, 64, c.stream));
  StagedMesh sm;
  rc = stage_mesh(ws, c, vertices, n_vertices, indices, n_indices, index_bytes, &sm);
  if (rc) return rc;
  const float* d_q = queries;
  float* d_out = out;
  if (c.mem_kind == M2S_MEM_HOST) {
    float* dq = ws.take<float>(n_queries * 3);
    d_out = ws.take<float>(n_queries);
    if (!dq || !d_out) return fail(M2S_ERR_HIP, "internal: workspace");
    M2S_HIP_CHECK(hipMemcpyAsync(dq, queries, n_queries * 12, hipMemcpyHostToDevice, c.stream));
    d_q = dq;
  }
  M2S_HIP_CHECK(hipEventRecord(st->ev[0], c.stream));
  DeviceMesh mesh;
  rc = build_device_mesh(ws, c.stream, sm.d_verts, n_vertices, sm.d_indices, n_indices, index_bytes, topology, n_tris, d_err, &mesh);
  if (rc) return rc;
  M2S_HIP_CHECK(hipEventRecord(st->ev[1], c.stream));
  M2S_HIP_CHECK(hipEventRecord(st->ev[2], c.stream));
  rc = launch_query_distance(ws, c.stream, mesh, d_q, n_queries, mode, sign_src, algorithm, d_out, d_err);
  if (rc) return rc;
  M2S_HIP_CHECK(hipEventRecord(st->ev[3], c.stream));
  if (c.mem_kind == M2S_MEM_HOST)
    M2S_HIP_CHECK(hipMemcpyAsync(out, d_out, n_queries * 4, hipMemcpyDeviceToHost, c.stream));
  if (n_out) *n_out = n_queries;
  return finish_call(c, *st, d_err, c.timings, n_tris, n_queries, false);
}

}  // extern "C"
