// capi.hip — the C ABI of include/m2s.h: argument checks that mirror the reference's panics,
// per-device workspace, host<->device staging for the drop-in (host pointer) case, phase timing.
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/m2s.h"
#include "capi_internal.h"
#include "common.h"
#include "tuning.h"

namespace m2s {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

void clear_error() { g_err[0] = 0; }

std::mutex g_mu;
std::map<std::pair<int, int>, std::unique_ptr<DeviceState>> g_dev;   // (device, lane) -> context

DeviceState* find_state(int device, int lane) {
  std::lock_guard<std::mutex> lk(g_mu);
  auto& p = g_dev[{device, lane}];
  if (!p) p.reset(new DeviceState());
  return p.get();
}

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

// M2S_HOST_TIMES=1: where the host time of a call goes (first calls especially: runtime start, code objects, workspace, pinned ring).
struct HostClock {
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now(), last = t0;
  bool on = tuning().host_times != 0;
  char line[512] = "";
  void lap(const char* what) {
    if (!on) return;
    const auto now = std::chrono::steady_clock::now();
    const size_t n = strlen(line);
    snprintf(line + n, sizeof(line) - n, " %s %.2f ms,", what, std::chrono::duration<double, std::milli>(now - last).count());
    last = now;
  }
  void done(const char* call) {
    if (!on) return;
    fprintf(stderr, "[m2s host time] %s:%s total %.2f ms\n", call, line, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
  }
};

// HIP multiplexes a process's streams onto a few hardware queues (4 by default): every stream this library creates can
// end up sharing a queue with one of the caller's and serialise work the caller meant to overlap (measured: the two
// alternating piece streams of the sharded driver lost their overlap, 2.5 -> 3.0 ms per step, once the library held
// two streams of its own).  So streams are created only when a call really needs them.
static int init_state(DeviceState& s) {   // caller holds s.mu; the device is current
  if (!s.h_err) {
    for (auto& e : s.ev) M2S_HIP_CHECK(hipEventCreate(&e));
    M2S_HIP_CHECK(hipHostMalloc((void**)&s.h_err, 64, hipHostMallocDefault));
  }
  return 0;
}

static int own_stream(DeviceState& s, hipStream_t* out) {
  if (!s.stream) M2S_HIP_CHECK(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
  *out = s.stream;
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
  s.warmed_cap = 0;   // a fresh block: m2s_warmup has its pages to touch again
  const size_t want = bytes + bytes / 8 + (1u << 20);
  M2S_HIP_CHECK(hipMalloc((void**)&s.base, want));
  s.cap = want;
  return 0;
}

int select_scratch(DeviceState& s, hipStream_t stream) {
  if (s.active >= 0) { s.scratch[s.active].base = s.base; s.scratch[s.active].cap = s.cap; }
  int pick = -1;
  for (int i = 0; i < s.n_scratch; ++i)
    if (s.scratch[i].stream == stream) pick = i;
  if (pick < 0 && s.n_scratch < DeviceState::SCRATCH) {
    pick = s.n_scratch++;
    s.scratch[pick] = DeviceState::Scratch{};
    s.scratch[pick].stream = stream;
  }
  if (pick < 0) {   // more streams than blocks: hand the least recently used block over once its work has drained
    pick = 0;
    for (int i = 1; i < s.n_scratch; ++i)
      if (s.scratch[i].last_use < s.scratch[pick].last_use) pick = i;
    M2S_HIP_CHECK(hipDeviceSynchronize());
    s.scratch[pick].stream = stream;
  }
  s.scratch[pick].last_use = ++s.tick;
  s.base = s.scratch[pick].base;
  s.cap = s.scratch[pick].cap;
  s.active = pick;
  return 0;
}

void release_scratch(DeviceState& s) {
  if (s.active >= 0) { s.scratch[s.active].base = s.base; s.scratch[s.active].cap = s.cap; }
  else if (s.base) (void)hipFree(s.base);
  for (int i = 0; i < s.n_scratch; ++i)
    if (s.scratch[i].base) (void)hipFree(s.scratch[i].base);
  s.n_scratch = 0;
  s.active = -1;
  s.base = nullptr;
  s.cap = 0;
  s.warmed_cap = 0;
}

int resolve_ctx(const m2s_opts* opts, CallCtx* c, DeviceState** st) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(M2S_ERR_HIP, "no HIP device available: the MI355X kernels cannot run (there is no CPU fallback)");
  int dev = -1;
  if (opts) {
    // struct_size 0 or M2S_OPTS_V1_SIZE: a version-0.1 caller, the fields after stream_mode do not exist
    if (opts->struct_size != 0 && opts->struct_size < M2S_OPTS_V1_SIZE) return fail(M2S_ERR_BAD_ARG, "m2s_opts.struct_size too small");
    if (opts->struct_size >= sizeof(m2s_opts)) {
      if (opts->lane < 0 || opts->lane >= M2S_MAX_LANES) return fail(M2S_ERR_BAD_ARG, "m2s_opts.lane %d outside [0, %d)", opts->lane, M2S_MAX_LANES);
      if (opts->n_peer_out > M2S_MAX_PEERS) return fail(M2S_ERR_BAD_ARG, "m2s_opts.n_peer_out %u > %d", opts->n_peer_out, M2S_MAX_PEERS);
      if (opts->n_peer_out && !opts->peer_out) return fail(M2S_ERR_BAD_ARG, "m2s_opts.peer_out is NULL");
      if (opts->peer_mode < M2S_PEER_PUSH || opts->peer_mode > M2S_PEER_TRAIL) return fail(M2S_ERR_BAD_ARG, "bad m2s_opts.peer_mode");
      c->lane = opts->lane;
      c->peer_mode = opts->peer_mode;
      c->peers.n = opts->n_peer_out;
      for (uint32_t i = 0; i < opts->n_peer_out; ++i) {
        if (!opts->peer_out[i]) return fail(M2S_ERR_BAD_ARG, "m2s_opts.peer_out[%u] is NULL", i);
        c->peers.p[i] = opts->peer_out[i];
      }
    }
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
  if (c->peers.n && c->mem_kind != M2S_MEM_DEVICE) return fail(M2S_ERR_BAD_ARG, "m2s_opts.peer_out needs mem_kind == M2S_MEM_DEVICE");
  *st = find_state(dev, c->lane);
  c->lock = std::unique_lock<std::mutex>((*st)->mu);
  int rc = init_state(**st);
  if (rc) return rc;
  (*st)->early_planes = false;
  if (opts && (opts->stream || opts->stream_mode == 1)) c->stream = (hipStream_t)opts->stream;
  else if ((rc = own_stream(**st, &c->stream)) != 0) return rc;
  return select_scratch(**st, c->stream);
}

namespace {

int check_mesh_args(const float* vertices, size_t n_vertices, const void* indices, size_t n_indices, int index_bytes,
                    int topology) {
  if (topology != M2S_TRIANGLE_LIST && topology != M2S_TRIANGLE_STRIP) return fail(M2S_ERR_BAD_ARG, "bad topology %d", topology);
  if (n_vertices && !vertices) return fail(M2S_ERR_BAD_ARG, "vertices is NULL");
  if (indices && index_bytes != 2 && index_bytes != 4) return fail(M2S_ERR_BAD_ARG, "index_bytes must be 2 or 4");
  if (!indices && n_indices) return fail(M2S_ERR_BAD_ARG, "n_indices > 0 with NULL indices");
  if (n_vertices >= 0x7fffffffull || n_indices >= 0x17fffffffull) return fail(M2S_ERR_BAD_ARG, "mesh too large for 32-bit indexing");
  return 0;
}

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

// Stream the sign planes of this call are built on: the device's side stream (ordered after everything enqueued on the
// caller's stream so far), or the caller's stream itself for an asynchronous call.
static bool side_stream_wanted(bool synchronous_call) {
  // asynchronous calls are the pieces of a caller who overlaps them on streams of its own: leave the hardware queues to those
  return synchronous_call;
}
// Side work (seed passes, sign planes) runs BESIDE the build on streams of the LOWEST priority: the build is a chain of ~30 small,
// latency-bound kernels on the caller's stream and is what the cut lists wait for; the wide flooding passes should take the CUs it leaves.
static int create_side_stream(hipStream_t* s) {
  int lo = 0, hi = 0;
  if (hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && lo != hi) {   // lo = numerically greatest = lowest priority
    if (hipStreamCreateWithPriority(s, hipStreamNonBlocking, lo) == hipSuccess) return 0;
    (void)hipGetLastError();
  }
  M2S_HIP_CHECK(hipStreamCreateWithFlags(s, hipStreamNonBlocking));
  return 0;
}
static int ensure_side_stream(DeviceState& st) {
  if (!st.side_stream) {
    if (const int rc = create_side_stream(&st.side_stream)) return rc;
    M2S_HIP_CHECK(hipEventCreateWithFlags(&st.fork_ev, hipEventDisableTiming));
  }
  return 0;
}
static int sign_stream_begin(DeviceState& st, hipStream_t main, bool synchronous_call, hipStream_t* out) {
  st.planes_done = nullptr;
  *out = main;
  if (!side_stream_wanted(synchronous_call)) return 0;
  if (int rc = ensure_side_stream(st)) return rc;
  M2S_HIP_CHECK(hipEventRecord(st.fork_ev, main));
  M2S_HIP_CHECK(hipStreamWaitEvent(st.side_stream, st.fork_ev, 0));
  *out = st.side_stream;
  return 0;
}
// Records ev[2] ("planes done") on the stream that built them; the dominant launch will wait for it if that was the side stream.
static int sign_stream_end(DeviceState& st, hipStream_t main, hipStream_t used) {
  M2S_HIP_CHECK(hipEventRecord(st.ev[2], used));
  st.planes_done = used != main ? st.ev[2] : nullptr;
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
    (void)hipEventElapsedTime(&b, st.early_planes ? st.ev[5] : st.ev[1], st.ev[2]);   // early planes: their own span on side_stream2
    float sd = 0;
    if (had_seed_event) {
      // planes beside the seed passes: the seed phase then runs from the end of the build (ev[1]) to the dominant launch,
      // which also waits for the planes; sign_ms and seed_ms overlap and do not add up to the total
      (void)hipEventElapsedTime(&sd, (st.planes_done || st.early_planes) ? st.ev[1] : st.ev[2], st.ev[4]);
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
  if (e & ERRF_TRAIL_TIMEOUT) return fail(M2S_ERR_HIP, "the trailing peer push gave up waiting for the walk (M2S_PEER_TRAIL)");
  if (e & ERRF_BUILD_TIMEOUT) return fail(M2S_ERR_HIP, "the LBVH build gave up waiting for a neighbouring tile of its sort");
  if (e & ERRF_SPLIT_OVERFLOW) return fail(M2S_ERR_HIP, "internal: a suspended packet found no accumulator slot (split walk)");
  return M2S_OK;
}


// Seed passes + dominant launch (+ optional M2S_STATS counters); records ev[4] before and ev[3] after
// the dominant launch.
// Real x-ranges [first, first + count) layers of the slab's chunks (one for a contiguous slab).
static std::vector<std::pair<uint32_t, uint32_t>> slab_chunks(const GridParams& g) {
  std::vector<std::pair<uint32_t, uint32_t>> r;
  const uint32_t layers = g.xe - g.xb;
  if (g.chunk_log >= 31u) { r.emplace_back(g.xb, layers); return r; }
  const uint32_t C = 1u << g.chunk_log;
  for (uint32_t v = 0; v < layers; v += C) r.emplace_back(slab_x(g, v), std::min(C, layers - v));
  return r;
}

// M2S_STATS: traversal counters of the packet walk (a counting variant of k_packet), printed on stderr.
static int stats_begin(Arena& ws, hipStream_t stream, DeviceMesh* mesh, unsigned long long** d_stats) {
  *d_stats = nullptr;
  if (!tuning().stats) return 0;
#ifndef M2S_STATS_BUILD
  static bool told = false;
  if (!told) fprintf(stderr, "[m2s] M2S_STATS: this build carries no counting kernels; load mesh_to_sdf_amd/libm2s_stats.so (make -C mesh_to_sdf_amd/csrc stats) through M2S_LIB\n");
  told = true;
  return 0;
#endif
  *d_stats = ws.take<unsigned long long>(128);
  if (!*d_stats) return fail(M2S_ERR_HIP, "internal: workspace");
  unsigned long long init[128] = {0};
  init[7] = (unsigned long long)tuning().stats;
  M2S_HIP_CHECK(hipMemcpyAsync(*d_stats, init, sizeof(init), hipMemcpyHostToDevice, stream));
  M2S_HIP_CHECK(hipStreamSynchronize(stream));
  mesh->stats = *d_stats;
  return 0;
}
static int stats_end(hipStream_t stream, const unsigned long long* d_stats) {
  if (!d_stats) return 0;
  unsigned long long h[128];
  M2S_HIP_CHECK(hipMemcpyAsync(h, d_stats, sizeof(h), hipMemcpyDeviceToHost, stream));
  M2S_HIP_CHECK(hipStreamSynchronize(stream));
  const double w = h[3] ? (double)h[3] : 1.0;
  fprintf(stderr, "[m2s stats] packets %llu: per packet node tests %.1f, leaf pre-tests %.1f, exact triangle tests %.1f; node tests that pruned %.1f (by the slab term alone %.1f, by a sphere test %.1f)\n",
          h[3], h[0] / w, h[1] / w, h[2] / w, h[4] / w, h[5] / w, h[6] / w);
  fprintf(stderr, "[m2s stats]   longest packet: %llu node tests, %llu exact triangle tests (the launch cannot end before its chain does); most work units in one packet %llu, mean %.1f\n", h[72], h[73], h[74],
          (h[0] + h[1] + 4.0 * h[2]) / w);
  if (h[0] && h[1] && h[2])
    fprintf(stderr, "[m2s stats]   lanes served: %.1f of 64 want the node per node test, %.1f reach the leaf per pre-test, %.1f reach the triangle per exact evaluation (queued pairs per packet: %.0f)\n",
            (double)h[75] / h[0], (double)h[76] / h[1], (double)h[77] / h[2], (double)h[77] / w);
  {  // work units (node tests + pre-tests + 4 x exact evaluations) per packet, in octaves
    char line[512] = "";
    for (int o = 0; o < 24; ++o)
      if (h[80 + o]) snprintf(line + strlen(line), sizeof(line) - strlen(line), " [%u,%u): %llu", o ? 1u << o : 0u, 2u << o, h[80 + o]);
    fprintf(stderr, "[m2s stats]   packets by work units:%s\n", line);
  }
  if (h[104])
    fprintf(stderr, "[m2s stats]   k_cut: %llu waves (4 x 4 x 4 bricks each), node visits per wave %.1f (longest %llu), of them on nodes larger than the wave's block %.1f (%.1f %%), larger than 4 x the block %.1f (%.1f %%)\n",
            h[104], (double)h[105] / h[104], h[108], (double)h[106] / h[104], 100.0 * h[106] / (h[105] ? h[105] : 1), (double)h[107] / h[104], 100.0 * h[107] / (h[105] ? h[105] : 1));
  if (h[112])
    fprintf(stderr, "[m2s stats]   k_cut coarse level: %llu waves (4 x 4 x 4 blocks x one of 8 subtrees each), node visits per wave %.1f (longest %llu), on nodes larger than a block %.1f, larger than 4 x a block %.1f\n",
            h[112], (double)h[113] / h[112], h[116], (double)h[114] / h[112], (double)h[115] / h[112]);
  // grid path: by distance of the packet's first voxel to its seed triangle, in cells: [0,1) [1,2) [2,4) ... [64,inf)
  for (int bk = 0; bk < 8; ++bk) {
    const unsigned long long* q = h + 8 + 8 * bk;
    if (q[3])
      fprintf(stderr, "[m2s stats]   band %d (d >= %d cells): %5.1f %% of packets, node tests %.1f, pre-tests %.1f, exact %.1f (reached lanes per exact test %.1f), cut ranges %.1f (%.0f B of node records; all within 4 KiB for %.1f %% of the packets)\n", bk, bk ? 1 << (bk - 1) : 0,
              100.0 * q[3] / w, (double)q[0] / q[3], (double)q[1] / q[3], (double)q[2] / q[3], q[2] ? (double)q[5] / q[2] : 0.0, (double)q[4] / q[3],
              (double)q[6] / q[3], 100.0 * q[7] / q[3]);
  }
  return 0;
}

int run_grid_distance(Arena& ws, const CallCtx& c, DeviceState& st, DeviceMesh mesh, const GridParams& g, int sign_method,
                      const uint32_t* plane, float* d_out, int* d_err) {
  unsigned long long* d_stats = nullptr;
  int rc = stats_begin(ws, c.stream, &mesh, &d_stats);
  if (rc) return rc;
  // peers: M2S_PEER_STORE hands them to the walk's epilogue; M2S_PEER_PUSH (run_grid_distance_push) never comes here with any
  rc = launch_grid_distance(ws, c.stream, mesh, g, sign_method == M2S_SIGN_RAYCAST ? MODE_UNSIGNED : MODE_NORMAL_FOLD,
                            plane, c.algorithm, d_out, d_err, st.ev[4], st.planes_done, !c.sync,
                            st.have_raw_seeds ? &st.raw_seeds : nullptr, st.have_raw_seeds ? st.seeds_done : nullptr,
                            c.peers.n ? &c.peers : nullptr);
  st.have_raw_seeds = false;
  if (rc) return rc;
  M2S_HIP_CHECK(hipEventRecord(st.ev[3], c.stream));
  return stats_end(c.stream, d_stats);
}


// M2S_PEER_PUSH: the slab is walked in x-pieces on the call's stream; every finished piece is copied to all peers by one
// wide-store kernel on the context's copy stream while the next piece is being walked, so that only the last piece's
// push is exposed (xGMI is point-to-point: 7 peers = 7 links in parallel, ~1/8 of the grid over each).  Seeds and cut
// lists are prepared once for the slab (the pieces are nothing but walks).  Records ev[4] before the first and ev[3]
// after the last walk; on return the call's stream also waits for the last push.
int run_grid_distance_push(Arena& ws, const CallCtx& c, DeviceState& st, const DeviceMesh& mesh, const GridParams& g,
                           int sign_method, const uint32_t* plane, float* d_out, int* d_err, uint32_t* pieces_out) {
  const uint64_t row = (uint64_t)g.n[1] * g.n[2];
  const uint32_t layers = g.xe - g.xb;
  const uint32_t want_pieces = tuning().push_pieces;
  const uint64_t bx = 2ull << g.bl[0];                              // whole cut-list blocks (2 bricks) along x
  uint64_t lpp = (layers + want_pieces - 1) / want_pieces;
  if (g.chunk_log < 31u) lpp = 1ull << g.chunk_log;                // interleaved slab: a piece = a chunk (contiguous in the grid; a multiple of 4 bricks)
  else lpp = std::max<uint64_t>(bx, (lpp + bx - 1) / bx * bx);
  // thin pieces walk badly (one launch per piece, each ending in a partly filled tail; 512^3 x blob-100k cut into pieces of
  // 64 / 32 / 16 layers: 10.9 / 11.6 / 16.1 ms of walks in total): at least 8 bricks of layers per piece
  if (g.chunk_log >= 31u) {
    lpp = std::max<uint64_t>(lpp, 8ull << g.bl[0]);
    if ((uint64_t)layers * row * 4 < (8u << 20)) lpp = std::max<uint64_t>(lpp, layers);   // small slabs: one piece
  }
  const uint32_t pieces = (uint32_t)((layers + lpp - 1) / lpp);
  *pieces_out = pieces;
  if (!st.copy_stream) M2S_HIP_CHECK(hipStreamCreateWithFlags(&st.copy_stream, hipStreamNonBlocking));
  while (st.piece_events.size() < (size_t)pieces + 1) {
    hipEvent_t e;
    M2S_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    st.piece_events.push_back(e);
  }
  const int mode = sign_method == M2S_SIGN_RAYCAST ? MODE_UNSIGNED : MODE_NORMAL_FOLD;
  GridWalkPlan plan;
  if (st.have_raw_seeds) M2S_HIP_CHECK(hipStreamWaitEvent(c.stream, st.seeds_done, 0));
  int rc = prepare_grid_walk(ws, c.stream, mesh, g, c.algorithm, !c.sync, &plan, st.have_raw_seeds ? &st.raw_seeds : nullptr);
  st.have_raw_seeds = false;
  if (rc) return rc;
  if (st.planes_done) M2S_HIP_CHECK(hipStreamWaitEvent(c.stream, st.planes_done, 0));
  M2S_HIP_CHECK(hipEventRecord(st.ev[4], c.stream));
  for (uint32_t i = 0; i < pieces; ++i) {
    GridParams gp = g;
    gp.xb = slab_x(g, (uint32_t)(i * lpp));                          // a piece is contiguous in the grid
    gp.xe = gp.xb + (uint32_t)std::min<uint64_t>(lpp, layers - i * lpp);
    gp.chunk_log = 31;
    gp.period = 0;
    rc = launch_grid_walk(c.stream, mesh, gp, mode, plane, c.algorithm, plan, (uint32_t)((i * lpp) >> g.bl[0]), d_out, d_err);
    if (rc) return rc;
    if (i + 1 == pieces) M2S_HIP_CHECK(hipEventRecord(st.ev[3], c.stream));
    M2S_HIP_CHECK(hipEventRecord(st.piece_events[i], c.stream));
    M2S_HIP_CHECK(hipStreamWaitEvent(st.copy_stream, st.piece_events[i], 0));
    rc = launch_push_cells(st.copy_stream, d_out, c.peers, (uint64_t)gp.xb * row, (uint64_t)(gp.xe - gp.xb) * row);
    if (rc) return rc;
  }
  M2S_HIP_CHECK(hipEventRecord(st.piece_events[pieces], st.copy_stream));
  M2S_HIP_CHECK(hipStreamWaitEvent(c.stream, st.piece_events[pieces], 0));
  return 0;
}

// M2S_PEER_TRAIL: ONE walk over the whole slab (no pieces, no tails between them) with its packets ordered so that the
// x-layers finish in order (super-bricks at most 2 bricks wide in x); the walk counts finished packets per unit of 2 bricks
// of layers, and k_push_trailing — a few workgroups on the copy stream, launched AFTER the walk so that a runtime that maps
// both streams onto one hardware queue merely serialises them — pushes every unit to the peers as soon as it is complete.
// Only the last unit's push (1/8 of a 64-layer slab) is exposed, and every store instruction still carries 1 KiB.
int run_grid_distance_trail(Arena& ws, const CallCtx& c, DeviceState& st, const DeviceMesh& mesh, const GridParams& g0,
                            int sign_method, const uint32_t* plane, float* d_out, int* d_err) {
  GridParams g = g0;
  g.xl_cap = trail_unit_log(g) + 1u;
  set_super_brick_magic(g);
  if (!st.copy_stream) M2S_HIP_CHECK(hipStreamCreateWithFlags(&st.copy_stream, hipStreamNonBlocking));
  while (st.piece_events.size() < 2) {
    hipEvent_t e;
    M2S_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    st.piece_events.push_back(e);
  }
  const int mode = sign_method == M2S_SIGN_RAYCAST ? MODE_UNSIGNED : MODE_NORMAL_FOLD;
  GridWalkPlan plan;
  if (st.have_raw_seeds) M2S_HIP_CHECK(hipStreamWaitEvent(c.stream, st.seeds_done, 0));
  int rc = prepare_grid_walk(ws, c.stream, mesh, g, c.algorithm, !c.sync, &plan, st.have_raw_seeds ? &st.raw_seeds : nullptr);
  st.have_raw_seeds = false;
  if (rc) return rc;
  const uint64_t row = (uint64_t)g.n[1] * g.n[2];
  const bool trailing = !plan.lane_walk && plan.brute_acc == nullptr && c.algorithm == 0 && mesh.n_nodes != 0 && g.chunk_log >= 31u;   // k_packet counts its packets; the other walks do not
  PeerOut walk_peers{};
  if (trailing) {
    const size_t counters = (size_t)trail_units(g) * (trail_rows(g) + 1u);
    uint32_t* progress = ws.take<uint32_t>(counters);
    if (!progress) return fail(M2S_ERR_HIP, "internal: workspace");
    M2S_HIP_CHECK(hipMemsetAsync(progress, 0, counters * sizeof(uint32_t), c.stream));
    walk_peers.progress = progress;
    walk_peers.unit_log = trail_unit_log(g);
    walk_peers.rows = trail_rows(g);
    walk_peers.units = trail_units(g);
  }
  if (st.planes_done) M2S_HIP_CHECK(hipStreamWaitEvent(c.stream, st.planes_done, 0));
  M2S_HIP_CHECK(hipEventRecord(st.piece_events[0], c.stream));          // counters are zero, inputs are ready
  M2S_HIP_CHECK(hipEventRecord(st.ev[4], c.stream));
  rc = launch_grid_walk(c.stream, mesh, g, mode, plane, c.algorithm, plan, 0, d_out, d_err, trailing ? &walk_peers : nullptr);
  if (rc) return rc;
  M2S_HIP_CHECK(hipEventRecord(st.ev[3], c.stream));
  if (trailing) {
    PeerOut push = c.peers;
    push.progress = walk_peers.progress;
    push.unit_log = walk_peers.unit_log;
    push.rows = walk_peers.rows;
    push.units = walk_peers.units;
    M2S_HIP_CHECK(hipStreamWaitEvent(st.copy_stream, st.piece_events[0], 0));
    rc = launch_push_trailing(st.copy_stream, d_out, push, g, d_err);
  } else {
    M2S_HIP_CHECK(hipStreamWaitEvent(st.copy_stream, st.ev[3], 0));
    for (const auto& ch : slab_chunks(g)) {
      rc = launch_push_cells(st.copy_stream, d_out, c.peers, (uint64_t)ch.first * row - g.out_off, (uint64_t)ch.second * row);
      if (rc) return rc;
    }
  }
  if (rc) return rc;
  M2S_HIP_CHECK(hipEventRecord(st.piece_events[1], st.copy_stream));
  M2S_HIP_CHECK(hipStreamWaitEvent(c.stream, st.piece_events[1], 0));
  return 0;
}

// AccelerationMethod + SignMethod -> kernel mode, sign source, algorithm (include/m2s.h lists the rules).
void select_generic_mode(int accel, int sign_method, int req_algorithm, int* mode, int* sign_src, int* algorithm) {
  *mode = MODE_UNSIGNED;
  *sign_src = SIGN_NONE;
  *algorithm = req_algorithm;
  switch (accel) {
    case M2S_ACCEL_NONE:   // no acceleration structure in the reference either: literal brute force
      *algorithm = 1;
      if (sign_method == M2S_SIGN_RAYCAST) { *mode = MODE_UNSIGNED; *sign_src = SIGN_XRAY_ALL; }
      else *mode = MODE_NORMAL_FOLD;
      break;
    case M2S_ACCEL_BVH:
      if (sign_method == M2S_SIGN_RAYCAST) { *mode = MODE_UNSIGNED; *sign_src = SIGN_RAYS3; }
      else *mode = MODE_NORMAL_FOLD;
      break;
    case M2S_ACCEL_RTREE: *mode = MODE_NEAREST_NORMAL; break;
    default: *mode = MODE_UNSIGNED; *sign_src = SIGN_RAYS3; break;
  }
}

int fill_grid_params(const m2s_grid* grid, const m2s_opts* opts, GridParams* g, size_t* slab_cells) {
  const uint64_t gx = grid->cell_count[0], gy = grid->cell_count[1], gz = grid->cell_count[2];
  if (gx >= 0x7fffffffull || gy >= 0x7fffffffull || gz >= 0x7fffffffull) return fail(M2S_ERR_BAD_ARG, "cell_count too large");
  const uint64_t xb = opts ? opts->x_begin : 0, xe = (opts && opts->x_end) ? opts->x_end : gx;
  // sign.hip counts the grid lines a triangle's window covers (and hands them out in chunks) in 32 bits
  if (gx * gy >= (1ull << 32) || gx * gz >= (1ull << 32) || gy * gz >= (1ull << 32))
    return fail(M2S_ERR_BAD_ARG, "grid face with 2^32 or more lines (cell_count %llu x %llu x %llu)", (unsigned long long)gx, (unsigned long long)gy, (unsigned long long)gz);
  if (xb > xe || xe > gx) return fail(M2S_ERR_BAD_ARG, "x-slab [%llu,%llu) outside [0,%llu)", (unsigned long long)xb, (unsigned long long)xe, (unsigned long long)gx);
  for (int k = 0; k < 3; ++k) {
    g->first[k] = grid->first_cell[k];
    g->size[k] = grid->cell_size[k];
    g->n[k] = (uint32_t)grid->cell_count[k];
  }
  g->xb = (uint32_t)xb;
  g->xe = (uint32_t)xe;
  g->nzw = (uint32_t)((gz + 31) / 32);
  g->out_off = 0;
  g->xl_cap = 0;
  g->chunk_log = 31;
  g->period = 0;
  choose_brick_shape(g->size, g->bl);
  uint64_t layers = xe - xb;
  const uint64_t period = (opts && opts->struct_size >= sizeof(m2s_opts)) ? opts->x_period : 0;
  if (period != 0 && layers != 0) {
    // interleaved slab: chunks [xb + j * period, + C), C = x_end - x_begin a power of two that holds whole cut-list waves
    // (4 bricks along x), every chunk inside the grid; the virtual slab is the chunks laid end to end
    // a chunk holds whole waves of the cut-list kernel and whole push pieces: 4 packet bricks along x (m2s.h).  (A chunk of ONE
    // brick used to pass this check; the peer push then rounded its piece up to two bricks and walked two chunks as one
    // contiguous range — wrong cells, silently.)
    const uint64_t C = layers, brick_layers = 4ull << g->bl[0];
    uint32_t clog = 0;
    while ((1ull << clog) < C) ++clog;
    if ((1ull << clog) != C || C % brick_layers != 0 || period % C != 0 || period < C || period >= (1ull << 31))
      return fail(M2S_ERR_BAD_ARG, "x_period %llu with a chunk of %llu layers: the chunk must be a power of two and a multiple of 4 packet bricks (%llu layers), the period a multiple of the chunk",
                  (unsigned long long)period, (unsigned long long)C, (unsigned long long)brick_layers);
    uint64_t chunks = 0;
    while (xb + chunks * period < gx) {
      if (xb + chunks * period + C > gx) return fail(M2S_ERR_BAD_ARG, "x_period: the chunk at x = %llu leaves the grid (%llu layers)", (unsigned long long)(xb + chunks * period), (unsigned long long)gx);
      ++chunks;
    }
    g->chunk_log = clog;
    g->period = (uint32_t)period;
    layers = chunks * C;
    g->xe = g->xb + (uint32_t)layers;       // end of the VIRTUAL slab (common.h slab_x)
  }
  set_super_brick_magic(*g);
  *slab_cells = (size_t)layers * gy * gz;
  return 0;
}



}  // namespace

// ---- host-pointer result path ----------------------------------------------------------------------
// The drop-in call returns the grid in caller-owned PAGEABLE memory (the reference returns a Vec<f32>).  One
// hipMemcpy of 512 MiB into pageable memory runs at ~10 GB/s and only starts when the last voxel is done
// (55 of the 73 ms of such a call).  Instead the slab is computed in x-pieces; each finished piece crosses PCIe
// into a pinned ring buffer on a second stream while the next piece computes, and host threads move it from
// the ring into the caller's array (spreading its first-touch page faults as well).
struct ParallelCopy {
  std::vector<std::thread> th;
  void start(char* dst, const char* src, size_t n) {
    const unsigned hw = std::thread::hardware_concurrency();
    const size_t parts = std::max<size_t>(1, std::min<size_t>({(size_t)8, (size_t)(hw ? hw : 1), n / (4u << 20) + 1}));
    const size_t per = (n + parts - 1) / parts;
    for (size_t k = 0; k < parts; ++k) {
      const size_t o = k * per;
      if (o >= n) break;
      const size_t l = std::min(per, n - o);
      th.emplace_back([=]() { memcpy(dst + o, src + o, l); });
    }
  }
  void join() {
    for (auto& t : th) t.join();
    th.clear();
  }
  ~ParallelCopy() { join(); }
};

int ensure_ring(DeviceState& st, size_t bytes) {
  if (!st.copy_stream) M2S_HIP_CHECK(hipStreamCreateWithFlags(&st.copy_stream, hipStreamNonBlocking));
  if (bytes <= st.ring_bytes) return 0;
  for (auto& r : st.ring) {
    if (r) (void)hipHostFree(r);
    r = nullptr;
  }
  st.ring_bytes = 0;
  for (auto& r : st.ring) M2S_HIP_CHECK(hipHostMalloc((void**)&r, bytes, hipHostMallocDefault));
  st.ring_bytes = bytes;
  return 0;
}

// Computes the slab [g.xb, g.xe) into d_slab piece by piece and streams it to `out_slab` (host, slab-relative).
// Records ev[3] after the last piece's kernels.  *pieces_out = number of dominant launches.
int run_grid_distance_to_host(Arena& ws, const CallCtx& c, DeviceState& st, const DeviceMesh& mesh, const GridParams& g,
                              int sign_method, const uint32_t* plane, float* d_slab, int* d_err, float* out_slab,
                              uint32_t* pieces_out) {
  const uint64_t row = (uint64_t)g.n[1] * g.n[2];
  const uint32_t layers = g.xe - g.xb;
  const size_t piece_mb = tuning().host_piece_mb;   // 32 MiB; 16: 15.5, 32: 13.7, 64: 14.4, 128: 15.9 ms for the 512^3 call
  uint64_t lpp = row ? std::max<uint64_t>(1, (piece_mb << 20) / 4 / row) : layers;
  if ((uint64_t)layers * row * 4 >= (16u << 20)) lpp = std::min<uint64_t>(lpp, (layers + 3) / 4);   // >= 4 pieces: something to overlap
  const uint64_t bx = 2ull << g.bl[0];
  lpp = std::max<uint64_t>(bx, lpp / bx * bx);                    // whole cut-list blocks (2 bricks) along x
  const uint32_t pieces = (uint32_t)((layers + lpp - 1) / lpp);
  *pieces_out = pieces;
  int rc = ensure_ring(st, (size_t)std::min<uint64_t>(lpp, layers) * row * 4);
  if (rc) return rc;
  std::vector<hipEvent_t> done(pieces), copied(pieces);
  for (uint32_t i = 0; i < pieces; ++i) {
    M2S_HIP_CHECK(hipEventCreateWithFlags(&done[i], hipEventDisableTiming));
    M2S_HIP_CHECK(hipEventCreateWithFlags(&copied[i], hipEventDisableTiming));
  }
  auto cleanup = [&]() {
    for (uint32_t i = 0; i < pieces; ++i) { (void)hipEventDestroy(done[i]); (void)hipEventDestroy(copied[i]); }
  };
  const int mode = sign_method == M2S_SIGN_RAYCAST ? MODE_UNSIGNED : MODE_NORMAL_FOLD;
  // seeds and cut lists once for the whole slab; the pieces are then nothing but walks (a piece of its own would pay the
  // 0.5 ms of dependent small kernels again: 16.6 -> 14 ms for the 512^3 call)
  GridWalkPlan plan;
  if (st.have_raw_seeds && hipStreamWaitEvent(c.stream, st.seeds_done, 0) != hipSuccess) { cleanup(); return fail(M2S_ERR_HIP, "hipStreamWaitEvent failed"); }
  rc = prepare_grid_walk(ws, c.stream, mesh, g, c.algorithm, false, &plan, st.have_raw_seeds ? &st.raw_seeds : nullptr);
  st.have_raw_seeds = false;
  if (rc) { cleanup(); return rc; }
  if (st.planes_done && hipStreamWaitEvent(c.stream, st.planes_done, 0) != hipSuccess) { cleanup(); return fail(M2S_ERR_HIP, "hipStreamWaitEvent failed"); }
  for (uint32_t i = 0; i < pieces; ++i) {                          // all kernels first: the GPU never waits for the host
    GridParams gp = g;
    gp.xb = g.xb + (uint32_t)(i * lpp);
    gp.xe = (uint32_t)std::min<uint64_t>(g.xe, gp.xb + lpp);
    rc = launch_grid_walk(c.stream, mesh, gp, mode, plane, c.algorithm, plan, (uint32_t)((i * lpp) >> g.bl[0]), d_slab, d_err);
    if (rc) { cleanup(); return rc; }
    if (hipEventRecord(done[i], c.stream) != hipSuccess) { cleanup(); return fail(M2S_ERR_HIP, "hipEventRecord failed"); }
  }
  if (hipEventRecord(st.ev[3], c.stream) != hipSuccess) { cleanup(); return fail(M2S_ERR_HIP, "hipEventRecord failed"); }
  ParallelCopy mover[DeviceState::RING];
  auto piece_span = [&](uint32_t i, size_t* off, size_t* bytes) {
    const uint64_t x0 = (uint64_t)i * lpp, x1 = std::min<uint64_t>(layers, x0 + lpp);
    *off = (size_t)(x0 * row);
    *bytes = (size_t)((x1 - x0) * row * 4);
  };
  hipError_t e = hipSuccess;
  for (uint32_t i = 0; i <= pieces && e == hipSuccess; ++i) {
    if (i < pieces) {
      const int b = i % DeviceState::RING;
      mover[b].join();                                             // piece i-RING has left this buffer
      size_t off, bytes;
      piece_span(i, &off, &bytes);
      e = hipStreamWaitEvent(st.copy_stream, done[i], 0);
      if (e == hipSuccess) e = hipMemcpyAsync(st.ring[b], d_slab + off, bytes, hipMemcpyDeviceToHost, st.copy_stream);
      if (e == hipSuccess) e = hipEventRecord(copied[i], st.copy_stream);
    }
    if (i >= 1 && e == hipSuccess) {
      const uint32_t j = i - 1;
      size_t off, bytes;
      piece_span(j, &off, &bytes);
      e = hipEventSynchronize(copied[j]);
      if (e == hipSuccess) mover[j % DeviceState::RING].start(reinterpret_cast<char*>(out_slab + off), st.ring[j % DeviceState::RING], bytes);
    }
  }
  for (auto& m : mover) m.join();
  cleanup();
  if (e != hipSuccess) return fail(M2S_ERR_HIP, "pipelined device-to-host copy failed: %s", hipGetErrorString(e));
  return 0;
}


// Large host<->device transfers of caller-owned pageable arrays at PCIe speed: through the pinned ring, with
// host threads doing the pageable side.  `stream` is the call's stream: H2D data is visible to work enqueued
// on it afterwards; D2H starts after everything enqueued on it so far.  Both return with the copy complete.
constexpr size_t STAGE_CHUNK = 32u << 20;
constexpr size_t STAGE_MIN = 8u << 20;      // below this a plain hipMemcpyAsync is as good

int staged_h2d(DeviceState& st, hipStream_t stream, char* d_dst, const char* h_src, size_t bytes) {
  if (bytes < STAGE_MIN) {
    M2S_HIP_CHECK(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, stream));
    return 0;
  }
  int rc = ensure_ring(st, STAGE_CHUNK);
  if (rc) return rc;
  hipEvent_t sent[DeviceState::RING];
  for (auto& e : sent) M2S_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  hipError_t e = hipSuccess;
  const size_t n = (bytes + STAGE_CHUNK - 1) / STAGE_CHUNK;
  ParallelCopy mover;
  for (size_t i = 0; i < n && e == hipSuccess; ++i) {
    const int b = (int)(i % DeviceState::RING);
    const size_t off = i * STAGE_CHUNK, len = std::min(STAGE_CHUNK, bytes - off);
    if (i >= (size_t)DeviceState::RING) e = hipEventSynchronize(sent[b]);     // the ring slot has been read by the DMA
    if (e != hipSuccess) break;
    mover.start(st.ring[b], h_src + off, len);
    mover.join();
    e = hipMemcpyAsync(d_dst + off, st.ring[b], len, hipMemcpyHostToDevice, st.copy_stream);
    if (e == hipSuccess) e = hipEventRecord(sent[b], st.copy_stream);
  }
  if (e == hipSuccess) e = hipStreamSynchronize(st.copy_stream);   // complete (and the ring is free) before returning
  for (auto& ev : sent) (void)hipEventDestroy(ev);
  if (e != hipSuccess) return fail(M2S_ERR_HIP, "staged host-to-device copy failed: %s", hipGetErrorString(e));
  (void)stream;   // the data is resident before anything else is enqueued on `stream`
  return 0;
}

int staged_d2h(DeviceState& st, hipStream_t stream, char* h_dst, const char* d_src, size_t bytes) {
  if (bytes < STAGE_MIN) {
    M2S_HIP_CHECK(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, stream));
    return 0;
  }
  int rc = ensure_ring(st, STAGE_CHUNK);
  if (rc) return rc;
  hipEvent_t ready, got[DeviceState::RING];
  M2S_HIP_CHECK(hipEventCreateWithFlags(&ready, hipEventDisableTiming));
  for (auto& e : got) M2S_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  hipError_t e = hipEventRecord(ready, stream);
  if (e == hipSuccess) e = hipStreamWaitEvent(st.copy_stream, ready, 0);
  const size_t n = (bytes + STAGE_CHUNK - 1) / STAGE_CHUNK;
  ParallelCopy mover[DeviceState::RING];
  for (size_t i = 0; i <= n && e == hipSuccess; ++i) {
    if (i < n) {
      const int b = (int)(i % DeviceState::RING);
      mover[b].join();
      const size_t off = i * STAGE_CHUNK, len = std::min(STAGE_CHUNK, bytes - off);
      e = hipMemcpyAsync(st.ring[b], d_src + off, len, hipMemcpyDeviceToHost, st.copy_stream);
      if (e == hipSuccess) e = hipEventRecord(got[b], st.copy_stream);
    }
    if (i >= 1 && e == hipSuccess) {
      const size_t j = i - 1;
      const int b = (int)(j % DeviceState::RING);
      const size_t off = j * STAGE_CHUNK, len = std::min(STAGE_CHUNK, bytes - off);
      e = hipEventSynchronize(got[b]);
      if (e == hipSuccess) mover[b].start(h_dst + off, st.ring[b], len);
    }
  }
  for (auto& m : mover) m.join();
  (void)hipEventDestroy(ready);
  for (auto& ev : got) (void)hipEventDestroy(ev);
  if (e != hipSuccess) return fail(M2S_ERR_HIP, "staged device-to-host copy failed: %s", hipGetErrorString(e));
  return 0;
}


// Same pipeline with a file as the sink: the ring slot is written out directly (no host copy of the whole array).
int staged_d2h_to_file(DeviceState& st, hipStream_t stream, FILE* f, const char* d_src, size_t bytes) {
  int rc = ensure_ring(st, STAGE_CHUNK);
  if (rc) return rc;
  hipEvent_t ready, got[DeviceState::RING];
  M2S_HIP_CHECK(hipEventCreateWithFlags(&ready, hipEventDisableTiming));
  for (auto& e : got) M2S_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  hipError_t e = hipEventRecord(ready, stream);
  if (e == hipSuccess) e = hipStreamWaitEvent(st.copy_stream, ready, 0);
  const size_t n = (bytes + STAGE_CHUNK - 1) / STAGE_CHUNK;
  bool io_ok = true;
  for (size_t i = 0; i < n + 2 && e == hipSuccess; ++i) {
    if (i >= 2) {                                                 // write chunk i-2 (its slot is reused by chunk i+1)
      const size_t j = i - 2;
      const int b = (int)(j % DeviceState::RING);
      const size_t off = j * STAGE_CHUNK, len = std::min(STAGE_CHUNK, bytes - off);
      e = hipEventSynchronize(got[b]);
      if (e == hipSuccess && io_ok) io_ok = fwrite(st.ring[b], 1, len, f) == len;
    }
    if (i < n) {
      const int b = (int)(i % DeviceState::RING);
      const size_t off = i * STAGE_CHUNK, len = std::min(STAGE_CHUNK, bytes - off);
      e = hipMemcpyAsync(st.ring[b], d_src + off, len, hipMemcpyDeviceToHost, st.copy_stream);
      if (e == hipSuccess) e = hipEventRecord(got[b], st.copy_stream);
    }
  }
  (void)hipEventDestroy(ready);
  for (auto& ev : got) (void)hipEventDestroy(ev);
  if (e != hipSuccess) return fail(M2S_ERR_HIP, "staged device-to-host copy failed: %s", hipGetErrorString(e));
  return io_ok ? 0 : M2S_ERR_IO;
}

}  // namespace m2s

// Persistent mesh: triangle records + LBVH resident on one device, reusable across calls
// (it also lets one step be split into several slab calls whose all-gathers
// overlap the next slab's compute).  Caches the sign planes of the last grid it was used with.
struct m2s_mesh {
  std::mutex mu;    // the mesh's own state (cached planes, pending timings); taken before the context's lock
  int device = -1;
  int lane = 0;     // context whose spare blocks this mesh recycles
  char* mem = nullptr;
  size_t mem_bytes = 0;
  m2s::DeviceMesh dm{};
  size_t n_tris = 0;
  float build_ms = 0.0f;
  bool plane_valid = false;
  m2s_grid plane_grid{};
  char* plane_mem = nullptr;
  size_t plane_bytes = 0;
  const uint32_t* plane = nullptr;
  hipEvent_t plane_ready = nullptr;   // recorded after the sign planes were built, on plane_stream
  hipStream_t plane_stream = nullptr;
  bool multi_stream = false;          // the planes have been read from a stream other than plane_stream
  struct Pending {
    hipEvent_t a, b;
    uint64_t units;
    uint32_t launches;
  };
  std::vector<Pending> pending;  // dominant-launch event pairs of asynchronous calls, not yet read
  std::vector<hipEvent_t> free_events;
  // durations of asynchronous calls whose events have already been read and recycled (the list above stays bounded
  // however long the mesh lives: completed pairs are folded in here at the start of every asynchronous call)
  double acc_ms = 0.0;
  uint64_t acc_units = 0;
  uint32_t acc_launches = 0;
  int* d_err_async = nullptr;     // device error word of asynchronous calls (inside `mem`), read by m2s_mesh_drain_timings
  bool async_readers = false;     // asynchronous walks may still be reading the sign planes
  // walks of the tree that a later call on ANOTHER stream cannot see as finished (calls on one stream are ordered by it): the last call was
  // asynchronous (tree_last_async), or an asynchronous call was followed by a call on a different stream (tree_async_other); both end with
  // m2s_mesh_drain_timings or a device synchronisation
  bool tree_last_async = false, tree_async_other = false;
  hipStream_t tree_stream = nullptr;
  bool tree_used = false;
};

// The leaf size a call wants its tree to have (grid_leaf_max / query_leaf_max): the resident tree is re-marked when it differs — one
// 5 us launch on the call's stream.  Walks that this stream does not order before the launch must have finished first.
// INVARIANT the multi-stream cases rest on: k_releaf only moves the `tri` marks between two valid cuts of the same tree (a subtree of at
// most leaf_max triangles is walked as one leaf, and every mixture of old and new marks is still a cut: each root-to-leaf path meets a
// mark), and the walks take the exact minimum over whatever leaves they meet.  So a call on another stream that finds the host-side size
// already equal to its wish, while the re-marking launch is still in flight, computes the same bits at a slightly different cost.
static int remark_leaves(m2s_mesh* m, const m2s::CallCtx& c, uint32_t want) {
  using namespace m2s;
  const bool other_stream = m->tree_used && m->tree_stream != c.stream;
  if (want != m->dm.leaf_max) {
    if (m->tree_async_other || (other_stream && m->tree_last_async)) {
      M2S_HIP_CHECK(hipDeviceSynchronize());
      m->tree_async_other = false;
      m->tree_last_async = false;
    }
    const int rc = set_leaf_size(c.stream, &m->dm, want);
    if (rc) return rc;
  }
  if (other_stream && m->tree_last_async) m->tree_async_other = true;
  m->tree_stream = c.stream;
  m->tree_used = true;
  m->tree_last_async = !c.sync;
  return 0;
}

// Folds the finished entries of m->pending into the accumulators and recycles their events.
static void reap_pending(m2s_mesh* m, bool wait) {
  size_t keep = 0;
  for (size_t i = 0; i < m->pending.size(); ++i) {
    auto& p = m->pending[i];
    const hipError_t q = wait ? hipEventSynchronize(p.b) : hipEventQuery(p.b);
    if (q == hipSuccess) {
      float ms = 0.0f;
      (void)hipEventElapsedTime(&ms, p.a, p.b);
      m->acc_ms += ms;
      m->acc_units += p.units;
      m->acc_launches += p.launches;
      m->free_events.push_back(p.a);
      m->free_events.push_back(p.b);
    } else {
      if (q != hipErrorNotReady) (void)hipGetLastError();
      m->pending[keep++] = p;
    }
  }
  m->pending.resize(keep);
}

using namespace m2s;

extern "C" {

size_t m2s_triangle_count(size_t n_vertices, size_t n_indices, int has_indices, int topology) {
  const size_t n = has_indices ? n_indices : n_vertices;
  if (topology == M2S_TRIANGLE_LIST) return n / 3;       // itertools::tuples drops a trailing partial triple
  return n >= 3 ? n - 2 : 0;                             // tuple_windows
}

int m2s_interleaved_slab(const m2s_grid* grid, int n, int k, uint64_t* x_begin, uint64_t* x_end, uint64_t* x_period) {
  // shard k of n takes the chunks k and n + k of 2n where the grid allows it (m2s_opts.x_period), else its contiguous slab
  uint64_t a = 0, b = 0;
  const uint64_t nx = grid ? grid->cell_count[0] : 0;
  m2s_slab_bounds(nx, n, k, &a, &b);
  *x_begin = a; *x_end = b; *x_period = 0;
  if (!grid || n < 2 || k < 0 || k >= n || nx % (2ull * (uint64_t)n) != 0) return 0;
  const uint64_t C = nx / (2ull * (uint64_t)n);
  uint32_t bl[3];
  choose_brick_shape(grid->cell_size, bl);
  if ((C & (C - 1)) != 0 || C % (4ull << bl[0]) != 0) return 0;   // whole cut-list waves / push pieces per chunk (fill_grid_params checks the same)
  *x_begin = (uint64_t)k * C; *x_end = (uint64_t)(k + 1) * C; *x_period = (uint64_t)n * C;
  return 1;
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
    DeviceState& ds = *kv.second;
    std::lock_guard<std::mutex> lk2(ds.mu);
    if (hipSetDevice(kv.first.first) != hipSuccess) continue;
    (void)hipDeviceSynchronize();
    release_scratch(ds);
    if (ds.spare_plane) (void)hipFree(ds.spare_plane);
    ds.spare_plane = nullptr;
    ds.spare_plane_bytes = 0;
    for (auto e : ds.timing_events) (void)hipEventDestroy(e);
    ds.timing_events.clear();
    if (ds.spare_mesh) (void)hipFree(ds.spare_mesh);
    ds.spare_mesh = nullptr;
    ds.spare_mesh_bytes = 0;
    for (auto& r : ds.ring) {
      if (r) (void)hipHostFree(r);
      r = nullptr;
    }
    ds.ring_bytes = 0;
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
  g_err[0] = 0;
  if (!grid) return fail(M2S_ERR_BAD_ARG, "grid is NULL");
  if (sign_method != M2S_SIGN_RAYCAST && sign_method != M2S_SIGN_NORMAL) return fail(M2S_ERR_BAD_ARG, "bad sign_method %d", sign_method);
  int rc = check_mesh_args(vertices, n_vertices, indices, n_indices, index_bytes, topology);
  if (rc) return rc;
  const size_t n_tris = m2s_triangle_count(n_vertices, n_indices, indices != nullptr, topology);
  GridParams g;
  size_t slab_cells = 0;
  rc = fill_grid_params(grid, opts, &g, &slab_cells);
  if (rc) return rc;
  const uint64_t ny = grid->cell_count[1], nz = grid->cell_count[2], xb = g.xb;
  if (slab_cells == 0) {  // a grid without cells: the reference returns an empty Vec
    if (opts && opts->timings) memset(opts->timings, 0, sizeof(*opts->timings));
    return M2S_OK;
  }
  if (!out) return fail(M2S_ERR_BAD_ARG, "out is NULL");

  HostClock hc;
  CallCtx c;
  DeviceState* st = nullptr;
  rc = resolve_ctx(opts, &c, &st);
  if (rc) return rc;
  hc.lap("runtime + context");
  if (g.chunk_log < 31u && c.mem_kind != M2S_MEM_DEVICE) return fail(M2S_ERR_BAD_ARG, "m2s_opts.x_period needs mem_kind == M2S_MEM_DEVICE");

  size_t need = bvh_workspace_bytes(n_tris) + 4096;
  if (sign_method == M2S_SIGN_RAYCAST) need += sign_workspace_bytes(g, n_tris);
  need += grid_distance_workspace_bytes(g, n_tris);
  if (c.mem_kind == M2S_MEM_HOST)
    need += align_up(n_vertices * 12) + align_up(n_indices * (size_t)(indices ? index_bytes : 0)) + align_up(slab_cells * 4) + 1024;
  rc = ensure_capacity(*st, need);
  if (rc) return rc;
  hc.lap("workspace");
  Arena ws{st->base, st->cap, 0};

  int* d_err = ws.take<int>(16);
  M2S_HIP_CHECK(hipMemsetAsync(d_err, 0, 64, c.stream));   // the first kernel-side call: loads this library's code objects
  StagedMesh sm;
  rc = stage_mesh(ws, c, vertices, n_vertices, indices, n_indices, index_bytes, &sm);
  if (rc) return rc;
  hc.lap("mesh staged");
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
  // Seed passes beside the build: jump flooding needs the triangle centroids only, which exist (in input order) right after
  // the first kernels of the build; its 0.5 ms of small dependent kernels then run on the side stream while the caller's
  // stream sorts, derives the hierarchy and fits the bounds (0.3 ms).  The ids are translated to sorted slots afterwards.
  st->have_raw_seeds = false;
  const uint32_t* plane = nullptr;
  st->early_planes = false;
  // tiny problems (cells x triangles small): all voxels against all triangles, no tree (distance.hip k_brute_split)
  const bool stats = tuning().stats != 0;
  const bool tiny = grid_is_tiny(g, n_tris, c.algorithm, sign_method == M2S_SIGN_RAYCAST) && !stats;
  const bool beside = !tiny && c.algorithm == 0 && !stats && grid_walk_wants_seeds(g, n_tris, c.algorithm);
  const std::function<int(const float4*, const TriRec*, int)> seeds_beside_build = [&](const float4* cen_raw, const TriRec* raw, int phase) -> int {
    if (!side_stream_wanted(c.sync)) return 0;               // no side stream for this call: seeds as part of the walk's preparation
    if (phase == 0) {                                        // the centroid / record kernels are enqueued: mark that point
      if (!st->seeds_done) {
        M2S_HIP_CHECK(hipEventCreateWithFlags(&st->seeds_done, hipEventDisableTiming));
        M2S_HIP_CHECK(hipEventCreateWithFlags(&st->seeds_fork, hipEventDisableTiming));
      }
      M2S_HIP_CHECK(hipEventRecord(st->seeds_fork, c.stream));
    }
    if (phase == 1 && sign_method == M2S_SIGN_RAYCAST && n_tris) {
      // The Raycast sign planes need nothing but the triangles themselves (marking is triangle-parallel and XOR is
      // commutative, so the input order serves as well as the sorted one): a third stream builds them beside the rest of
      // the build and the seed passes.  After the build they were the last thing the walk of a thin slab waited for
      // (8-GPU rank: 0.09 ms of its 2.0 ms step).
      if (!st->side_stream2) { if (const int prc = create_side_stream(&st->side_stream2)) return prc; }
      M2S_HIP_CHECK(hipStreamWaitEvent(st->side_stream2, st->seeds_fork, 0));
      M2S_HIP_CHECK(hipEventRecord(st->ev[5], st->side_stream2));
      DeviceMesh rm{};
      rm.tris = raw;
      rm.n_tris = (uint32_t)n_tris;
      int r2 = build_grid_sign_plane(ws, st->side_stream2, rm, g, &plane, true);
      if (r2) return r2;
      M2S_HIP_CHECK(hipEventRecord(st->ev[2], st->side_stream2));
      st->early_planes = true;
    }
    if (!beside) return 0;
    // A large lattice (512^3: 2 M brick centres, 0.5 ms of passes) must start at once to finish beside the build; a small
    // one (the slab of a multi-GPU rank: 0.1 ms) starts when the sort is enqueued, so that the host's ~100 us of side-stream
    // launches do not leave the caller's stream idle (common.h build_device_mesh).
    const bool big = (uint64_t)host_packet_bricks(g) > 600000u;
    if (phase == 0) {
      if (!big) return 0;
    } else if (big) {
      return 0;                                              // launched at phase 0
    }
    int r = ensure_side_stream(*st);
    if (r) return r;
    M2S_HIP_CHECK(hipStreamWaitEvent(st->side_stream, st->seeds_fork, 0));
    r = launch_grid_seeds(ws, st->side_stream, cen_raw, (uint32_t)n_tris, g, &st->raw_seeds);
    if (r) return r;
    M2S_HIP_CHECK(hipEventRecord(st->seeds_done, st->side_stream));
    st->have_raw_seeds = true;
    return 0;
  };
  rc = build_device_mesh(ws, c.stream, sm.d_verts, n_vertices, sm.d_indices, n_indices, index_bytes, topology, n_tris, d_err, &mesh,
                         &seeds_beside_build, tiny, grid_leaf_max(g, n_tris), (uint64_t)slab_cells);
  if (rc) return rc;
  hc.lap("build enqueued (code objects on a first call)");
  M2S_HIP_CHECK(hipEventRecord(st->ev[1], c.stream));
  st->planes_done = nullptr;
  if (st->early_planes) {
    st->planes_done = st->ev[2];                             // the walk waits for it
  } else if (sign_method == M2S_SIGN_RAYCAST) {
    hipStream_t ss;
    rc = sign_stream_begin(*st, c.stream, c.sync, &ss);
    if (rc) return rc;
    rc = build_grid_sign_plane(ws, ss, mesh, g, &plane, true);
    if (rc) return rc;
    rc = sign_stream_end(*st, c.stream, ss);
    if (rc) return rc;
  } else {
    M2S_HIP_CHECK(hipEventRecord(st->ev[2], c.stream));
  }
  if (c.mem_kind == M2S_MEM_HOST && !stats) {
    uint32_t pieces = 1;
    rc = run_grid_distance_to_host(ws, c, *st, mesh, g, sign_method, plane, d_slab, d_err, out + (size_t)xb * ny * nz, &pieces);
    if (rc) return rc;
    hc.lap("walks + pinned ring + copy into the caller's array");
    rc = finish_call(c, *st, d_err, c.timings, n_tris, slab_cells, sign_method == M2S_SIGN_RAYCAST, false);
    if (c.timings) c.timings->distance_launches = pieces;   // distance_ms then covers the seed passes of every piece too
    hc.done("m2s_generate_grid_sdf (host result)");
    return rc;
  }
  if (c.peers.n && c.peer_mode != M2S_PEER_STORE && !stats) {
    uint32_t pieces = 1;
    rc = c.peer_mode == M2S_PEER_TRAIL ? run_grid_distance_trail(ws, c, *st, mesh, g, sign_method, plane, d_out, d_err)
                                       : run_grid_distance_push(ws, c, *st, mesh, g, sign_method, plane, d_out, d_err, &pieces);
    if (rc) return rc;
    rc = finish_call(c, *st, d_err, c.timings, n_tris, slab_cells, sign_method == M2S_SIGN_RAYCAST, true);
    if (c.timings) c.timings->distance_launches = pieces;
    return rc;
  }
  rc = run_grid_distance(ws, c, *st, mesh, g, sign_method, plane, d_out, d_err);
  if (rc) return rc;
  if (c.mem_kind == M2S_MEM_HOST)
    M2S_HIP_CHECK(hipMemcpyAsync(out + (size_t)xb * ny * nz, d_slab, slab_cells * 4, hipMemcpyDeviceToHost, c.stream));
  rc = finish_call(c, *st, d_err, c.timings, n_tris, slab_cells, sign_method == M2S_SIGN_RAYCAST, true);
  hc.lap("enqueue + wait");
  hc.done("m2s_generate_grid_sdf");
  return rc;
}

int m2s_generate_sdf(const float* vertices, size_t n_vertices, const void* indices, size_t n_indices, int index_bytes,
                     int topology, const float* queries, size_t n_queries, int accel, int sign_method, float* out,
                     size_t* n_out, const m2s_opts* opts) {
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

  HostClock hc;
  CallCtx c;
  DeviceState* st = nullptr;
  rc = resolve_ctx(opts, &c, &st);
  if (rc) return rc;
  hc.lap("runtime + context");

  int mode, sign_src, algorithm;
  select_generic_mode(accel, sign_method, c.algorithm, &mode, &sign_src, &algorithm);

  size_t need = bvh_workspace_bytes(n_tris) + query_workspace_bytes(n_queries) + 4096;
  if (c.mem_kind == M2S_MEM_HOST)
    need += align_up(n_vertices * 12) + align_up(n_indices * (size_t)(indices ? index_bytes : 0)) + align_up(n_queries * 12) + align_up(n_queries * 4) + 1024;
  rc = ensure_capacity(*st, need);
  if (rc) return rc;
  hc.lap("workspace");
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
    rc = staged_h2d(*st, c.stream, reinterpret_cast<char*>(dq), reinterpret_cast<const char*>(queries), n_queries * 12);
    if (rc) return rc;
    d_q = dq;
    hc.lap("queries to the device (pinned ring on a first call)");
  }
  M2S_HIP_CHECK(hipEventRecord(st->ev[0], c.stream));
  if (query_is_tiny(n_queries, n_tris, algorithm, sign_src) && !tuning().stats) {
    // a small query set (the crate's documented use): the triangle records and nothing else — no keys, no sort, no tree (k_brute_split_q)
    DeviceMesh mesh;
    rc = build_device_mesh(ws, c.stream, sm.d_verts, n_vertices, sm.d_indices, n_indices, index_bytes, topology, n_tris, d_err, &mesh, nullptr, true);
    if (rc) return rc;
    M2S_HIP_CHECK(hipEventRecord(st->ev[1], c.stream));
    M2S_HIP_CHECK(hipEventRecord(st->ev[2], c.stream));
    rc = launch_query_brute_split(ws, c.stream, mesh, d_q, n_queries, mode, sign_src, d_out, d_err);
    if (rc) return rc;
    M2S_HIP_CHECK(hipEventRecord(st->ev[3], c.stream));
    if (c.mem_kind == M2S_MEM_HOST) {
      rc = staged_d2h(*st, c.stream, reinterpret_cast<char*>(out), reinterpret_cast<const char*>(d_out), n_queries * 4);
      if (rc) return rc;
    }
    if (n_out) *n_out = n_queries;
    rc = finish_call(c, *st, d_err, c.timings, n_tris, n_queries, false);
    hc.lap("records + all pairs + result");
    hc.done("m2s_generate_sdf (small query set)");
    return rc;
  }
  // What the queries need before the walk — bounding box, Morton keys, sort, packet table, gather (distance.hip prepare_query_walk) — does
  // not need the tree: a synchronous call runs it on the side stream beside the build
  QueryPlan qplan;
  const bool beside = algorithm == 0 && n_tris != 0 && n_queries >= 32768 && side_stream_wanted(c.sync) && !tuning().stats;
  if (beside) {
    rc = ensure_side_stream(*st);
    if (rc) return rc;
    if (!st->qprep_done) M2S_HIP_CHECK(hipEventCreateWithFlags(&st->qprep_done, hipEventDisableTiming));
    M2S_HIP_CHECK(hipEventRecord(st->fork_ev, c.stream));                 // the queries are on the device
    M2S_HIP_CHECK(hipStreamWaitEvent(st->side_stream, st->fork_ev, 0));
    if (!st->qlat_done) M2S_HIP_CHECK(hipEventCreateWithFlags(&st->qlat_done, hipEventDisableTiming));
    rc = prepare_query_walk(ws, st->side_stream, d_q, n_queries, n_tris, sign_src, algorithm, &qplan, st->qlat_done);
    if (rc) return rc;
    M2S_HIP_CHECK(hipEventRecord(st->qprep_done, st->side_stream));
  }
  // ... and the seed lattice needs the centroids only: flooded on the third stream from the input-order centroids while the tree is
  // built (the ids are translated to sorted slots behind the build, as on the grid path)
  QuerySeeds qseeds;
  bool seeds_beside = false;
  const std::function<int(const float4*, const TriRec*, int)> seeds_beside_build = [&](const float4* cen_raw, const TriRec*, int phase) -> int {
    if (!beside || !qplan.seeds) return 0;
    if (phase == 0) {                                                     // the centroid kernel is enqueued: mark that point
      if (!st->seeds_done) {
        M2S_HIP_CHECK(hipEventCreateWithFlags(&st->seeds_done, hipEventDisableTiming));
        M2S_HIP_CHECK(hipEventCreateWithFlags(&st->seeds_fork, hipEventDisableTiming));
      }
      M2S_HIP_CHECK(hipEventRecord(st->seeds_fork, c.stream));
      return 0;
    }
    // phase 1: the caller's stream holds the keys, the sort and the hierarchy by now — the host's time for these twelve launches
    if (!st->side_stream2) { if (const int prc = create_side_stream(&st->side_stream2)) return prc; }
    M2S_HIP_CHECK(hipStreamWaitEvent(st->side_stream2, st->seeds_fork, 0));
    M2S_HIP_CHECK(hipStreamWaitEvent(st->side_stream2, st->qlat_done, 0));
    const int r = launch_query_seeds(ws, st->side_stream2, cen_raw, (uint32_t)n_tris, qplan, true, &qseeds);
    if (r) return r;
    M2S_HIP_CHECK(hipEventRecord(st->seeds_done, st->side_stream2));
    seeds_beside = true;
    return 0;
  };
  DeviceMesh mesh;
  rc = build_device_mesh(ws, c.stream, sm.d_verts, n_vertices, sm.d_indices, n_indices, index_bytes, topology, n_tris, d_err, &mesh,
                         &seeds_beside_build, false, query_leaf_max(n_queries, n_tris, sign_src));
  if (rc) return rc;
  hc.lap("build enqueued (code objects on a first call)");
  M2S_HIP_CHECK(hipEventRecord(st->ev[1], c.stream));
  M2S_HIP_CHECK(hipEventRecord(st->ev[2], c.stream));
  unsigned long long* d_stats = nullptr;
  rc = stats_begin(ws, c.stream, &mesh, &d_stats);
  if (rc) return rc;
  if (beside) {
    M2S_HIP_CHECK(hipStreamWaitEvent(c.stream, st->qprep_done, 0));
    if (seeds_beside) M2S_HIP_CHECK(hipStreamWaitEvent(c.stream, st->seeds_done, 0));
  } else {
    rc = prepare_query_walk(ws, c.stream, d_q, n_queries, n_tris, sign_src, algorithm, &qplan);
    if (rc) return rc;
  }
  rc = launch_query_walk(ws, c.stream, mesh, d_q, qplan, mode, sign_src, algorithm, d_out, d_err, seeds_beside ? &qseeds : nullptr);
  if (rc) return rc;
  M2S_HIP_CHECK(hipEventRecord(st->ev[3], c.stream));
  rc = stats_end(c.stream, d_stats);
  if (rc) return rc;
  if (c.mem_kind == M2S_MEM_HOST) {
    rc = staged_d2h(*st, c.stream, reinterpret_cast<char*>(out), reinterpret_cast<const char*>(d_out), n_queries * 4);
    if (rc) return rc;
  }
  if (n_out) *n_out = n_queries;
  rc = finish_call(c, *st, d_err, c.timings, n_tris, n_queries, false);
  hc.lap("sort + walk + result");
  hc.done("m2s_generate_sdf");
  return rc;
}

// Pays the one-off costs of a process's first call now (include/m2s.h).
int m2s_warmup(int device, size_t workspace_bytes, size_t host_ring_bytes) {
  g_err[0] = 0;
  m2s_opts o{};
  o.struct_size = sizeof(m2s_opts);
  o.device = device;
  o.mem_kind = M2S_MEM_DEVICE;
  o.synchronous = 1;
  CallCtx c;
  DeviceState* st = nullptr;
  int rc = resolve_ctx(&o, &c, &st);                             // runtime, device, the context's events and stream
  if (rc) return rc;
  if (workspace_bytes) { rc = ensure_capacity(*st, workspace_bytes); if (rc) return rc; }
  if (host_ring_bytes) { rc = ensure_ring(*st, std::min<size_t>(host_ring_bytes, (size_t)64 << 20)); if (rc) return rc; }
  rc = ensure_side_stream(*st);
  if (rc) return rc;
  if (!st->side_stream2) { rc = create_side_stream(&st->side_stream2); if (rc) return rc; }
  if (!st->copy_stream) M2S_HIP_CHECK(hipStreamCreateWithFlags(&st->copy_stream, hipStreamNonBlocking));
  if (st->warmed_cap == 0 || st->cap > st->warmed_cap) {        // (a repeated call with nothing new to touch does none of this again)
    // the first copy between PAGEABLE host memory and the device makes the runtime set up its staging path (measured: 15 ms of a first
    // call's mesh upload): a few bytes each way through the workspace
    constexpr size_t probe = (size_t)4 << 20;                     // large enough to take the runtime's chunked staging path
    rc = ensure_capacity(*st, probe);
    if (rc) return rc;
    std::vector<char> tmp(probe, 0);
    M2S_HIP_CHECK(hipMemcpyAsync(st->base, tmp.data(), probe, hipMemcpyHostToDevice, c.stream));
    M2S_HIP_CHECK(hipMemcpyAsync(tmp.data(), st->base, probe, hipMemcpyDeviceToHost, c.stream));
    // ... and the first touch of freshly allocated device memory maps its pages (17 ms for the 0.7 GB of a 512^3 call): touch it all now
    M2S_HIP_CHECK(hipMemsetAsync(st->base, 0, workspace_bytes ? st->cap : 256, c.stream));
    M2S_HIP_CHECK(hipStreamSynchronize(c.stream));
    st->warmed_cap = st->cap;
  }
  warm_bvh(c.stream);                                            // one empty kernel per translation unit: the runtime loads a code object
  warm_sign(c.stream);                                           // (all the unit's kernels) at the first launch out of it
  warm_distance(c.stream);
  warm_serde(c.stream);
  warm_client(c.stream);
  warm_sortlib(c.stream);                                        // (the rocPRIM sorts of large meshes and of the query path: units of their own,
  warm_sortlib_query(c.stream);                                  // which a grid call over a mesh of <= 229 376 triangles never loads)
  M2S_HIP_CHECK(hipGetLastError());
  M2S_HIP_CHECK(hipStreamSynchronize(c.stream));
  return M2S_OK;
}


// ---- persistent mesh ---------------------------------------------------------------------------
int m2s_mesh_create(const float* vertices, size_t n_vertices, const void* indices, size_t n_indices, int index_bytes,
                    int topology, const m2s_opts* opts, m2s_mesh** out_mesh) {
  g_err[0] = 0;
  if (!out_mesh) return fail(M2S_ERR_BAD_ARG, "out_mesh is NULL");
  *out_mesh = nullptr;
  int rc = check_mesh_args(vertices, n_vertices, indices, n_indices, index_bytes, topology);
  if (rc) return rc;
  const size_t n_tris = m2s_triangle_count(n_vertices, n_indices, indices != nullptr, topology);
  CallCtx c;
  DeviceState* st = nullptr;
  rc = resolve_ctx(opts, &c, &st);
  if (rc) return rc;
  m2s_mesh* m = new m2s_mesh();
  m->device = c.device;
  m->lane = c.lane;
  m->n_tris = n_tris;
  m->mem_bytes = bvh_workspace_bytes(n_tris) + 8192;
  if (c.mem_kind == M2S_MEM_HOST) m->mem_bytes += align_up(n_vertices * 12) + align_up(n_indices * (size_t)(indices ? index_bytes : 0)) + 1024;
  if (st->spare_mesh && st->spare_mesh_bytes >= m->mem_bytes) {   // recycle (steady-state: no hipMalloc per call)
    m->mem = st->spare_mesh;
    m->mem_bytes = st->spare_mesh_bytes;
    st->spare_mesh = nullptr;
    st->spare_mesh_bytes = 0;
  } else if (hipMalloc((void**)&m->mem, m->mem_bytes) != hipSuccess) {
    const size_t want = m->mem_bytes;
    delete m;
    return fail(M2S_ERR_HIP, "hipMalloc of %zu bytes for the mesh failed", want);
  }
  Arena ws{m->mem, m->mem_bytes, 0};
  int* d_err = ws.take<int>(32);
  m->d_err_async = d_err + 16;
  auto bail = [&](int code) { (void)hipFree(m->mem); delete m; return code; };
  if (hipMemsetAsync(d_err, 0, 128, c.stream) != hipSuccess) return bail(fail(M2S_ERR_HIP, "memset failed"));
  StagedMesh sm;
  rc = stage_mesh(ws, c, vertices, n_vertices, indices, n_indices, index_bytes, &sm);
  if (rc) return bail(rc);
  (void)hipEventRecord(st->ev[0], c.stream);
  rc = build_device_mesh(ws, c.stream, sm.d_verts, n_vertices, sm.d_indices, n_indices, index_bytes, topology, n_tris, d_err, &m->dm);
  if (rc) return bail(rc);
  (void)hipEventRecord(st->ev[1], c.stream);
  if (hipMemcpyAsync(st->h_err, d_err, sizeof(int), hipMemcpyDeviceToHost, c.stream) != hipSuccess ||
      hipStreamSynchronize(c.stream) != hipSuccess)
    return bail(fail(M2S_ERR_HIP, "mesh build failed: %s", hipGetErrorString(hipGetLastError())));
  (void)hipEventElapsedTime(&m->build_ms, st->ev[0], st->ev[1]);
  if (*st->h_err & ERRF_INDEX_OOB) return bail(fail(M2S_ERR_BAD_ARG, "vertex index out of range (the reference panics indexing `vertices`)"));
  if (*st->h_err & ERRF_BUILD_TIMEOUT) return bail(fail(M2S_ERR_HIP, "the LBVH build gave up waiting for a neighbouring tile of its sort"));
  *out_mesh = m;
  return M2S_OK;
}

void m2s_mesh_destroy(m2s_mesh* m) {
  if (!m) return;
  DeviceState& ds = *find_state(m->device, m->lane);
  std::unique_lock<std::mutex> lk(ds.mu);
  if (hipSetDevice(m->device) == hipSuccess) {
    (void)hipDeviceSynchronize();
    if (m->mem && !ds.spare_mesh) { ds.spare_mesh = m->mem; ds.spare_mesh_bytes = m->mem_bytes; }
    else if (m->mem) (void)hipFree(m->mem);
    if (m->plane_mem && (!ds.spare_plane || ds.spare_plane_bytes < m->plane_bytes)) {
      if (ds.spare_plane) (void)hipFree(ds.spare_plane);
      ds.spare_plane = m->plane_mem;
      ds.spare_plane_bytes = m->plane_bytes;
    } else if (m->plane_mem) (void)hipFree(m->plane_mem);
    if (m->plane_ready) (void)hipEventDestroy(m->plane_ready);
    for (auto& p : m->pending) { m->free_events.push_back(p.a); m->free_events.push_back(p.b); }
    for (auto e : m->free_events) {
      if (ds.timing_events.size() < 64) ds.timing_events.push_back(e);
      else (void)hipEventDestroy(e);
    }
  }
  lk.unlock();
  delete m;
}

size_t m2s_mesh_triangle_count(const m2s_mesh* m) { return m ? m->n_tris : 0; }

static bool same_grid(const m2s_grid& a, const m2s_grid& b) { return memcmp(&a, &b, sizeof(m2s_grid)) == 0; }

int m2s_mesh_generate_grid_sdf(m2s_mesh* m, const m2s_grid* grid, int sign_method, float* out, const m2s_opts* opts) {
  g_err[0] = 0;
  if (!m) return fail(M2S_ERR_BAD_ARG, "mesh is NULL");
  std::lock_guard<std::mutex> mlk(m->mu);
  if (!grid) return fail(M2S_ERR_BAD_ARG, "grid is NULL");
  if (sign_method != M2S_SIGN_RAYCAST && sign_method != M2S_SIGN_NORMAL) return fail(M2S_ERR_BAD_ARG, "bad sign_method %d", sign_method);
  GridParams g;
  size_t slab_cells = 0;
  int rc = fill_grid_params(grid, opts, &g, &slab_cells);
  if (rc) return rc;
  if (slab_cells == 0) {
    if (opts && opts->timings) memset(opts->timings, 0, sizeof(*opts->timings));
    return M2S_OK;
  }
  if (!out) return fail(M2S_ERR_BAD_ARG, "out is NULL");
  m2s_opts o{};
  if (opts) memcpy(&o, opts, (opts->struct_size >= sizeof(m2s_opts)) ? sizeof(m2s_opts) : (size_t)M2S_OPTS_V1_SIZE);
  else o.device = -1;
  if (o.device < 0) o.device = m->device;
  if (o.device != m->device) return fail(M2S_ERR_BAD_ARG, "mesh lives on device %d, call asked for %d", m->device, o.device);
  if (!opts) o.synchronous = 1;
  CallCtx c;
  DeviceState* st = nullptr;
  rc = resolve_ctx(&o, &c, &st);
  if (rc) return rc;
  if (g.chunk_log < 31u && c.mem_kind != M2S_MEM_DEVICE) return fail(M2S_ERR_BAD_ARG, "m2s_opts.x_period needs mem_kind == M2S_MEM_DEVICE");
  const uint64_t ny = grid->cell_count[1], nz = grid->cell_count[2], xb = g.xb;

  size_t need = grid_distance_workspace_bytes(g, m->n_tris) + 8192;
  if (c.mem_kind == M2S_MEM_HOST) need += align_up(slab_cells * 4) + 1024;
  rc = ensure_capacity(*st, need);
  if (rc) return rc;
  Arena ws{st->base, st->cap, 0};
  int* d_err = ws.take<int>(16);
  if (c.sync) M2S_HIP_CHECK(hipMemsetAsync(d_err, 0, 64, c.stream));
  else d_err = m->d_err_async;   // asynchronous calls report through the mesh: m2s_mesh_drain_timings reads and clears it
  float* d_out = out;
  float* d_slab = nullptr;
  if (c.mem_kind == M2S_MEM_HOST) {
    d_slab = ws.take<float>(slab_cells);
    if (!d_slab) return fail(M2S_ERR_HIP, "internal: workspace");
    d_out = d_slab;
    g.out_off = (uint64_t)xb * ny * nz;
  }
  if (!c.sync) reap_pending(m, false);
  M2S_HIP_CHECK(hipEventRecord(st->ev[0], c.stream));
  M2S_HIP_CHECK(hipEventRecord(st->ev[1], c.stream));
  st->planes_done = nullptr;
  st->have_raw_seeds = false;
  rc = remark_leaves(m, c, grid_leaf_max(g, m->n_tris));   // leaves of 4 - 16 triangles where a brick meets several
  if (rc) return rc;
  bool built_planes = false;
  const uint32_t* plane = nullptr;
  if (sign_method == M2S_SIGN_RAYCAST) {
    if (!m->plane_valid || !same_grid(m->plane_grid, *grid)) {
      // readers of the old planes may still be in flight: asynchronous walks (any stream), or walks on another stream
      if (m->multi_stream || m->async_readers) { M2S_HIP_CHECK(hipDeviceSynchronize()); m->multi_stream = false; m->async_readers = false; }
      const size_t bytes = sign_workspace_bytes(g, m->n_tris);
      if (bytes > m->plane_bytes) {
        if (m->plane_mem) { M2S_HIP_CHECK(hipDeviceSynchronize()); M2S_HIP_CHECK(hipFree(m->plane_mem)); m->plane_mem = nullptr; m->plane_bytes = 0; }
        if (st->spare_plane && st->spare_plane_bytes >= bytes) {   // recycled from the last destroyed mesh (idle since then)
          m->plane_mem = st->spare_plane;
          m->plane_bytes = st->spare_plane_bytes;
          st->spare_plane = nullptr;
          st->spare_plane_bytes = 0;
        } else {
          M2S_HIP_CHECK(hipMalloc((void**)&m->plane_mem, bytes));
          m->plane_bytes = bytes;
        }
      }
      Arena pw{m->plane_mem, m->plane_bytes, 0};
      hipStream_t ss;
      rc = sign_stream_begin(*st, c.stream, c.sync, &ss);
      if (rc) return rc;
      rc = build_grid_sign_plane(pw, ss, m->dm, g, &m->plane, false);   // kept per grid, whatever slab asks first
      if (rc) return rc;
      m->plane_grid = *grid;
      m->plane_valid = true;
      built_planes = true;
      if (!m->plane_ready) M2S_HIP_CHECK(hipEventCreateWithFlags(&m->plane_ready, hipEventDisableTiming));
      M2S_HIP_CHECK(hipEventRecord(m->plane_ready, ss));
      m->plane_stream = ss;
      rc = sign_stream_end(*st, c.stream, ss);   // this call's walk waits for ev[2]
      if (rc) return rc;
    } else if (c.stream != m->plane_stream) {   // built (perhaps still being built) on another stream
      M2S_HIP_CHECK(hipStreamWaitEvent(c.stream, m->plane_ready, 0));
      m->multi_stream = true;
    }
    plane = m->plane;
  }
  if (!built_planes) M2S_HIP_CHECK(hipEventRecord(st->ev[2], c.stream));
  const bool stats = tuning().stats != 0;
  if (c.mem_kind == M2S_MEM_HOST && !stats) {   // host result: x-pieces stream out while the next computes
    uint32_t pieces = 1;
    rc = run_grid_distance_to_host(ws, c, *st, m->dm, g, sign_method, plane, d_slab, d_err, out + (size_t)xb * ny * nz, &pieces);
    if (rc) return rc;
    rc = finish_call(c, *st, d_err, c.timings, m->n_tris, slab_cells, built_planes, false);
    if (c.timings) c.timings->distance_launches = pieces;
    return rc;
  }
  uint32_t walk_launches = 1;
  if (c.peers.n && c.peer_mode == M2S_PEER_TRAIL && !stats)
    rc = run_grid_distance_trail(ws, c, *st, m->dm, g, sign_method, plane, d_out, d_err);
  else if (c.peers.n && c.peer_mode == M2S_PEER_PUSH && !stats)
    rc = run_grid_distance_push(ws, c, *st, m->dm, g, sign_method, plane, d_out, d_err, &walk_launches);
  else
    rc = run_grid_distance(ws, c, *st, m->dm, g, sign_method, plane, d_out, d_err);
  if (rc) return rc;
  if (c.mem_kind == M2S_MEM_HOST) {
    rc = staged_d2h(*st, c.stream, reinterpret_cast<char*>(out + (size_t)xb * ny * nz), reinterpret_cast<const char*>(d_slab), slab_cells * 4);
    if (rc) return rc;
  }
  if (!c.sync) {
    if (plane) m->async_readers = true;
    // Asynchronous call: no host sync now.  Keep the event pair around the dominant launch(es) so that their
    // duration can be read later (m2s_mesh_drain_timings): ev[4] was recorded just before the first walk, ev[3]
    // right after the last; both move into `pending` and the context gets fresh events for the next call.
    hipEvent_t fresh[2];
    for (hipEvent_t& e : fresh) {
      if (!m->free_events.empty()) { e = m->free_events.back(); m->free_events.pop_back(); }
      else if (!st->timing_events.empty()) { e = st->timing_events.back(); st->timing_events.pop_back(); }
      else M2S_HIP_CHECK(hipEventCreate(&e));
    }
    m->pending.push_back({st->ev[4], st->ev[3], (uint64_t)slab_cells, walk_launches});
    st->ev[4] = fresh[0];
    st->ev[3] = fresh[1];
    return M2S_OK;
  }
  rc = finish_call(c, *st, d_err, c.timings, m->n_tris, slab_cells, built_planes, true);
  if (c.timings) c.timings->distance_launches = walk_launches;
  return rc;
}

int m2s_mesh_drain_timings(m2s_mesh* m, m2s_timings* t) {
  if (!m || !t) return fail(M2S_ERR_BAD_ARG, "NULL argument");
  std::lock_guard<std::mutex> mlk(m->mu);
  memset(t, 0, sizeof(*t));
  t->accel_build_ms = m->build_ms;
  t->n_triangles = m->n_tris;
  M2S_HIP_CHECK(hipSetDevice(m->device));
  reap_pending(m, true);
  if (!m->pending.empty()) return fail(M2S_ERR_HIP, "an asynchronous call failed on the device: %s", hipGetErrorString(hipGetLastError()));
  t->distance_ms = (float)m->acc_ms;
  t->n_units = m->acc_units;
  t->distance_launches = m->acc_launches;
  m->acc_ms = 0.0;
  m->acc_units = 0;
  m->acc_launches = 0;
  m->async_readers = false;   // every asynchronous walk has finished
  m->tree_last_async = false;
  m->tree_async_other = false;
  // deferred error report of the asynchronous calls since the last drain (the reference panics: lib.rs:257)
  int e = 0;
  M2S_HIP_CHECK(hipMemcpy(&e, m->d_err_async, sizeof(int), hipMemcpyDeviceToHost));
  if (e) M2S_HIP_CHECK(hipMemset(m->d_err_async, 0, sizeof(int)));
  if (e & ERRF_NAN) return fail(M2S_ERR_NAN, "NaN distance (lib.rs:257) in an asynchronous call since the last drain");
  return M2S_OK;
}

int m2s_mesh_generate_sdf(m2s_mesh* m, const float* queries, size_t n_queries, int accel, int sign_method, float* out,
                          size_t* n_out, const m2s_opts* opts) {
  g_err[0] = 0;
  if (n_out) *n_out = 0;
  if (!m) return fail(M2S_ERR_BAD_ARG, "mesh is NULL");
  std::lock_guard<std::mutex> mlk(m->mu);
  if (accel < M2S_ACCEL_NONE || accel > M2S_ACCEL_RTREE_BVH) return fail(M2S_ERR_BAD_ARG, "bad accel %d", accel);
  if (sign_method != M2S_SIGN_RAYCAST && sign_method != M2S_SIGN_NORMAL) return fail(M2S_ERR_BAD_ARG, "bad sign_method %d", sign_method);
  if (n_queries && !queries) return fail(M2S_ERR_BAD_ARG, "queries is NULL");
  if (n_queries >= 0xffffffc0ull) return fail(M2S_ERR_BAD_ARG, "too many queries for one call");
  if (m->n_tris == 0 && accel == M2S_ACCEL_RTREE_BVH) return M2S_OK;
  if (m->n_tris == 0 && accel == M2S_ACCEL_RTREE && n_queries) return fail(M2S_ERR_EMPTY_MESH, "Rtree on a mesh without triangles (rtree.rs:117 unwrap on None)");
  if (n_queries == 0) return M2S_OK;
  if (!out) return fail(M2S_ERR_BAD_ARG, "out is NULL");
  m2s_opts o{};
  if (opts) memcpy(&o, opts, (opts->struct_size >= sizeof(m2s_opts)) ? sizeof(m2s_opts) : (size_t)M2S_OPTS_V1_SIZE);
  else o.device = -1;
  if (o.device < 0) o.device = m->device;
  if (o.device != m->device) return fail(M2S_ERR_BAD_ARG, "mesh lives on device %d, call asked for %d", m->device, o.device);
  if (!opts) o.synchronous = 1;
  CallCtx c;
  DeviceState* st = nullptr;
  int rc = resolve_ctx(&o, &c, &st);
  if (rc) return rc;
  int mode, sign_src, algorithm;
  select_generic_mode(accel, sign_method, c.algorithm, &mode, &sign_src, &algorithm);
  size_t need = query_workspace_bytes(n_queries) + 8192;
  if (c.mem_kind == M2S_MEM_HOST) need += align_up(n_queries * 12) + align_up(n_queries * 4) + 1024;
  rc = ensure_capacity(*st, need);
  if (rc) return rc;
  Arena ws{st->base, st->cap, 0};
  int* d_err = ws.take<int>(16);
  M2S_HIP_CHECK(hipMemsetAsync(d_err, 0, 64, c.stream));
  const float* d_q = queries;
  float* d_out = out;
  if (c.mem_kind == M2S_MEM_HOST) {
    float* dq = ws.take<float>(n_queries * 3);
    d_out = ws.take<float>(n_queries);
    if (!dq || !d_out) return fail(M2S_ERR_HIP, "internal: workspace");
    rc = staged_h2d(*st, c.stream, reinterpret_cast<char*>(dq), reinterpret_cast<const char*>(queries), n_queries * 12);
    if (rc) return rc;
    d_q = dq;
  }
  M2S_HIP_CHECK(hipEventRecord(st->ev[0], c.stream));
  M2S_HIP_CHECK(hipEventRecord(st->ev[1], c.stream));
  M2S_HIP_CHECK(hipEventRecord(st->ev[2], c.stream));
  rc = remark_leaves(m, c, query_is_tiny(n_queries, m->n_tris, algorithm, sign_src) ? m->dm.leaf_max : query_leaf_max(n_queries, m->n_tris, sign_src));
  if (rc) return rc;
  rc = launch_query_distance(ws, c.stream, m->dm, d_q, n_queries, mode, sign_src, algorithm, d_out, d_err);
  if (rc) return rc;
  M2S_HIP_CHECK(hipEventRecord(st->ev[3], c.stream));
  if (c.mem_kind == M2S_MEM_HOST) {
    rc = staged_d2h(*st, c.stream, reinterpret_cast<char*>(out), reinterpret_cast<const char*>(d_out), n_queries * 4);
    if (rc) return rc;
  }
  if (n_out) *n_out = n_queries;
  return finish_call(c, *st, d_err, c.timings, m->n_tris, n_queries, false);
}

// Peer-write bandwidth probe (include/m2s.h): the copy kernel of M2S_PEER_PUSH, timed with HIP events on a stream of its own.
int m2s_peer_bandwidth(const float* src, float* const* peers, uint32_t n_peers, size_t n_cells, int device, float* gbps_each, float* gbps_all) {
  g_err[0] = 0;
  if (!src || (n_peers && !peers) || n_peers > M2S_MAX_PEERS) return fail(M2S_ERR_BAD_ARG, "m2s_peer_bandwidth: bad arguments");
  if (n_cells == 0 || n_peers == 0) { if (gbps_all) *gbps_all = 0.0f; return M2S_OK; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(M2S_ERR_HIP, "no HIP device available");
  if (device < 0) M2S_HIP_CHECK(hipGetDevice(&device));
  if (device >= ndev) return fail(M2S_ERR_BAD_ARG, "device %d out of range (%d devices)", device, ndev);
  int prev_device = device;
  (void)hipGetDevice(&prev_device);
  // everything created below is released, and the caller's current device restored, on every way out
  struct Guard {
    hipStream_t st = nullptr;
    hipEvent_t a = nullptr, b = nullptr;
    int restore;
    ~Guard() {
      if (a) (void)hipEventDestroy(a);
      if (b) (void)hipEventDestroy(b);
      if (st) (void)hipStreamDestroy(st);
      (void)hipSetDevice(restore);
    }
  } guard;
  guard.restore = prev_device;
  M2S_HIP_CHECK(hipSetDevice(device));
  M2S_HIP_CHECK(hipStreamCreateWithFlags(&guard.st, hipStreamNonBlocking));
  M2S_HIP_CHECK(hipEventCreate(&guard.a));
  M2S_HIP_CHECK(hipEventCreate(&guard.b));
  const hipStream_t st = guard.st;
  const hipEvent_t a = guard.a, b = guard.b;
  int rc = M2S_OK;
  auto timed = [&](const PeerOut& po, float* gbps) -> int {
    float best = 0.0f;
    for (int rep = 0; rep < 4; ++rep) {                           // the first repetition maps pages and warms the link up
      M2S_HIP_CHECK(hipEventRecord(a, st));
      const int r = launch_push_cells(st, src, po, 0, n_cells);
      if (r) return r;
      M2S_HIP_CHECK(hipEventRecord(b, st));
      M2S_HIP_CHECK(hipStreamSynchronize(st));
      float ms = 0.0f;
      M2S_HIP_CHECK(hipEventElapsedTime(&ms, a, b));
      if (rep && ms > 0.0f) best = std::max(best, (float)((double)n_cells * 4.0 * po.n / (ms * 1e-3) / 1e9));
    }
    *gbps = best;
    return 0;
  };
  for (uint32_t k = 0; k < n_peers && !rc; ++k) {
    if (!peers[k]) { rc = fail(M2S_ERR_BAD_ARG, "m2s_peer_bandwidth: peers[%u] is NULL", k); break; }
    PeerOut one{};
    one.n = 1;
    one.p[0] = peers[k];
    float g = 0.0f;
    rc = timed(one, &g);
    if (gbps_each) gbps_each[k] = g;
  }
  if (!rc && gbps_all) {
    PeerOut all{};
    all.n = n_peers;
    for (uint32_t k = 0; k < n_peers; ++k) all.p[k] = peers[k];
    rc = timed(all, gbps_all);
  }
  return rc;
}

// Test hook (not part of include/m2s.h): FNV-1a digests of the resident arrays of a mesh — triangle records, pre-test planes,
// box nodes, oriented bounds, centroids, slot table, scene words.  tests/test_gpu_build.py compares two builds of one mesh
// (M2S_BUILD=0 / 1) with it: the lean build must leave the same tree, byte for byte.
int m2s_tuning_set(const char* name, const char* value) {
  clear_error();
  if (tuning_set(name, value) != 0) return fail(M2S_ERR_BAD_ARG, "m2s_tuning_set: unknown knob or unparsable value: %s=%s", name ? name : "(null)", value ? value : "(default)");
  return M2S_OK;
}
int m2s_tuning_describe(char* buffer, int capacity) { return tuning_describe(buffer, capacity); }

int m2s_debug_mesh_digest(m2s_mesh* m, uint64_t out[8]) {
  g_err[0] = 0;
  if (!m || !out) return fail(M2S_ERR_BAD_ARG, "NULL argument");
  std::lock_guard<std::mutex> mlk(m->mu);
  M2S_HIP_CHECK(hipSetDevice(m->device));
  M2S_HIP_CHECK(hipDeviceSynchronize());
  const size_t n = m->n_tris, nn = n ? 2 * n - 1 : 0;
  const void* ptr[7] = {m->dm.tris, m->dm.planes, m->dm.nodes, m->dm.ext, m->dm.cen, m->dm.slot_of, m->dm.scene};
  const size_t bytes[7] = {n * sizeof(TriRec), n * sizeof(TriPlanes), nn * sizeof(NodeRec), nn * sizeof(NodeExt), n * 16, n * 4, n ? 32u : 0u};
  std::vector<unsigned char> h;
  for (int k = 0; k < 7; ++k) {
    uint64_t f = 1469598103934665603ull;
    if (bytes[k]) {
      h.resize(bytes[k]);
      M2S_HIP_CHECK(hipMemcpy(h.data(), ptr[k], bytes[k], hipMemcpyDeviceToHost));
      for (size_t i = 0; i < bytes[k]; ++i) { f ^= h[i]; f *= 1099511628211ull; }
    }
    out[k] = f;
  }
  out[7] = n;
  return M2S_OK;
}

// Test hook (not part of include/m2s.h): the cut-list word of one range and what the walk decodes from it (distance.hip CutList) —
// host arithmetic only, no device needed.  tests/test_capi_cpu.py checks the superset property for every tree size.
int m2s_debug_cut_code(uint32_t n_nodes, uint32_t start, uint32_t len, uint32_t out[3]) {
  g_err[0] = 0;
  if (!out || n_nodes == 0 || n_nodes > (1u << 26) || len == 0 || start >= n_nodes || len > n_nodes - start) return fail(M2S_ERR_BAD_ARG, "bad range");
  cut_word_roundtrip(n_nodes, start, len, &out[0], &out[1], &out[2]);
  return M2S_OK;
}

// Test hook (not part of include/m2s.h): the leaf size a call would ask of its tree — out[0] for a grid call over `grid` (grid_leaf_max: by
// triangles per 4^3-voxel brick of the whole grid), out[1] for a query call of n_queries (query_leaf_max: 2 under the lane walk, else by queries per
// triangle), out[2] = 1 if that query call takes the lane walk.  Host arithmetic only; tests/test_capi_cpu.py pins the rules DESIGN.md states.
int m2s_debug_leaf_sizes(const m2s_grid* grid, size_t n_tris, size_t n_queries, uint32_t out[3]) {
  g_err[0] = 0;
  if (!grid || !out) return fail(M2S_ERR_BAD_ARG, "NULL argument");
  GridParams g;
  size_t slab_cells = 0;
  const int rc = fill_grid_params(grid, nullptr, &g, &slab_cells);
  if (rc) return rc;
  out[0] = grid_leaf_max(g, n_tris);
  out[1] = query_leaf_max(n_queries, n_tris, SIGN_RAYS3);
  out[2] = query_walk_is_lane(n_queries, n_tris, SIGN_RAYS3) ? 1u : 0u;
  return M2S_OK;
}

}  // extern "C"
