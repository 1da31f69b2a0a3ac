// multi.hip — generate_grid_sdf over several GPUs of one node from one process (include/m2s.h:
// m2s_generate_grid_sdf_multi), and the cross-process helpers (m2s_shared_alloc, m2s_ipc_*).
//
// every voxel depends only on the replicated mesh, so the grid shards into contiguous x-slabs with
// no data-path exchange; what remains is delivering the slabs.  Host results: every device streams its slab into the
// caller's array over its own PCIe link.  Device-resident results: every device writes its slab into all peers'
// buffers over xGMI itself (m2s_opts.peer_out; the links are point-to-point, so seven peers are seven links in
// parallel), or — where peer access is not available — one in-place ncclAllGather from librccl.
//
// One host thread per shard: each calls the ordinary single-device entry point with its (device, lane) context,
// x-slab and peer list, so the multi-device path adds no second implementation of anything.
#include <dlfcn.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <rccl/rccl.h>   // types and prototypes only: the library itself is loaded on first use (dlopen)

#include "../../include/m2s.h"
#include "capi_internal.h"
#include "common.h"

namespace m2s {
namespace {

// ---- librccl, loaded on demand -------------------------------------------------------------------
// The product must load (and serve single-GPU callers) on machines without RCCL, and linking it would pull a
// several-hundred-MB library into every process that only wants one GPU.
struct Rccl {
  void* handle = nullptr;
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclBroadcast) Broadcast = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  std::string why;
  bool ok = false;
};

Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      r.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (r.handle) break;
    }
    if (!r.handle) { r.why = std::string("dlopen(librccl.so.1) failed: ") + (dlerror() ? dlerror() : "?"); return; }
    auto sym = [&](const char* n) { void* p = dlsym(r.handle, n); if (!p) r.why = std::string("librccl lacks ") + n; return p; };
    r.CommInitAll = (decltype(r.CommInitAll))sym("ncclCommInitAll");
    r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
    r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
    r.Broadcast = (decltype(r.Broadcast))sym("ncclBroadcast");
    r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
    r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    r.ok = r.CommInitAll && r.CommDestroy && r.AllGather && r.Broadcast && r.GroupStart && r.GroupEnd && r.GetErrorString;
  });
  return r;
}

// Communicators of the last device list (ncclCommInitAll costs tens of ms; a caller repeats the same list).
struct CommCache {
  std::mutex mu;
  std::vector<int> devices;
  std::vector<ncclComm_t> comms;
  std::vector<hipStream_t> streams;
};
CommCache g_comms;

int ensure_comms(const std::vector<int>& devices) {
  Rccl& r = rccl();
  if (!r.ok) return fail(M2S_ERR_HIP, "RCCL exchange unavailable: %s", r.why.c_str());
  if (g_comms.devices == devices) return 0;
  for (size_t k = 0; k < g_comms.comms.size(); ++k) {
    (void)hipSetDevice(g_comms.devices[k]);
    (void)r.CommDestroy(g_comms.comms[k]);
    (void)hipStreamDestroy(g_comms.streams[k]);
  }
  g_comms.comms.assign(devices.size(), nullptr);
  g_comms.streams.assign(devices.size(), nullptr);
  g_comms.devices.clear();
  const ncclResult_t e = r.CommInitAll(g_comms.comms.data(), (int)devices.size(), devices.data());
  if (e != ncclSuccess) {
    g_comms.comms.clear();
    g_comms.streams.clear();
    return fail(M2S_ERR_HIP, "ncclCommInitAll over %zu devices failed: %s", devices.size(), r.GetErrorString(e));
  }
  g_comms.devices = devices;   // from here on the cache describes what exists: a failure below tears it down again
  for (size_t k = 0; k < devices.size(); ++k) {
    if (hipSetDevice(devices[k]) != hipSuccess || hipStreamCreateWithFlags(&g_comms.streams[k], hipStreamNonBlocking) != hipSuccess) {
      const hipError_t err = hipGetLastError();
      for (size_t j = 0; j < devices.size(); ++j) {
        (void)hipSetDevice(devices[j]);
        if (g_comms.comms[j]) (void)r.CommDestroy(g_comms.comms[j]);
        if (g_comms.streams[j]) (void)hipStreamDestroy(g_comms.streams[j]);
      }
      g_comms.comms.clear();
      g_comms.streams.clear();
      g_comms.devices.clear();
      return fail(M2S_ERR_HIP, "stream for the RCCL exchange on device %d: %s", devices[k], hipGetErrorString(err));
    }
  }
  return 0;
}

// In-place gather of the x-slabs: buffer k holds slab k on entry, every buffer the whole grid on return.
int rccl_gather(const std::vector<int>& devices, float* const* outs, const std::vector<uint64_t>& xb,
                const std::vector<uint64_t>& xe, uint64_t row, uint64_t period, uint64_t nx) {
  std::lock_guard<std::mutex> lk(g_comms.mu);
  int rc = ensure_comms(devices);
  if (rc) return rc;
  Rccl& r = rccl();
  const size_t n = devices.size();
  bool even = true;
  for (size_t k = 0; k < n; ++k) even &= (xe[k] - xb[k]) == (xe[0] - xb[0]);
  ncclResult_t e = r.GroupStart();
  if (period) {
    // interleaved: every period [j * period, (j + 1) * period) is n equal chunks in shard order: one in-place all-gather each
    for (uint64_t x0 = 0; x0 < nx && e == ncclSuccess; x0 += period)
      for (size_t k = 0; k < n && e == ncclSuccess; ++k)
        e = r.AllGather(outs[k] + (x0 + xb[k]) * row, outs[k] + x0 * row, (size_t)((xe[k] - xb[k]) * row), ncclFloat, g_comms.comms[k], g_comms.streams[k]);
  } else if (even) {
    for (size_t k = 0; k < n && e == ncclSuccess; ++k)   // sendbuff = recvbuff + rank * count: in place
      e = r.AllGather(outs[k] + xb[k] * row, outs[k], (size_t)((xe[k] - xb[k]) * row), ncclFloat, g_comms.comms[k], g_comms.streams[k]);
  } else {
    for (size_t root = 0; root < n && e == ncclSuccess; ++root)   // uneven slabs: one broadcast per slab
      for (size_t k = 0; k < n && e == ncclSuccess; ++k)
        if (xe[root] > xb[root])
          e = r.Broadcast(outs[k] + xb[root] * row, outs[k] + xb[root] * row, (size_t)((xe[root] - xb[root]) * row), ncclFloat, (int)root,
                          g_comms.comms[k], g_comms.streams[k]);
  }
  const ncclResult_t e2 = r.GroupEnd();
  if (e == ncclSuccess) e = e2;
  if (e != ncclSuccess) return fail(M2S_ERR_HIP, "RCCL all-gather failed: %s", r.GetErrorString(e));
  for (size_t k = 0; k < n; ++k) {
    M2S_HIP_CHECK(hipSetDevice(devices[k]));
    M2S_HIP_CHECK(hipStreamSynchronize(g_comms.streams[k]));
  }
  return 0;
}

// Peer access between every pair of distinct devices of the list; false if some pair cannot.
bool enable_peer_access(const std::vector<int>& devices) {
  static std::mutex mu;
  static std::vector<std::pair<int, int>> enabled;
  std::lock_guard<std::mutex> lk(mu);
  for (int a : devices)
    for (int b : devices) {
      if (a == b) continue;
      bool have = false;
      for (auto& p : enabled) have |= p.first == a && p.second == b;
      if (have) continue;
      int can = 0;
      if (hipDeviceCanAccessPeer(&can, a, b) != hipSuccess || !can) { (void)hipGetLastError(); return false; }
      if (hipSetDevice(a) != hipSuccess) return false;
      const hipError_t e = hipDeviceEnablePeerAccess(b, 0);
      if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { (void)hipGetLastError(); return false; }
      (void)hipGetLastError();
      enabled.emplace_back(a, b);
    }
  return true;
}

// ---- shard workers -------------------------------------------------------------------------------------------
// One host thread per shard.  Round 2 created and joined n std::threads in every call (0.1-0.3 ms of host time against a rank
// step of 1.7 ms, and a thread that could not be created ended in std::terminate).  The workers now live as long as the
// process: shard 0 runs on the caller's thread, shard k on worker k - 1, which sleeps on a condition variable between calls.
// One multi call at a time owns the pool; a second one arriving meanwhile falls back to threads of its own.
class ShardPool {
 public:
  // runs job(0..n-1) concurrently and returns when all are done; false if a thread could not be created (nothing left running)
  bool run(int n, const std::function<void(int)>& job) {
    if (n <= 1) { if (n == 1) job(0); return true; }
    std::unique_lock<std::mutex> own(owner_, std::try_to_lock);
    if (!own.owns_lock()) return run_adhoc(n, job);
    if (pid_ != getpid()) {
      // a child of fork(): the workers' threads stayed with the parent — posting to them would wait for ever.  Their records are
      // abandoned (a std::thread that believes it is joinable must not be destroyed) and the child grows workers of its own.
      for (auto& w : workers_) (void)w.release();
      workers_.clear();
      pid_ = getpid();
    }
    try {
      while ((int)workers_.size() < n - 1) {
        workers_.emplace_back(new Worker());
        Worker* w = workers_.back().get();
        w->th = std::thread([w] { w->loop(); });
      }
    } catch (...) {
      if (!workers_.empty() && !workers_.back()->th.joinable()) workers_.pop_back();
      return false;
    }
    for (int k = 1; k < n; ++k) workers_[k - 1]->post([&job, k] { job(k); });
    job(0);
    for (int k = 1; k < n; ++k) workers_[k - 1]->wait();
    return true;
  }
  ~ShardPool() {
    for (auto& w : workers_) w->stop();
    for (auto& w : workers_) if (w->th.joinable()) w->th.join();
  }

 private:
  struct Worker {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::function<void()> task;
    bool has = false, done = false, quit = false;
    void loop() {
      for (;;) {
        std::function<void()> t;
        {
          std::unique_lock<std::mutex> lk(mu);
          cv.wait(lk, [this] { return has || quit; });
          if (quit) return;
          t = std::move(task);
          has = false;
        }
        t();
        {
          std::lock_guard<std::mutex> lk(mu);
          done = true;
        }
        cv.notify_all();
      }
    }
    void post(std::function<void()> t) {
      {
        std::lock_guard<std::mutex> lk(mu);
        task = std::move(t);
        has = true;
        done = false;
      }
      cv.notify_all();
    }
    void wait() {
      std::unique_lock<std::mutex> lk(mu);
      cv.wait(lk, [this] { return done; });
    }
    void stop() {
      {
        std::lock_guard<std::mutex> lk(mu);
        quit = true;
      }
      cv.notify_all();
    }
  };
  static bool run_adhoc(int n, const std::function<void(int)>& job) {
    std::vector<std::thread> th;
    bool ok = true;
    try {
      for (int k = 1; k < n; ++k) th.emplace_back(job, k);
    } catch (...) {
      ok = false;
    }
    if (ok) job(0);
    for (auto& t : th) t.join();
    return ok && (int)th.size() == n - 1;
  }
  std::mutex owner_;
  std::vector<std::unique_ptr<Worker>> workers_;
  pid_t pid_ = getpid();
};
ShardPool& shard_pool() {
  static ShardPool p;
  return p;
}

// ---- M2S_PART_ADAPTIVE: slabs of equal cost, from the previous call's per-shard times ------------------------------
// The cost of an x-layer is not uniform and not known in advance (the deep interior of a body, the layers that cut a large cap of
// its surface); a caller that repeats the call — a time loop, the bench — can have it measured: after every call the slabs are
// re-cut so that the piecewise-constant cost density of the last call integrates to equal shares (m2s_balanced_slabs).
struct AdaptiveKey {
  m2s_grid grid;
  size_t n_vertices, n_indices;
  int sign_method, n;
  // ... and where and how the shards run: the same grid on other devices, with another delivery, is another problem
  int mem_kind, exchange, peer_mode;
  int devices[M2S_MAX_PEERS + 1];
  unsigned long long mesh_tag;   // a fingerprint of the first and last vertices / indices passed (host pointers: their bytes; device pointers: the addresses)
  bool operator<(const AdaptiveKey& o) const { return memcmp(this, &o, sizeof(*this)) < 0; }
};
std::mutex g_adaptive_mu;
std::map<AdaptiveKey, std::vector<uint64_t>> g_adaptive;   // key -> n + 1 slab boundaries for the NEXT call

// The part of a multi-device call that does not depend on what is computed: shards -> (device, lane), memory kind, exchange.
struct MultiPlan {
  int n = 0;
  std::vector<int> devices, lanes, distinct;
  int mem_kind = M2S_MEM_HOST, exchange = M2S_XCHG_NONE, peer_mode = M2S_PEER_PUSH;
};
int plan_multi(const m2s_multi_opts* opts, float* const* outs, bool empty, MultiPlan* mp) {
  if (!outs) return fail(M2S_ERR_BAD_ARG, "outs is NULL");
  if (opts && opts->struct_size != 0 && opts->struct_size < M2S_MULTI_OPTS_V1_SIZE) return fail(M2S_ERR_BAD_ARG, "m2s_multi_opts.struct_size too small");
  const int visible = m2s_device_count();
  if (visible <= 0) return fail(M2S_ERR_HIP, "no HIP device available: the MI355X kernels cannot run (there is no CPU fallback)");
  int n = opts ? opts->n_devices : 0;
  if (n == 0) n = visible;
  if (n < 0 || n > M2S_MAX_PEERS + 1) return fail(M2S_ERR_BAD_ARG, "n_devices %d outside [1, %d]", n, M2S_MAX_PEERS + 1);
  mp->n = n;
  mp->devices.assign((size_t)n, 0);
  mp->lanes.assign((size_t)n, 0);
  for (int k = 0; k < n; ++k) {
    mp->devices[k] = (opts && opts->devices) ? opts->devices[k] : k;
    if (mp->devices[k] < 0 || mp->devices[k] >= visible) return fail(M2S_ERR_BAD_ARG, "devices[%d] = %d, but %d device(s) are visible", k, mp->devices[k], visible);
    int lane = 0;
    for (int j = 0; j < k; ++j) lane += mp->devices[j] == mp->devices[k];   // shards that share a device get contexts of their own
    if (lane >= M2S_MAX_LANES) return fail(M2S_ERR_BAD_ARG, "more than %d shards on device %d", M2S_MAX_LANES, mp->devices[k]);
    mp->lanes[k] = lane;
  }
  mp->mem_kind = opts ? opts->mem_kind : M2S_MEM_HOST;
  if (mp->mem_kind != M2S_MEM_HOST && mp->mem_kind != M2S_MEM_DEVICE) return fail(M2S_ERR_BAD_ARG, "bad mem_kind");
  int exchange = opts ? opts->exchange : M2S_XCHG_AUTO;
  if (exchange < M2S_XCHG_AUTO || exchange > M2S_XCHG_NONE) return fail(M2S_ERR_BAD_ARG, "bad exchange");
  mp->peer_mode = opts ? opts->peer_mode : M2S_PEER_PUSH;
  if (mp->peer_mode < M2S_PEER_PUSH || mp->peer_mode > M2S_PEER_TRAIL) return fail(M2S_ERR_BAD_ARG, "bad peer_mode");
  for (int k = 0; k < (mp->mem_kind == M2S_MEM_HOST ? 1 : n); ++k)
    if (!outs[k] && !empty) return fail(M2S_ERR_BAD_ARG, "outs[%d] is NULL", k);
  for (int d : mp->devices) { bool seen = false; for (int e : mp->distinct) seen |= e == d; if (!seen) mp->distinct.push_back(d); }
  // (an explicit RCCL request is honoured even for one shard: a 1-rank in-place all-gather, which is how the RCCL
  // path is exercised on a 1-GPU box)
  if (mp->mem_kind == M2S_MEM_HOST || (n == 1 && exchange != M2S_XCHG_RCCL)) exchange = M2S_XCHG_NONE;
  else if (exchange == M2S_XCHG_AUTO || exchange == M2S_XCHG_PEER) {
    const bool ok = enable_peer_access(mp->distinct);
    if (!ok && exchange == M2S_XCHG_PEER) return fail(M2S_ERR_HIP, "peer access between the listed devices is not available");
    exchange = ok ? M2S_XCHG_PEER : M2S_XCHG_RCCL;
  }
  if (exchange == M2S_XCHG_RCCL && mp->distinct.size() != mp->devices.size())
    return fail(M2S_ERR_BAD_ARG, "the RCCL exchange needs distinct devices (one communicator rank per device)");
  mp->exchange = exchange;
  if (opts && opts->exchange_used) *opts->exchange_used = exchange;
  return 0;
}

// A replica of a device-resident array of devices[0] on another device (no peer access: RCCL exchange).
int replicate(int dst_device, int src_device, const void* src, size_t bytes, void** staged) {
  *staged = nullptr;
  if (!bytes) return 0;
  if (hipSetDevice(dst_device) != hipSuccess || hipMalloc(staged, bytes + 256) != hipSuccess) return fail(M2S_ERR_HIP, "replica: hipMalloc failed on device %d", dst_device);
  if (hipMemcpyPeer(*staged, dst_device, src, src_device, bytes) != hipSuccess) return fail(M2S_ERR_HIP, "replica: hipMemcpyPeer failed");
  return 0;
}

}  // namespace
}  // namespace m2s

using namespace m2s;

extern "C" {

void m2s_slab_bounds(uint64_t nx, int n, int k, uint64_t* x_begin, uint64_t* x_end) {
  if (n <= 0) { *x_begin = 0; *x_end = nx; return; }
  const uint64_t base = nx / (uint64_t)n, rem = nx % (uint64_t)n, kk = (uint64_t)k;
  const uint64_t x0 = kk * base + (kk < rem ? kk : rem);
  *x_begin = x0;
  *x_end = x0 + base + (kk < rem ? 1 : 0);
}

int m2s_balanced_slabs(uint64_t nx, int n, uint64_t unit, const uint64_t* prev_bounds, const float* cost, uint64_t* new_bounds) {
  clear_error();
  if (n <= 0 || !prev_bounds || !cost || !new_bounds) return fail(M2S_ERR_BAD_ARG, "m2s_balanced_slabs: NULL argument or n <= 0");
  if (unit == 0) unit = 1;
  if (prev_bounds[0] != 0 || prev_bounds[n] != nx) return fail(M2S_ERR_BAD_ARG, "m2s_balanced_slabs: prev_bounds must run from 0 to nx");
  double total = 0.0;
  for (int k = 0; k < n; ++k) {
    if (prev_bounds[k + 1] < prev_bounds[k]) return fail(M2S_ERR_BAD_ARG, "m2s_balanced_slabs: prev_bounds not ascending");
    if (!(cost[k] >= 0.0f) || !(cost[k] < 3.0e38f)) return fail(M2S_ERR_BAD_ARG, "m2s_balanced_slabs: cost[%d] is not a finite non-negative number", k);
    total += prev_bounds[k + 1] > prev_bounds[k] ? (double)cost[k] : 0.0;
  }
  const uint64_t units = (nx + unit - 1) / unit;                  // the last unit may be partial
  new_bounds[0] = 0;
  new_bounds[n] = nx;
  if (total <= 0.0 || units < (uint64_t)n) {                      // nothing measured, or fewer units than shards: even slabs
    for (int k = 1; k < n; ++k) { uint64_t a, b; m2s_slab_bounds(nx, n, k, &a, &b); new_bounds[k] = a; }
    return M2S_OK;
  }
  // C(x): cumulative cost, linear inside every previous slab; boundary k goes where C reaches k / n of the total
  int seg = 0;
  double before = 0.0;                                            // cost of the previous slabs left of `seg`
  for (int k = 1; k < n; ++k) {
    const double want = total * (double)k / (double)n;
    while (seg < n - 1 && (prev_bounds[seg + 1] == prev_bounds[seg] || before + (double)cost[seg] < want)) {
      before += prev_bounds[seg + 1] > prev_bounds[seg] ? (double)cost[seg] : 0.0;
      ++seg;
    }
    const double len = (double)(prev_bounds[seg + 1] - prev_bounds[seg]);
    const double frac = cost[seg] > 0.0f ? std::min(1.0, std::max(0.0, (want - before) / (double)cost[seg])) : 0.0;
    const double x = (double)prev_bounds[seg] + frac * len;
    uint64_t u = (uint64_t)(x / (double)unit + 0.5);              // to whole units ...
    const uint64_t lo = new_bounds[k - 1] / unit + 1, hi = units - (uint64_t)(n - k);   // ... every shard keeps at least one
    u = std::min(std::max(u, lo), hi);
    new_bounds[k] = u * unit;
  }
  return M2S_OK;
}

int m2s_generate_grid_sdf_multi(const float* vertices, size_t n_vertices, const void* indices, size_t n_indices,
                                int index_bytes, int topology, const m2s_grid* grid, int sign_method, float* const* outs,
                                const m2s_multi_opts* opts) {
  clear_error();
  const auto t0 = std::chrono::steady_clock::now();
  if (!grid) return fail(M2S_ERR_BAD_ARG, "grid is NULL");
  if (opts && opts->struct_size != 0 && opts->struct_size < M2S_MULTI_OPTS_V1_SIZE) return fail(M2S_ERR_BAD_ARG, "m2s_multi_opts.struct_size too small");
  const bool v2 = opts && opts->struct_size >= M2S_MULTI_OPTS_V2_SIZE, v3 = opts && opts->struct_size >= sizeof(m2s_multi_opts);
  // a version-0.1 caller (no `partition` field) reads its slabs from m2s_slab_bounds: contiguous for it
  int partition = v2 ? opts->partition : M2S_PART_CONTIGUOUS;
  if (partition < M2S_PART_AUTO || partition > M2S_PART_ADAPTIVE) return fail(M2S_ERR_BAD_ARG, "bad partition");
  const uint64_t nx = grid->cell_count[0], row = grid->cell_count[1] * grid->cell_count[2];
  const bool empty = nx == 0 || row == 0;
  MultiPlan mp;
  if (const int prc = plan_multi(opts, outs, empty, &mp)) return prc;
  const int n = mp.n, mem_kind = mp.mem_kind, exchange = mp.exchange, peer_mode = mp.peer_mode;
  const std::vector<int>& devices = mp.devices;
  const std::vector<int>& lanes = mp.lanes;

  // the mesh lives on devices[0] in device mode: the other devices read it through a staging copy of their own
  const size_t vbytes = n_vertices * 12, ibytes = indices ? n_indices * (size_t)index_bytes : 0;
  std::vector<uint64_t> xb((size_t)n), xe((size_t)n);
  uint64_t period = 0;
  for (int k = 0; k < n; ++k) m2s_slab_bounds(nx, n, k, &xb[k], &xe[k]);
  // M2S_XCHG_NONE leaves buffer k with shard k's cells only: the caller finds them through m2s_slab_bounds unless it asked for
  // another partition explicitly (and reads `slabs` back)
  if (partition == M2S_PART_AUTO && exchange == M2S_XCHG_NONE) partition = M2S_PART_CONTIGUOUS;
  AdaptiveKey akey;
  memset(&akey, 0, sizeof(akey));   // compared bytewise, padding included
  if (partition == M2S_PART_ADAPTIVE && n > 1 && !empty) {
    akey.grid = *grid; akey.n_vertices = n_vertices; akey.n_indices = n_indices; akey.sign_method = sign_method; akey.n = n;
    akey.mem_kind = mem_kind; akey.exchange = exchange; akey.peer_mode = mp.peer_mode;
    for (int k = 0; k < n; ++k) akey.devices[k] = devices[k];
    if (mem_kind == M2S_MEM_HOST && vertices && n_vertices) {
      unsigned long long h = 1469598103934665603ull;
      const unsigned char* vb = reinterpret_cast<const unsigned char*>(vertices);
      const size_t nb = n_vertices * 12, take = std::min<size_t>(nb, 256);
      for (size_t i = 0; i < take; ++i) h = (h ^ vb[i]) * 1099511628211ull;
      for (size_t i = nb - take; i < nb; ++i) h = (h ^ vb[i]) * 1099511628211ull;
      akey.mesh_tag = h;
    } else {
      akey.mesh_tag = (unsigned long long)(uintptr_t)vertices ^ ((unsigned long long)(uintptr_t)indices << 1);
    }
    std::lock_guard<std::mutex> lk(g_adaptive_mu);
    auto it = g_adaptive.find(akey);
    if (it != g_adaptive.end())
      for (int k = 0; k < n; ++k) { xb[k] = it->second[k]; xe[k] = it->second[k + 1]; }
  }
  if ((partition == M2S_PART_AUTO || partition == M2S_PART_INTERLEAVED) && mem_kind == M2S_MEM_DEVICE && n > 1) {
    std::vector<uint64_t> ib((size_t)n), ie((size_t)n);
    bool ok = true;
    for (int k = 0; k < n && ok; ++k) ok = m2s_interleaved_slab(grid, n, k, &ib[k], &ie[k], &period) != 0;
    if (ok) { xb = ib; xe = ie; }
    else period = 0;
  }
  if (partition == M2S_PART_INTERLEAVED && period == 0 && n > 1)
    return fail(M2S_ERR_BAD_ARG, "this grid cannot be cut into interleaved chunks (m2s_interleaved_slab; device memory only)");

  if (v3 && opts->partition_used) *opts->partition_used = period ? M2S_PART_INTERLEAVED : (partition == M2S_PART_ADAPTIVE ? M2S_PART_ADAPTIVE : M2S_PART_CONTIGUOUS);
  if (v3 && opts->slabs)
    for (int k = 0; k < n; ++k) { opts->slabs[3 * k] = xb[k]; opts->slabs[3 * k + 1] = xe[k]; opts->slabs[3 * k + 2] = period; }
  std::vector<int> rcs((size_t)n, 0);
  std::vector<std::string> errs((size_t)n);
  std::vector<m2s_timings> own_timings(partition == M2S_PART_ADAPTIVE && !(opts && opts->timings) ? (size_t)n : 0);
  m2s_timings* shard_timings = (opts && opts->timings) ? opts->timings : (own_timings.empty() ? nullptr : own_timings.data());
  auto shard = [&](int k) {
    m2s_opts o{};
    o.struct_size = sizeof(m2s_opts);
    o.device = devices[k];
    o.lane = lanes[k];
    o.mem_kind = mem_kind;
    o.algorithm = opts ? opts->algorithm : 0;
    o.x_begin = xb[k];
    o.x_end = xe[k];
    o.x_period = (uint32_t)period;
    o.synchronous = 1;
    o.timings = shard_timings ? &shard_timings[k] : nullptr;
    if (shard_timings) memset(&shard_timings[k], 0, sizeof(m2s_timings));
    float* peers[M2S_MAX_PEERS];
    if (exchange == M2S_XCHG_PEER) {
      uint32_t np = 0;
      for (int j = 0; j < n; ++j)
        if (j != k && outs[j] != outs[k]) peers[np++] = outs[j];
      o.n_peer_out = np;
      o.peer_out = peers;
      o.peer_mode = peer_mode;
    }
    const float* v = vertices;
    const void* ix = indices;
    void* staged = nullptr;
    int rc = 0;
    if (xe[k] == xb[k]) { rcs[k] = 0; return; }   // more shards than layers
    // With peer access enabled the kernels of device k read the mesh where it lies, on devices[0], over xGMI (1.8 MB for 100k
    // triangles, read once by the first kernel of the build).  Without it (RCCL exchange) the mesh is replicated by a peer copy.
    if (mem_kind == M2S_MEM_DEVICE && devices[k] != devices[0] && (vbytes || ibytes) && exchange != M2S_XCHG_PEER) {
      if (hipSetDevice(devices[k]) != hipSuccess || hipMalloc(&staged, vbytes + ibytes + 512) != hipSuccess) rc = fail(M2S_ERR_HIP, "mesh replica: hipMalloc failed on device %d", devices[k]);
      char* sv = (char*)staged;
      char* si = sv + (vbytes + 255) / 256 * 256;
      if (!rc && vbytes && hipMemcpyPeer(sv, devices[k], vertices, devices[0], vbytes) != hipSuccess) rc = fail(M2S_ERR_HIP, "mesh replica: hipMemcpyPeer failed");
      if (!rc && ibytes && hipMemcpyPeer(si, devices[k], indices, devices[0], ibytes) != hipSuccess) rc = fail(M2S_ERR_HIP, "mesh replica: hipMemcpyPeer failed");
      v = (const float*)sv;
      if (indices) ix = si;
    }
    if (!rc) rc = m2s_generate_grid_sdf(v, n_vertices, ix, n_indices, index_bytes, topology, grid, sign_method,
                                        mem_kind == M2S_MEM_HOST ? outs[0] : outs[k], &o);
    if (rc) errs[k] = m2s_last_error();
    if (staged) (void)hipFree(staged);
    rcs[k] = rc;
  };
  if (!shard_pool().run(n, shard)) return fail(M2S_ERR_HIP, "could not start the shard threads");
  for (int k = 0; k < n; ++k)
    if (rcs[k]) return fail(rcs[k], "shard %d (device %d): %s", k, devices[k], errs[k].c_str());
  if (partition == M2S_PART_ADAPTIVE && n > 1 && !empty && shard_timings) {
    // next call's slabs: equal shares of this call's cost (what shards: everything but the LBVH build, which every shard repeats)
    std::vector<uint64_t> prev((size_t)n + 1), next((size_t)n + 1);
    std::vector<float> cost((size_t)n);
    // (the phases a shard computes — sign planes, seeds and cut lists, walk — not its total: that also holds the time it waited for a peer
    // or a collective, and a slow link is not to be balanced as if it were work)
    for (int k = 0; k < n; ++k) { prev[k] = xb[k]; cost[k] = std::max(shard_timings[k].sign_ms + shard_timings[k].seed_ms + shard_timings[k].distance_ms, 0.0f); }
    prev[n] = nx;
    uint32_t bl[3];
    choose_brick_shape(grid->cell_size, bl);
    // boundaries on whole SUPER-bricks (8 packet bricks along x): a slab that is not a multiple of them walks in narrower super-bricks
    // and loses more locality than balance can win (emulated 8-GPU ranks, 512^3 x blob-100k: slabs of 56 layers took 2.01 ms where
    // 64 layers took 1.83; tools/exp_rank_step.py --partition adaptive) — so the cut only moves where shards own many of them
    if (m2s_balanced_slabs(nx, n, 8ull << bl[0], prev.data(), cost.data(), next.data()) == M2S_OK) {
      std::lock_guard<std::mutex> lk(g_adaptive_mu);
      if (g_adaptive.size() > 64) g_adaptive.clear();
      g_adaptive[akey] = next;
    }
  }
  // every shard returned synchronously: all slabs (and, with the peer exchange, all pushes) are complete
  if (exchange == M2S_XCHG_RCCL && !empty) {
    const int rc = rccl_gather(devices, outs, xb, xe, row, period, nx);
    if (rc) return rc;
  }
  if (opts && opts->wall_ms) *opts->wall_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
  return M2S_OK;
}

// generate_sdf over several GPUs: shard k takes the contiguous query range [q0_k, q1_k) (sizes differ by at most one; every query
// depends only on the replicated mesh).  Host memory: every device returns its range into the caller's array over its own PCIe
// link.  Device memory: mesh and queries lie on devices[0]; with peer access the other devices read them where they lie and every
// shard copies its finished range into all other buffers itself (xGMI is point-to-point: n - 1 links in parallel); without it the
// inputs are replicated by peer copies and the ranges are gathered by RCCL (one in-place all-gather, or one broadcast per range
// when the count does not divide).
int m2s_generate_sdf_multi(const float* vertices, size_t n_vertices, const void* indices, size_t n_indices, int index_bytes,
                           int topology, const float* queries, size_t n_queries, int accel, int sign_method, float* const* outs,
                           size_t* n_out, const m2s_multi_opts* opts) {
  clear_error();
  const auto t0 = std::chrono::steady_clock::now();
  if (n_out) *n_out = 0;
  if (n_queries && !queries) return fail(M2S_ERR_BAD_ARG, "queries is NULL");
  MultiPlan mp;
  if (const int prc = plan_multi(opts, outs, n_queries == 0, &mp)) return prc;
  const int n = mp.n;
  const std::vector<int>& devices = mp.devices;
  std::vector<uint64_t> qb((size_t)n), qe((size_t)n);
  for (int k = 0; k < n; ++k) m2s_slab_bounds((uint64_t)n_queries, n, k, &qb[k], &qe[k]);
  const size_t vbytes = n_vertices * 12, ibytes = indices ? n_indices * (size_t)index_bytes : 0;

  std::vector<int> rcs((size_t)n, 0);
  std::vector<size_t> written((size_t)n, 0);
  std::vector<std::string> errs((size_t)n);
  auto shard = [&](int k) {
    const size_t count = (size_t)(qe[k] - qb[k]);
    if (count == 0) { rcs[k] = 0; return; }       // more shards than queries
    m2s_opts o{};
    o.struct_size = sizeof(m2s_opts);
    o.device = devices[k];
    o.lane = mp.lanes[k];
    o.mem_kind = mp.mem_kind;
    o.algorithm = opts ? opts->algorithm : 0;
    o.synchronous = 1;
    o.timings = (opts && opts->timings) ? &opts->timings[k] : nullptr;
    const float* v = vertices;
    const void* ix = indices;
    const float* q = queries + 3 * (size_t)qb[k];
    void *sv = nullptr, *si = nullptr, *sq = nullptr;
    int rc = 0;
    if (mp.mem_kind == M2S_MEM_DEVICE && devices[k] != devices[0] && mp.exchange != M2S_XCHG_PEER) {
      rc = replicate(devices[k], devices[0], vertices, vbytes, &sv);
      if (!rc) rc = replicate(devices[k], devices[0], indices, ibytes, &si);
      if (!rc) rc = replicate(devices[k], devices[0], q, count * 12, &sq);
      if (sv) v = (const float*)sv;
      if (si) ix = si;
      if (sq) q = (const float*)sq;
    }
    float* out = (mp.mem_kind == M2S_MEM_HOST ? outs[0] : outs[k]) + qb[k];
    if (!rc) rc = m2s_generate_sdf(v, n_vertices, ix, n_indices, index_bytes, topology, q, count, accel, sign_method, out, &written[k], &o);
    if (rc) errs[k] = m2s_last_error();
    if (!rc && mp.exchange == M2S_XCHG_PEER && written[k]) {
      // the call was synchronous: the range is complete in outs[k]; hand it to every other buffer (skip aliases of this one)
      for (int j = 0; j < n && !rc; ++j) {
        if (j == k || outs[j] == outs[k]) continue;
        if (hipMemcpyPeer(outs[j] + qb[k], devices[j], out, devices[k], count * 4) != hipSuccess) rc = fail(M2S_ERR_HIP, "peer copy of the result range failed");
      }
      if (rc) errs[k] = m2s_last_error();
    }
    for (void* p : {sv, si, sq}) if (p) (void)hipFree(p);
    rcs[k] = rc;
  };
  if (!shard_pool().run(n, shard)) return fail(M2S_ERR_HIP, "could not start the shard threads");
  for (int k = 0; k < n; ++k)
    if (rcs[k]) return fail(rcs[k], "shard %d (device %d): %s", k, devices[k], errs[k].c_str());
  size_t total = 0;
  for (int k = 0; k < n; ++k) total += written[k];   // n_queries, or 0: RTREE_BVH on a mesh without triangles (every shard agrees)
  if (mp.exchange == M2S_XCHG_RCCL && total) {
    const int rc = rccl_gather(devices, outs, qb, qe, 1, 0, (uint64_t)n_queries);
    if (rc) return rc;
  }
  if (n_out) *n_out = total;
  if (opts && opts->wall_ms) *opts->wall_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
  return M2S_OK;
}

// ---- one process per GPU: buffers that other processes can map ---------------------------------------------
int m2s_shared_alloc(size_t bytes, int device, void** device_ptr) {
  clear_error();
  if (!device_ptr) return fail(M2S_ERR_BAD_ARG, "device_ptr is NULL");
  *device_ptr = nullptr;
  M2S_HIP_CHECK(hipSetDevice(device));
  M2S_HIP_CHECK(hipMalloc(device_ptr, bytes ? bytes : 256));
  return M2S_OK;
}

int m2s_shared_free(void* device_ptr, int device) {
  clear_error();
  if (!device_ptr) return M2S_OK;
  M2S_HIP_CHECK(hipSetDevice(device));
  M2S_HIP_CHECK(hipDeviceSynchronize());
  M2S_HIP_CHECK(hipFree(device_ptr));
  return M2S_OK;
}

static_assert(sizeof(hipIpcMemHandle_t) == M2S_IPC_HANDLE_BYTES, "hipIpcMemHandle_t is expected to be 64 bytes");

int m2s_ipc_export(const void* device_ptr, uint8_t handle[M2S_IPC_HANDLE_BYTES]) {
  clear_error();
  if (!device_ptr || !handle) return fail(M2S_ERR_BAD_ARG, "NULL argument");
  hipIpcMemHandle_t h;
  M2S_HIP_CHECK(hipIpcGetMemHandle(&h, const_cast<void*>(device_ptr)));
  memcpy(handle, &h, sizeof(h));
  return M2S_OK;
}

int m2s_ipc_open(const uint8_t handle[M2S_IPC_HANDLE_BYTES], int device, void** device_ptr) {
  clear_error();
  if (!handle || !device_ptr) return fail(M2S_ERR_BAD_ARG, "NULL argument");
  *device_ptr = nullptr;
  hipIpcMemHandle_t h;
  memcpy(&h, handle, sizeof(h));
  M2S_HIP_CHECK(hipSetDevice(device));
  M2S_HIP_CHECK(hipIpcOpenMemHandle(device_ptr, h, hipIpcMemLazyEnablePeerAccess));
  return M2S_OK;
}

int m2s_ipc_close(void* device_ptr, int device) {
  clear_error();
  if (!device_ptr) return M2S_OK;
  M2S_HIP_CHECK(hipSetDevice(device));
  M2S_HIP_CHECK(hipIpcCloseMemHandle(device_ptr));
  return M2S_OK;
}

}  // extern "C"
