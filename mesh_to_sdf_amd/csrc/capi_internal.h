// capi_internal.h — per-device state and call context shared by the C-ABI translation units
// (capi.hip: distance path; capi_io.hip: container encode/decode, cell ordering, instance merge).
#pragma once
#include <cstdio>
#include <mutex>
#include <vector>

#include "../../include/m2s.h"
#include "common.h"

namespace m2s {

struct DeviceState {
  std::mutex mu;   // held by an entry point for the whole call (CallCtx::lock); one context per (device, lane)
  // Scratch of the call in progress.  It belongs to the STREAM the call is enqueued on (resolve_ctx selects it):
  // calls on one stream are ordered by the stream and share one block, calls on different streams get different
  // blocks, so asynchronous calls on two streams (the x-pieces of a sharded grid) may overlap on the device.
  char* base = nullptr;
  size_t cap = 0;
  size_t warmed_cap = 0;   // m2s_warmup: capacity of the workspace block it has already touched
  struct Scratch {
    hipStream_t stream = nullptr;
    char* base = nullptr;
    size_t cap = 0;
    uint64_t last_use = 0;
  };
  static constexpr int SCRATCH = 4;
  Scratch scratch[SCRATCH];
  int n_scratch = 0, active = -1;
  uint64_t tick = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // [5]: start of early sign planes (side_stream2)
  hipStream_t side_stream2 = nullptr;   // one-shot Raycast grid calls: the sign planes, from the input-order records, beside build and seeds
  bool early_planes = false;            // this call's planes were launched on side_stream2 (ev[5] .. ev[2])
  // Raycast sign planes depend on the mesh only: they are built on `side_stream` while the caller's stream runs the seed
  // passes and cut lists; the dominant launch waits for ev[2], which is then recorded on the side stream.
  hipStream_t side_stream = nullptr;
  hipEvent_t fork_ev = nullptr;
  hipEvent_t planes_done = nullptr;   // ev[2] while a call's planes are in flight on the side stream, else nullptr
  // one-shot grid calls also run the jump-flooding seed passes on the side stream, beside the sort / hierarchy of the build
  // (they only need the input-order centroids): the lattice, and the event recorded after the last pass
  SeedLattice raw_seeds;
  hipEvent_t seeds_done = nullptr, seeds_fork = nullptr;
  bool have_raw_seeds = false;
  hipEvent_t qlat_done = nullptr;     // ... and its bounding box + seed-lattice description are enqueued
  hipEvent_t qprep_done = nullptr;    // one-shot query calls: the query preparation (sort, packets) on the side stream has finished
  int* h_err = nullptr;  // pinned
  char* spare_mesh = nullptr;  // last destroyed m2s_mesh block, recycled by the next m2s_mesh_create
  size_t spare_mesh_bytes = 0;
  char* spare_plane = nullptr;  // sign-plane block of the last destroyed m2s_mesh (hipFree / hipMalloc cost 0.2 + 0.1 ms per mesh)
  size_t spare_plane_bytes = 0;
  std::vector<hipEvent_t> timing_events;   // timing-enabled events handed back by destroyed meshes
  // host-pointer calls: pinned ring + copy stream for the pipelined D2H of the result (capi.hip)
  static constexpr int RING = 3;
  char* ring[RING] = {nullptr, nullptr, nullptr};
  size_t ring_bytes = 0;
  hipStream_t copy_stream = nullptr;
  std::vector<hipEvent_t> piece_events;   // M2S_PEER_PUSH: piece i walked / last push done (no timing)
};

struct CallCtx {
  int device = -1;
  int mem_kind = M2S_MEM_HOST;
  int algorithm = 0;
  bool sync = true;
  hipStream_t stream = nullptr;
  m2s_timings* timings = nullptr;
  uint64_t x_begin = 0, x_end = 0;
  int lane = 0;
  PeerOut peers{};                       // m2s_opts.peer_out (grid path, device memory)
  int peer_mode = 0;
  std::unique_lock<std::mutex> lock;     // the context's lock, taken by resolve_ctx, released when the CallCtx dies
};

extern std::mutex g_mu;                  // guards the table of contexts only; every context has its own lock
DeviceState* find_state(int device, int lane);   // creates the context if needed; does NOT lock it
void clear_error();
int fail(int code, const char* fmt, ...);
int resolve_ctx(const m2s_opts* opts, CallCtx* c, DeviceState** st);
int ensure_capacity(DeviceState& s, size_t bytes);
int select_scratch(DeviceState& s, hipStream_t stream);   // called by resolve_ctx
void release_scratch(DeviceState& s);                     // device must be idle
// Pageable host arrays <-> device at PCIe speed through the pinned ring (capi.hip); complete on return.
int staged_h2d(DeviceState& st, hipStream_t stream, char* d_dst, const char* h_src, size_t bytes);
int staged_d2h(DeviceState& st, hipStream_t stream, char* h_dst, const char* d_src, size_t bytes);
int staged_d2h_to_file(DeviceState& st, hipStream_t stream, FILE* f, const char* d_src, size_t bytes);   // M2S_ERR_IO on a short write
inline size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

}  // namespace m2s
