/* m2s.h — C ABI of the MI355X-native hot path of Azkellas/mesh_to_sdf.
 *
 * Drop-in boundary.  The reference has no FFI layer of its own; its public surface for this
 * path is two generic Rust functions (paths relative to /root/reference/mesh_to_sdf/src):
 *
 *   pub fn generate_sdf<V: Point, I: Copy + Into<u32> + Sync + Send>(
 *       vertices: &[V], indices: Topology<I>, query_points: &[V],
 *       acceleration_method: AccelerationMethod) -> Vec<f32>              lib.rs:291-300
 *   pub fn generate_grid_sdf<V: Point + Sync + Send, I: ...>(
 *       vertices: &[V], indices: Topology<I>, grid: &Grid<V>,
 *       sign_method: SignMethod) -> Vec<f32>                              generate/grid.rs:265-274
 *
 * The entry points below are exactly what a Rust `extern "C"` block bound behind those two
 * signatures needs (see INTEGRATION.md for the shim): plain pointers and sizes, packed
 * xyz float triples (`[f32; 3]` is zero-copy), u16 or u32 indices, enums as ints.
 *
 * Semantics.  Distances are the reference's f32 arithmetic (geo.rs:70-138 closest point,
 * point.rs:99-126 operation order, no FMA contraction) minimised over ALL triangles, i.e. what
 * generate_sdf(AccelerationMethod::None(..)) returns and what the reference's own test
 * generate/grid.rs:693-724 asserts the grid path to equal.  (The reference's grid path is a
 * heap-ordered label propagation whose result depends on the thread count and is >= this
 * minimum on <0.5% of cells; see DESIGN.md "Exact vs propagation".)
 *
 * Threading: synchronous and re-entrant.  The library keeps one context (workspace, streams, events) per
 * (device, m2s_opts.lane); an entry point holds only ITS context's lock, so calls on different devices — or on
 * different lanes of one device — run concurrently from different host threads (m2s_generate_grid_sdf_multi does
 * exactly that, one thread per device).  Calls on the same (device, lane) are serialised; the work they enqueue is
 * not: asynchronous calls (m2s_opts.synchronous = 0) on different streams use separate scratch blocks and may
 * overlap on the device.
 *
 * Limits (validated, M2S_ERR_BAD_ARG): n_vertices < 2^31, n_indices < 3 * 2^31, at most 2^25 triangles per mesh
 * (32-bit byte offsets into the 96-byte triangle records), n_queries < 2^32 - 64 per call, every cell_count < 2^31
 * and every product of two cell counts < 2^32 (grid lines per face are counted in 32 bits).
 *
 * Non-finite QUERY coordinates (m2s_generate_sdf): with None / Bvh and SignMethod::Raycast the result is +f32::MAX, as in the
 * reference (every distance is NaN and f32::min drops it, default.rs:47; no ray hits); with SignMethod::Normal the call returns
 * M2S_ERR_NAN where the reference panics (lib.rs:257).  Rtree / RtreeBvh: the reference measures the distance to whichever
 * triangle rstar's nearest_neighbor returns for a NaN / inf point — unspecified by that crate — so the value for such a query
 * is unspecified here too (+f32::MAX for RtreeBvh today); the other queries of the call are unaffected.
 */
#ifndef M2S_H
#define M2S_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define M2S_VERSION_MAJOR 0
#define M2S_VERSION_MINOR 5   /* 0.5: + m2s_tuning_set, m2s_tuning_describe; 0.4: + m2s_warmup, m2s_peer_bandwidth, m2s_balanced_slabs, M2S_PART_ADAPTIVE, m2s_multi_opts.partition_used / slabs (additive) */

/* Return codes.  The reference panics where this ABI returns a negative code; the Rust shim
 * turns a negative code back into panic!(m2s_last_error()). */
#define M2S_OK 0
#define M2S_ERR_BAD_ARG (-1)     /* null pointer / bad enum / vertex index out of range (reference: index panic) */
#define M2S_ERR_NAN (-2)         /* "NaN distance" — lib.rs:257 expect(), Normal sign only */
#define M2S_ERR_EMPTY_MESH (-3)  /* AccelerationMethod::Rtree on a mesh with no triangle — generic/rtree.rs:117 unwrap() */
#define M2S_ERR_HIP (-4)         /* HIP runtime failure, or no HIP device / kernels not loadable */

/* Topology — lib.rs:151-167.  List: consecutive triples, a trailing partial triple is dropped
 * (itertools::tuples).  Strip: sliding window, NO winding flip on odd triangles (lib.rs:188-191).
 * indices == NULL means 0..n_vertices (Topology::*(None)). */
enum m2s_topology { M2S_TRIANGLE_LIST = 0, M2S_TRIANGLE_STRIP = 1 };

/* SignMethod — lib.rs:204-216 (default Raycast). */
enum m2s_sign_method { M2S_SIGN_RAYCAST = 0, M2S_SIGN_NORMAL = 1 };

/* AccelerationMethod — lib.rs:224-239 (default RtreeBvh).  It selects the SIGN RULE, which is
 * observable, not just a speed path:
 *   NONE  + Raycast: parity of +X hits over all triangles          generic/default.rs:32-38,65-72
 *   NONE  + Normal : compare_distances fold over all triangles     generic/default.rs:40-59
 *   BVH   + Raycast: best of three +X/+Y/+Z rays from the query    generic/bvh.rs:106-141
 *   BVH   + Normal : compare_distances fold                        generic/bvh.rs:82-94
 *   RTREE          : normal sign of the single nearest triangle    generic/rtree.rs:113-125 (sign_method ignored)
 *   RTREE_BVH      : nearest distance + best of three rays         generic/rtree_bvh.rs:123-173 (sign_method ignored)
 */
enum m2s_accel { M2S_ACCEL_NONE = 0, M2S_ACCEL_BVH = 1, M2S_ACCEL_RTREE = 2, M2S_ACCEL_RTREE_BVH = 3 };

/* Grid<V> — grid.rs:30-37: centre of the first cell, cell size (may differ per axis, may be
 * negative), cell count.  Output index of cell (x,y,z) is z + y*nz + x*ny*nz (grid.rs:122-124). */
typedef struct m2s_grid {
  float first_cell[3];
  float cell_size[3];
  uint64_t cell_count[3];
} m2s_grid;

/* Where the data pointers of a call live. */
enum m2s_mem_kind {
  M2S_MEM_HOST = 0,   /* vertices/indices/queries/out are host pointers (the drop-in case; H2D/D2H inside the call) */
  M2S_MEM_DEVICE = 1  /* all four are device pointers on `device`; nothing crosses PCIe */
};

/* Phase timings of the last call, milliseconds, measured with HIP events on the call's stream
 * (the reference logs the same three phases, generate/grid.rs:303-307,342-346,369-373). */
typedef struct m2s_timings {
  float accel_build_ms;  /* topology flatten + triangle records + LBVH */
  float sign_ms;         /* grid-line ray parity planes (Raycast grid path) */
  float distance_ms;     /* nearest-triangle search (+ fused sign resolve) — the dominant kernel's launch */
  float total_ms;        /* first kernel to last kernel, device side */
  float seed_ms;         /* grid path: what precedes the dominant launch besides the build — jump-flooding seed lattice
                            (one triangle per packet brick) and the cut lists (k_cut) */
  float reserved_f;
  uint64_t n_triangles;
  uint64_t n_units;      /* voxels or queries produced by this call */
  uint32_t distance_launches;  /* number of launches of the dominant kernel in this call */
  uint32_t reserved;
} m2s_timings;

/* Optional per-call options; pass NULL for defaults.  Set struct_size = sizeof(m2s_opts). */
typedef struct m2s_opts {
  uint32_t struct_size;
  int32_t device;       /* HIP device ordinal; -1 = current device */
  void* stream;         /* hipStream_t to enqueue on; NULL = the library's own stream for that device */
  int32_t mem_kind;     /* enum m2s_mem_kind */
  int32_t algorithm;    /* 0 = default (LBVH); 1 = brute force over all triangles (validation / AccelerationMethod::None) */
  /* Grid path only: compute the x-slab [x_begin, x_end) of cells (cell axis 0 is the slowest
   * axis, so a slab is one contiguous range of the output).  x_end == 0 means the whole grid.
   * `out` always addresses the WHOLE grid; only the slab's range is written. */
  uint64_t x_begin;
  uint64_t x_end;
  m2s_timings* timings; /* filled when non-NULL (forces a stream sync before returning) */
  int32_t synchronous;  /* device-memory calls: 1 (default when opts==NULL) = sync before return; 0 = leave work enqueued */
  int32_t stream_mode;  /* 0: stream == NULL selects the library's own (non-blocking) stream;
                           1: `stream` is used exactly as given, and NULL means the device's default (null) stream —
                              needed by callers whose "current stream" IS the default stream (torch does this) and who
                              order other work (e.g. an RCCL collective) after this call without a host sync */
  /* ---- fields below exist when struct_size >= sizeof(m2s_opts) of version 0.2 (M2S_OPTS_V1_SIZE bytes = version 0.1) ---- */
  int32_t lane;         /* context lane on `device`, 0 .. M2S_MAX_LANES-1: calls on different lanes do not serialise (Threading) */
  uint32_t n_peer_out;  /* grid path, device memory: number of entries of peer_out (<= M2S_MAX_PEERS) */
  float* const* peer_out; /* whole-grid buffers like `out`, on OTHER devices (peer access enabled: same process after
                             hipDeviceEnablePeerAccess, or mapped from another process with m2s_ipc_open) or on this one:
                             the x-slab this call computes is written to each of them as well, so that after every shard's
                             call each buffer holds the whole grid with no separate all-gather */
  int32_t peer_mode;    /* how: 0 = M2S_PEER_PUSH, one copy kernel per slab piece after its walk, 16 B per lane = 1 KiB per wave
                                    store instruction (xGMI-friendly request size; the default);
                                1 = M2S_PEER_STORE, the walk's epilogue stores every value to every peer itself (no extra pass over
                                    the slab, but 16-byte runs: a 4x4x4 brick row);
                                2 = M2S_PEER_TRAIL, one walk over the whole slab whose packets count themselves per unit of 8
                                    x-layers, and a copy kernel beside it that pushes every unit (1 KiB per wave store) as soon as
                                    it is complete: no pieces, and only the last unit's push is exposed */
  uint32_t x_period;    /* grid path, device memory; 0 = [x_begin, x_end) is one contiguous slab.  Otherwise the call owns the CHUNKS
                           [x_begin + j * x_period, x_end + j * x_period), j = 0, 1, ... up to the end of the grid: interleaved slabs.
                           Shards that take the chunks r and N + r of 2N balance far better than N contiguous slabs when the
                           cost of a layer varies along x (the deep interior of a body is the expensive part).  x_end - x_begin must
                           be a power of two and a multiple of 4 packet bricks (16 layers for cubic cells), x_period a multiple of it */
} m2s_opts;
#define M2S_OPTS_V1_SIZE 56
#define M2S_MAX_LANES 16
#define M2S_MAX_PEERS 15
enum m2s_peer_mode { M2S_PEER_PUSH = 0, M2S_PEER_STORE = 1, M2S_PEER_TRAIL = 2 };

/* generate_sdf — lib.rs:291-311.
 * vertices: n_vertices packed xyz f32.  indices: n_indices values of index_bytes (2 or 4) each, or NULL.
 * queries: n_queries packed xyz.  out: n_queries f32.  *n_out (optional) receives the number of
 * distances written: n_queries, or 0 for RTREE_BVH on a mesh without triangles (the reference
 * returns an empty Vec there, generic/rtree_bvh.rs:104-106). */
int m2s_generate_sdf(const float* vertices, size_t n_vertices, const void* indices, size_t n_indices, int index_bytes,
                     int topology, const float* queries, size_t n_queries, int accel, int sign_method, float* out,
                     size_t* n_out, const m2s_opts* opts);

/* generate_grid_sdf — generate/grid.rs:265-378.  out: cell_count[0]*[1]*[2] f32, caller owned.
 * Semantics: the EXACT minimum over all triangles (what generate/grid.rs:693-724 asserts the grid path to equal).  SURVEY.md §8(b)
 * sketched a `semantics` argument selecting the reference's label propagation (generate/grid.rs:495-558) instead; it is
 * deliberately absent: that output depends on rayon::current_num_threads(), is never below the exact minimum and exceeds it on
 * 0.03 % (512^3) to 59 % (1 M triangles in 128^3) of the cells — there is no one result to reproduce. */
int m2s_generate_grid_sdf(const float* vertices, size_t n_vertices, const void* indices, size_t n_indices,
                          int index_bytes, int topology, const m2s_grid* grid, int sign_method, float* out,
                          const m2s_opts* opts);

/* ---- multi-GPU generate_grid_sdf (SURVEY.md 8b "device list", 8e) -----------------------------------------
 * generate/grid.rs:265-274 is the one signature a caller of the reference has; this is the same call spread over the
 * GPUs of one node from ONE process: one host thread per listed device, the mesh replicated (every device builds its own
 * LBVH: the build is sub-millisecond and needs no exchange), the grid cut into contiguous x-slabs (cell axis 0 is the
 * slowest axis of the output, grid.rs:122-124, so a slab is one contiguous range), sign planes marked per device for
 * the whole grid (hits are slab independent).  No data-path collective:
 *   mem_kind == M2S_MEM_HOST   vertices / indices / outs[0] are host pointers.  Every device uploads the mesh and streams
 *                              its finished slab straight into the caller's array over its own PCIe link (pinned ring,
 *                              pipelined with the compute) — the drop-in form behind generate_grid_sdf's Vec<f32>.
 *   mem_kind == M2S_MEM_DEVICE vertices / indices live on devices[0]; outs[k] is a whole-grid buffer on devices[k].
 *                              On return EVERY buffer holds the whole grid:
 *                                exchange M2S_XCHG_PEER  each device writes its slab into all buffers itself over xGMI
 *                                                        (hipDeviceEnablePeerAccess; m2s_opts.peer_out / peer_mode);
 *                                exchange M2S_XCHG_RCCL  ncclAllGather from librccl (loaded on first use), in place;
 *                                exchange M2S_XCHG_NONE  nothing: buffer k holds slab k only;
 *                                exchange M2S_XCHG_AUTO  PEER when every pair of devices can access each other, else RCCL.
 * `devices` may name a device more than once (two shards on one GPU: how the path is tested on a 1-GPU box).
 * Errors as m2s_generate_grid_sdf; the first failing shard's code is returned. */
enum m2s_exchange { M2S_XCHG_AUTO = 0, M2S_XCHG_PEER = 1, M2S_XCHG_RCCL = 2, M2S_XCHG_NONE = 3 };
typedef struct m2s_multi_opts {
  uint32_t struct_size;
  int32_t n_devices;        /* number of shards; 0 = one per visible device */
  const int32_t* devices;   /* n_devices HIP ordinals; NULL = 0 .. n_devices-1 */
  int32_t mem_kind;         /* enum m2s_mem_kind */
  int32_t exchange;         /* enum m2s_exchange; device memory only */
  int32_t peer_mode;        /* enum m2s_peer_mode for M2S_XCHG_PEER */
  int32_t algorithm;        /* as m2s_opts.algorithm */
  m2s_timings* timings;     /* optional, n_devices entries: per-shard phase timings */
  float* wall_ms;           /* optional: host wall time of the whole call */
  int32_t* exchange_used;   /* optional: the exchange that ran (M2S_XCHG_PEER / RCCL / NONE) */
  /* ---- fields below exist when struct_size >= M2S_MULTI_OPTS_V2_SIZE (version 0.2; M2S_MULTI_OPTS_V1_SIZE = without) ---- */
  int32_t partition;        /* enum m2s_partition.  A caller whose struct ends before this field gets M2S_PART_CONTIGUOUS. */
  int32_t reserved;
  /* ---- fields below exist when struct_size >= sizeof(m2s_multi_opts) (version 0.3) ---- */
  int32_t* partition_used;  /* optional: M2S_PART_CONTIGUOUS / INTERLEAVED / ADAPTIVE as it ran */
  uint64_t* slabs;          /* optional, 3 * n_devices entries: x_begin, x_end, x_period of every shard as it ran (m2s_opts meaning) */
} m2s_multi_opts;
#define M2S_MULTI_OPTS_V1_SIZE 56
#define M2S_MULTI_OPTS_V2_SIZE 64
/* How the grid is cut.  CONTIGUOUS: shard k computes one x-slab.  INTERLEAVED: shard k computes the chunks k and n + k of 2n
 * (m2s_interleaved_slab): the cost of a layer varies along x — the deep interior of a body is the expensive part — and
 * contiguous slabs leave the middle shards with 30 % more work than the outer ones.  AUTO: interleaved for device-resident
 * results that are exchanged (PEER / RCCL) where the grid allows it, contiguous otherwise (host results stream out slab by slab;
 * with M2S_XCHG_NONE buffer k holds m2s_slab_bounds(nx, n, k)).  ADAPTIVE: contiguous slabs of equal COST — the library keeps,
 * per (grid, mesh size, sign method, shard count), the slab boundaries that m2s_balanced_slabs derives from the per-shard
 * times of the previous call; the first call uses even slabs.  For callers that repeat a call (a time loop): the cost of a
 * layer is not known in advance, but it can be measured.  `slabs` reports what ran. */
enum m2s_partition { M2S_PART_AUTO = 0, M2S_PART_CONTIGUOUS = 1, M2S_PART_INTERLEAVED = 2, M2S_PART_ADAPTIVE = 3 };
int m2s_generate_grid_sdf_multi(const float* vertices, size_t n_vertices, const void* indices, size_t n_indices,
                                int index_bytes, int topology, const m2s_grid* grid, int sign_method, float* const* outs,
                                const m2s_multi_opts* opts);
/* generate_sdf over several GPUs (lib.rs:291-300 is still the one signature a caller has).  Shard k computes the contiguous query
 * range m2s_slab_bounds(n_queries, n, k): every query depends only on the mesh.  mem_kind / devices / exchange / timings /
 * wall_ms / exchange_used as for the grid call; partition and peer_mode are ignored.
 *   M2S_MEM_HOST    outs[0] = the caller's array of n_queries floats; every device returns its range over its own PCIe link.
 *   M2S_MEM_DEVICE  vertices, indices and queries lie on devices[0]; outs[k] = n_queries floats on devices[k]; on return EVERY
 *                   buffer holds all distances (M2S_XCHG_PEER: each shard copies its finished range into the other buffers itself
 *                   over xGMI and the other devices read the inputs where they lie; M2S_XCHG_RCCL: inputs replicated by peer
 *                   copies, ranges gathered by ncclAllGather / ncclBroadcast; M2S_XCHG_NONE: buffer k holds range k only).
 * *n_out (optional): n_queries, or 0 for RTREE_BVH on a mesh without triangles, as m2s_generate_sdf. */
int m2s_generate_sdf_multi(const float* vertices, size_t n_vertices, const void* indices, size_t n_indices, int index_bytes,
                           int topology, const float* queries, size_t n_queries, int accel, int sign_method, float* const* outs,
                           size_t* n_out, const m2s_multi_opts* opts);
/* Contiguous range [x_begin, x_end) of shard k out of n over nx units (x-layers, queries): sizes differ by at most one. */
void m2s_slab_bounds(uint64_t nx, int n, int k, uint64_t* x_begin, uint64_t* x_end);
/* Slab boundaries of equal cost.  prev_bounds[0 .. n] (0 = prev_bounds[0] <= ... <= prev_bounds[n] = nx) are the contiguous slabs of a
 * previous call and cost[k] >= 0 what shard k spent on its slab (any unit; leave out what every shard repeats, e.g. the LBVH
 * build).  With the cost density taken as constant inside every previous slab, new_bounds[0 .. n] cuts [0, nx) into n slabs of
 * equal cost, every boundary a multiple of `unit` layers (a packet brick: 4 for cubic cells; 0 = 1) and every slab at least one unit
 * (even slabs when nothing was measured or there are fewer units than shards).  Host-only, deterministic: every rank of a
 * multi-process run computes the same boundaries from the same gathered costs. */
int m2s_balanced_slabs(uint64_t nx, int n, uint64_t unit, const uint64_t* prev_bounds, const float* cost, uint64_t* new_bounds);
/* The balanced partition: shard k takes the chunks k and n + k of 2n (m2s_opts.x_begin / x_end / x_period) when the grid allows
 * it — returns 1 — else its contiguous slab with *x_period = 0 — returns 0. */
int m2s_interleaved_slab(const m2s_grid* grid, int n, int k, uint64_t* x_begin, uint64_t* x_end, uint64_t* x_period);

/* What the links give: the copy kernel of M2S_PEER_PUSH (16 B per lane, 1 KiB per wave store instruction) writes the first n_cells
 * floats of `src` (on `device`; -1 = current) into each of the n_peers buffers (peer-mapped device memory, as m2s_opts.peer_out) —
 * first one peer at a time (gbps_each[k], optional), then all peers in one launch (*gbps_all, optional: the sum over the links).
 * GB/s of payload, best of three timed launches per figure.  xGMI is point-to-point, so gbps_each is a per-link number and
 * gbps_all / n_peers shows what a link keeps when all are busy: together they decide whether a slab's delivery hides under the
 * walk.  Blocks until done.  The first n_cells floats of every peer buffer ARE OVERWRITTEN with src's (the probe is
 * the push itself); the caller's current device is left as it was. */
int m2s_peer_bandwidth(const float* src, float* const* peers, uint32_t n_peers, size_t n_cells, int device, float* gbps_each, float* gbps_all);

/* One process per GPU (torch.distributed / MPI launchers): the same no-collective exchange across processes.
 * Every rank allocates its whole-grid buffer with m2s_shared_alloc (a dedicated hipMalloc block, so that its IPC handle
 * maps exactly this buffer), exports it, exchanges the 64-byte handles by any host-side means, opens the other ranks'
 * handles and passes the mapped pointers as m2s_opts.peer_out.  A host-side barrier after the local stream has drained
 * tells a rank that its own buffer is complete (the peers' pushes are ordinary device writes, finished when their
 * streams are). */
#define M2S_IPC_HANDLE_BYTES 64
int m2s_shared_alloc(size_t bytes, int device, void** device_ptr);
int m2s_shared_free(void* device_ptr, int device);
int m2s_ipc_export(const void* device_ptr, uint8_t handle[M2S_IPC_HANDLE_BYTES]);
int m2s_ipc_open(const uint8_t handle[M2S_IPC_HANDLE_BYTES], int device, void** device_ptr);
int m2s_ipc_close(void* device_ptr, int device);

/* ---- persistent mesh (optional) ------------------------------------------------------------------
 * The reference rebuilds its acceleration structures inside every call (generate/grid.rs:95-111,
 * generic/rtree_bvh.rs:109-119).  A caller that queries the same mesh repeatedly — the reference's
 * own client regenerates the grid on every parameter change, mesh_to_sdf_client/src/sdf.rs:32-137 —
 * or that splits one grid into several x-slab calls can build once and reuse:
 * triangle records + LBVH stay resident on `opts->device`; the sign planes of the last grid are cached.
 * Results are identical to the one-shot entry points.
 * A call may re-mark the resident tree's leaves for its own grid / query set (how many triangles a leaf holds follows the triangles
 * per brick, or the queries per triangle: a 5 us launch on the call's stream).  When that happens while earlier asynchronous
 * calls, or calls on another stream, may still be walking the tree, the call synchronises the device first; callers that keep
 * several streams busy on one mesh avoid it by giving them grids of the same density class (the x-slabs of one grid always are). */
typedef struct m2s_mesh m2s_mesh;
int m2s_mesh_create(const float* vertices, size_t n_vertices, const void* indices, size_t n_indices, int index_bytes,
                    int topology, const m2s_opts* opts, m2s_mesh** out_mesh);
void m2s_mesh_destroy(m2s_mesh* mesh);
size_t m2s_mesh_triangle_count(const m2s_mesh* mesh);
int m2s_mesh_generate_grid_sdf(m2s_mesh* mesh, const m2s_grid* grid, int sign_method, float* out, const m2s_opts* opts);
int m2s_mesh_generate_sdf(m2s_mesh* mesh, const float* queries, size_t n_queries, int accel, int sign_method, float* out,
                          size_t* n_out, const m2s_opts* opts);
/* Device-memory calls made with opts->synchronous == 0 return before the GPU has finished and report no
 * per-call timings; this sums the dominant-kernel durations of all such calls since the last drain
 * (distance_ms, distance_launches, n_units; accel_build_ms = the mesh build).  Blocks until they finished.
 * Asynchronous calls also DEFER their device-side error report to this call: M2S_ERR_NAN if any of them met a
 * NaN distance in SignMethod::Normal (the reference panics, lib.rs:257). */
int m2s_mesh_drain_timings(m2s_mesh* mesh, m2s_timings* timings);

/* Grid helpers with the reference's exact f32 arithmetic (so callers need not re-derive it).
 * m2s_grid_from_bounding_box — Grid::from_bounding_box, grid.rs:59-74.
 * m2s_grid_cell_center      — Grid::get_cell_center,   grid.rs:135-141.
 * m2s_grid_cell_idx         — Grid::get_cell_idx,      grid.rs:122-124. */
void m2s_grid_from_bounding_box(const float bbox_min[3], const float bbox_max[3], const uint64_t cell_count[3],
                                m2s_grid* grid);
void m2s_grid_cell_center(const m2s_grid* grid, const uint64_t cell[3], float out[3]);
uint64_t m2s_grid_cell_idx(const m2s_grid* grid, const uint64_t cell[3]);

/* Number of triangles Topology::get_triangles (lib.rs:175-193) yields for these arguments. */
size_t m2s_triangle_count(size_t n_vertices, size_t n_indices, int has_indices, int topology);

/* ---- the data formats and callers either side of the path -------------------
 *
 * V1 container — mesh_to_sdf/src/serde.rs:75-221.  `save_to_file` writes rmp-serde's compact MessagePack
 * encoding of SerializeVersion::V1(SerializeSdf::{Generic,Grid}) (serde.rs:161-166): enums are one-entry
 * maps keyed by the variant name, structs are arrays, f32 is `ca` + 4 big-endian bytes, usize is the
 * shortest unsigned form, a point is `93 ca.. ca.. ca..`:
 *   Grid    81 a2 "V1" 81 a4 "Grid"    92 [93 first_cell(3 f32) cell_size(3 f32) 93 cell_count(3 uint)] [array n: f32...]
 *   Generic 81 a2 "V1" 81 a7 "Generic" 92 [array nq: point...] [array nd: f32...]
 * Pinned byte for byte on the reference's golden files tests/sdf_grid_v1.bin and tests/sdf_generic_v1.bin
 * (serde.rs:314-374).  The payload arrays (5 B per distance, 16 B per point) are produced / consumed by
 * HIP kernels so a device-resident result is encoded without a round trip through host f32 arrays.
 * `distances` / `queries` / `bytes` follow opts->mem_kind like every other data pointer. */
enum m2s_sdf_kind { M2S_SDF_GENERIC = 0, M2S_SDF_GRID = 1 };

typedef struct m2s_sdf_info {
  int32_t kind;              /* enum m2s_sdf_kind */
  int32_t canonical;         /* 1: both arrays use exactly the fixed-width encoding above (decoded by the HIP kernels);
                                0: some element uses another valid MessagePack number form (f64, ints) that serde would
                                   accept for an f32 — decoded by the scalar host reader */
  m2s_grid grid;             /* kind == M2S_SDF_GRID */
  uint64_t n_queries;        /* kind == M2S_SDF_GENERIC, else 0 */
  uint64_t n_distances;
  uint64_t queries_offset;   /* byte offset of the first element of each array (canonical == 1) */
  uint64_t distances_offset;
} m2s_sdf_info;

/* Exact size in bytes of the container; 0 if a count does not fit MessagePack's 32-bit array header. */
size_t m2s_sdf_grid_encoded_size(const m2s_grid* grid, size_t n_distances);
size_t m2s_sdf_generic_encoded_size(size_t n_queries, size_t n_distances);

/* serialize(&SerializeSdf::Grid(..)) — serde.rs:99-107,161-166.  n_distances is NOT required to equal the
 * grid's cell count (the reference does not check either).  *written (optional) receives the size. */
int m2s_sdf_encode_grid(const m2s_grid* grid, const float* distances, size_t n_distances, uint8_t* bytes,
                        size_t capacity, size_t* written, const m2s_opts* opts);
/* serialize(&SerializeSdf::Generic(..)) — serde.rs:87-95,161-166.  queries: n_queries packed xyz. */
int m2s_sdf_encode_generic(const float* queries, size_t n_queries, const float* distances, size_t n_distances,
                           uint8_t* bytes, size_t capacity, size_t* written, const m2s_opts* opts);

/* First half of deserialize() — serde.rs:169-176: reads the envelope and the array headers so the caller
 * can size its buffers.  M2S_ERR_BAD_ARG = SerdeError::DeserializationFailed. */
int m2s_sdf_probe(const uint8_t* bytes, size_t n_bytes, m2s_sdf_info* info, const m2s_opts* opts);
/* Second half: fills queries_out (n_queries*3 f32; may be NULL for a grid container) and distances_out
 * (n_distances f32).  Always synchronous (a malformed element is reported by the return code). */
int m2s_sdf_decode(const uint8_t* bytes, size_t n_bytes, float* queries_out, float* distances_out,
                   const m2s_opts* opts);

/* save_to_file / read_from_file — serde.rs:192-198, 216-220.  M2S_ERR_IO = SerdeError::IoError. */
#define M2S_ERR_IO (-5)
int m2s_sdf_save_grid(const char* path, const m2s_grid* grid, const float* distances, size_t n_distances,
                      const m2s_opts* opts);
int m2s_sdf_save_generic(const char* path, const float* queries, size_t n_queries, const float* distances,
                         size_t n_distances, const m2s_opts* opts);
int m2s_sdf_probe_file(const char* path, m2s_sdf_info* info);
int m2s_sdf_read_file(const char* path, float* queries_out, float* distances_out, const m2s_opts* opts);

/* Client post-step on a generated grid — mesh_to_sdf_client/src/sdf.rs:62-72 and :120:
 *   ordered_indices = (0..n).sorted_by(|i, j| data[i].total_cmp(&data[j])).map(|i| i as u32)   (stable sort)
 *   iso_limits      = data.iter().copied().minmax()                                            (itertools)
 * ordered_indices: n u32, same side as `distances` (opts->mem_kind).  iso_limits: 2 floats on the HOST, optional;
 * first of equal minima / last of equal maxima as itertools; NaNs (never produced by the generators) are skipped,
 * (NaN, NaN) if nothing is comparable.  n must be < 2^32 (the reference's u32 cast).  Always synchronous when
 * iso_limits is requested. */
int m2s_order_cells_by_distance(const float* distances, size_t n, uint32_t* ordered_indices, float* iso_limits,
                                const m2s_opts* opts);

/* Client pre-step — mesh_to_sdf_client/src/sdf_program.rs:607-632: a glTF scene holds several model instances;
 * the client merges them into the single vertex / index buffer the generator takes:
 *   vertices.extend(model.vertices.map(|v| transform.transform_point3(v.position)))   glam Mat4 (no FMA)
 *   indices.extend(model.indices.map(|i| i + len as u32))                             len = vertices so far
 * and takes the per-axis minmax of the merged vertices as the mesh bounding box.
 * `vertices`/`indices` of every instance and the outputs follow opts->mem_kind; the instance table itself and
 * `bbox` ({xmin,ymin,zmin,xmax,ymax,zmax}, optional) are host memory.  Outputs hold sum(n_vertices) xyz and
 * sum(n_indices) u32. */
typedef struct m2s_instance {
  const void* vertices;        /* first position (3 consecutive f32) */
  size_t n_vertices;
  size_t vertex_stride;        /* bytes between positions; 0 = 12 (packed).  The client's Vertex carries more attributes */
  const uint32_t* indices;
  size_t n_indices;
  float transform[16];         /* glam::Mat4, column-major: x_axis, y_axis, z_axis, w_axis */
} m2s_instance;
int m2s_merge_instances(const m2s_instance* instances, size_t n_instances, float* vertices_out, uint32_t* indices_out,
                        float* bbox, const m2s_opts* opts);

/* glTF 2.0 / GLB ingestion (host side) — what the reference client extracts from a file before the merge:
 * mesh_to_sdf_client/src/gltf/mod.rs:56-174 (models keyed by mesh index, one primitive per mesh survives;
 * POSITION + indices, sparse accessors and byteStride honoured; missing indices = 0..n, pbr/model.rs:29-32),
 * gltf/scene/mod.rs:56-160 (node tree, simplify_tree) and gltf/mod.rs:91-106 (flatten_hierarchy: children
 * first, world = parent * local in glam's f32 arithmetic).  Instances come out in that order, over all scenes.
 * Errors: M2S_ERR_IO (cannot read), M2S_ERR_BAD_ARG (invalid file, or a required extension other than
 * KHR_lights_punctual — the reference's gltf build rejects those, gltf/mod.rs:407-410). */
typedef struct m2s_gltf m2s_gltf;
typedef struct m2s_gltf_info {
  uint64_t n_scenes, n_models, n_instances;
  uint64_t n_vertices, n_indices;   /* summed over the instances = sizes of the merged buffers */
} m2s_gltf_info;
int m2s_gltf_open(const char* path, m2s_gltf** out, m2s_gltf_info* info);
/* Fills `capacity` >= n_instances entries with HOST pointers into the handle (valid until m2s_gltf_close);
 * pass them to m2s_merge_instances with opts->mem_kind == M2S_MEM_HOST. */
int m2s_gltf_instances(const m2s_gltf* gltf, m2s_instance* out, size_t capacity);
void m2s_gltf_close(m2s_gltf* gltf);

/* Library / device introspection. */
/* The one-off costs of a process's first call, paid now instead: the HIP runtime and this library's code objects on `device`
 * (-1 = the current one), the streams and events of the device's context, `workspace_bytes` of device workspace (0 = none: it
 * grows on demand; a 512^3 grid call needs ~0.4 GB, 10 M queries ~1.5 GB) and, for callers of the host-pointer entry points, the
 * pinned staging ring (`host_ring_bytes` > 0: three slots of min(host_ring_bytes, 64 MB)).  Optional, idempotent (a repeated call
 * touches only what has grown since), blocking.  The workspace it grows is the one of the library's OWN stream: a caller that passes
 * m2s_opts.stream (or stream_mode 1) works out of a block per stream, which still grows, and is first touched, in that caller's first call.
 * The reference's usage is one call per process (examples/demo.rs:29-54): without this the first call pays 20-40 ms (grid,
 * host result) to ~0.9 s (10 M queries through host pointers) on top of its steady-state time — INTEGRATION.md has the breakdown;
 * M2S_HOST_TIMES=1 prints it per call on stderr. */
int m2s_warmup(int device, size_t workspace_bytes, size_t host_ring_bytes);
int m2s_version(void);               /* major*1000 + minor */
int m2s_device_count(void);          /* HIP devices visible; 0 if none (every compute call then fails with M2S_ERR_HIP) */
const char* m2s_last_error(void);    /* thread-local message of the last failing call on this thread */
/* Drops the cached per-device workspace (device memory is otherwise kept between calls). */
void m2s_release_workspace(void);
/* Run-time knobs (mesh_to_sdf_amd/csrc/tuning.h lists them; DESIGN.md §9 says what each default rests on).  They are read from the
 * environment variables of the same names ONCE, at the library's first use; m2s_tuning_set changes one afterwards (value NULL or "" =
 * back to the default) and returns M2S_ERR_BAD_ARG for an unknown name or an unparsable value.  None of them changes a result — they
 * choose between code paths that produce the same bits, which is what the tests use them for.  Not synchronised with calls in flight
 * on other threads.  m2s_tuning_describe writes "NAME=value" lines (NUL-terminated, truncated to `capacity`) and returns the length
 * the full text needs. */
int m2s_tuning_set(const char* name, const char* value);
int m2s_tuning_describe(char* buffer, int capacity);

#ifdef __cplusplus
}
#endif
#endif /* M2S_H */
