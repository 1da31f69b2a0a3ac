// common.h — records shared by the host orchestration and the gfx950 kernels.
#pragma once
#include <functional>
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace m2s {

// ---- HBM layout --------------------------------------------------------------------------
// Triangle record, 96 B, stored in Morton (= BVH leaf) order.  Every lane of a
// wave tests the SAME triangle, so a record is fetched with wave-uniform (scalar) loads:
// a few s_load_dwordx4/x8 per triangle; the record is therefore packed per triangle, while
// everything that is read lane-parallel (queries, outputs, bit planes) is SoA / linear.
struct alignas(32) TriRec {
  float ax, ay, az;
  uint32_t cls;        // geo.hip.h tri_class (degeneracy class of geo.rs:73-88)
  float bx, by, bz;
  uint32_t index;      // triangle index in Topology::get_triangles order (tie-break for Rtree)
  float cx, cy, cz;
  float pad0;
  // edge vectors exactly as geo.rs computes them per call (b.sub(a), c.sub(a), c.sub(b)): they are
  // the same for all 64 lanes, so they are formed once at build time instead of on the VALU
  // the 4th components carry the raw normal ab x ac (geo.rs:60-64), same arithmetic as the reference, formed once
  float abx, aby, abz, nrx;
  float acx, acy, acz, nry;
  float bcx, bcy, bcz, nrz;
};
static_assert(sizeof(TriRec) == 96, "TriRec must be 96 bytes");

// Stackless BVH node, 32 B, pre-order (depth-first) layout:
//   the left child of node i is i+1; `skip` is the next node once the subtree of i is done
//   (== node count at the right spine), so traversal is   i = hit && !leaf ? i + 1 : skip.
// Leaves hold one triangle: the padded box of geo.rs:4-22 (1e-4), internal nodes the union.
struct alignas(32) NodeRec {
  float mnx, mny, mnz;
  uint32_t skip;
  float mxx, mxy, mxz;
  int32_t tri;         // >= 0: leaf, index into the TriRec array; -1: internal
};
static_assert(sizeof(NodeRec) == 32, "NodeRec must be 32 bytes");

// Oriented bound of a node, 48 B, same index as NodeRec.  Every triangle of the subtree lies inside
//   { x : dlo <= n.(x - c) <= dhi  and  |(x - c) - n (n.(x - c))| <= R }      (a disc-shaped slab)
// for the unit vector n (area-weighted mean normal; ANY unit vector keeps the bound valid).
// For a smooth surface patch the slab is thin, which is what an AABB cannot express: a query at
// distance D from a flat patch meets ~rho*pi*2*s*D leaf AABBs of thickness s but only the O(1)
// discs that cover its foot point.  lower bound^2 = max(dlo - t, t - dhi, 0)^2 + max(l - R, 0)^2
// (stored as mid = (dlo+dhi)/2, half = (dhi-dlo)/2 rounded up: the slab term is max(|t - mid| - half, 0))
// with t = n.(p - c), l^2 = |p - c|^2 - t^2.
// The record also repeats the node's `skip` and `tri`, so the nearest-triangle traversal reads
// ONLY this array (one 48-byte scalar load per node); NodeRec boxes serve the ray stabbing.
// Nodes with more than EXT_TRIVIAL_ABOVE triangles get the cylinder around their AABB instead
// (n = +x): at that size a surface patch is not flat and the loop over its triangles is not worth it.
struct alignas(16) NodeExt {
  float cx, cy, cz, R;
  float nx, ny, nz, mid;   // slab along n: |n.(p-c) - mid| <= half
  float half;
  uint32_t skip;
  int32_t tri;
  uint32_t pad;
};
constexpr uint32_t EXT_TRIVIAL_ABOVE = 2048;
static_assert(sizeof(NodeExt) == 48, "NodeExt must be 48 bytes");

// Leaf pre-test record, 64 B, same index as TriRec: unit normal n with offset dn = n.a, and for each edge the
// unit in-plane outward normal m_k with offset o_k (rounded outwards), so that for any point p
//   lb^2 = (n.p - dn)^2 + max(m_0.p - o_0, m_1.p - o_1, m_2.p - o_2, 0)^2  <=  dist(p, triangle)^2
// (equality in the face and edge regions, a slight underestimate in the vertex regions).  Much tighter than
// the leaf's disc: a lane needs the 140-instruction exact evaluation only if this bound reaches its best.
// Degenerate triangles carry an all-zero record (bound 0: always evaluated).
struct alignas(64) TriPlanes {
  float nx, ny, nz, dn;
  float m0x, m0y, m0z, o0;
  float m1x, m1y, m1z, o1;
  float m2x, m2y, m2z, o2;
};
static_assert(sizeof(TriPlanes) == 64, "TriPlanes must be 64 bytes");

struct DeviceMesh {
  const TriRec* tris;   // n_tris records, Morton order
  const float4* corners;    // n_tris x 3: the vertices alone, 48 B per triangle ((a, b.x) (b.yz, c.xy) (c.z, -, -, -)) — what the ray walks read
  const TriPlanes* planes;  // n_tris leaf pre-test records
  const float4* cen;    // n_tris triangle centroids (same order), for the jump-flooding seed pass
  const float4* cen_raw;     // the same centroids in INPUT triangle order (available before the sort)
  const uint32_t* slot_of;   // input triangle -> slot in the sorted arrays
  const NodeRec* nodes; // n_nodes = 2*n_tris - 1 (0 if n_tris == 0)
  const NodeExt* ext;   // n_nodes oriented bounds
  unsigned long long* stats;  // optional traversal counters (M2S_STATS=1), else nullptr
  uint32_t n_tris;
  uint32_t n_nodes;
  uint32_t leaf_max;    // subtrees of at most this many triangles are walked as one leaf (a grid call chooses by its grid: grid_leaf_max; set_leaf_size re-marks a resident tree)
  const uint32_t* slot_first;   // pre-order slot -> first triangle of its subtree (what a collapsed leaf's `tri` holds)
  const int* scene;     // 6 order-encoded ints: min xyz / max xyz of the triangle box centres (see bvh.hip)
};

// Largest finite |coordinate| of the mesh's triangle-box centres; feeds the pruning slack.  k_scene_reduce
// (bvh.hip) leaves it as float bits in scene[6], so a walk pays one wave-uniform load for it.
__device__ __forceinline__ float mesh_scale(const DeviceMesh& m) { return __int_as_float(m.scene[6]); }

struct GridParams {
  float first[3];
  float size[3];
  uint32_t n[3];
  uint32_t xb, xe;   // x-slab [xb, xe)
  uint32_t nzw;      // 32-bit words per (x,y) row of a bit plane = ceil(nz / 32)
  uint64_t out_off;  // subtracted from the whole-grid cell index when writing (slab staging buffers)
  // log2 extents of the 64-voxel packet brick along x, y, z (sum 6; 2,2,2 = 4x4x4).  Chosen per grid so that the
  // brick is as close to a cube in WORLD space as powers of two allow (anisotropic cell sizes): the walk's cost
  // grows with the brick's diameter, not its voxel count.
  uint32_t bl[3];
  // ceil(2^32 / d) for d = super-bricks along z and along y (distance.hip brick_coords divides a packet's super-brick
  // index by them: one multiply-high instead of a 30-instruction u32 division per wave); 0 = divide (d == 1, or a grid so
  // large that the product could be off by one)
  uint32_t sz_magic, sy_magic;
  uint32_t xl_cap;   // 0: super-bricks up to 8 bricks wide in x (distance.hip super_brick_xlog); k: at most 2^(k-1) bricks wide, so
                     // that the x-layers of a slab are finished in order at that granularity (M2S_PEER_TRAIL)
  // Interleaved slab (m2s_opts.x_period): the call owns the chunks [xb + j * period, xb + j * period + 2^chunk_log), j = 0, 1, ...
  // [xb, xe) then is the VIRTUAL slab — the chunks laid end to end — which is what bricks, seeds and cut lists are numbered
  // by; slab_x() turns a virtual layer into the grid's x.  chunk_log = 31: one contiguous slab (slab_x(v) = xb + v).
  uint32_t chunk_log, period;
};
// Grid x of virtual layer `v` (0-based) of the slab.
__host__ __device__ __forceinline__ uint32_t slab_x(const GridParams& g, uint32_t v) {
  return g.xb + (v >> g.chunk_log) * g.period + (v & ((1u << g.chunk_log) - 1u));
}

// Brick shape for a cell size: minimises max/min of the world extents |size[k]| * 2^bl[k] over all splits of 6.
// Exact for n * d < 2^32 (Granlund-Montgomery with a 32-bit multiplier): q = mulhi(n, ceil(2^32 / d)).
inline void set_super_brick_magic(GridParams& g) {
  const uint64_t nbx = (g.n[0] + (1u << g.bl[0]) - 1u) >> g.bl[0], nby = (g.n[1] + (1u << g.bl[1]) - 1u) >> g.bl[1],
                 nbz = (g.n[2] + (1u << g.bl[2]) - 1u) >> g.bl[2];
  const uint64_t sy = (nby + 7) >> 3, sz = (nbz + 7) >> 3, n_max = (nbx + 1) * sy * sz;   // super-brick indices stay below this
  const uint64_t d_max = sy > sz ? sy : sz;
  const bool safe = n_max * d_max < (1ull << 32);
  g.sz_magic = (safe && sz > 1) ? (uint32_t)(((1ull << 32) + sz - 1) / sz) : 0u;
  g.sy_magic = (safe && sy > 1) ? (uint32_t)(((1ull << 32) + sy - 1) / sy) : 0u;
}

inline void choose_brick_shape(const float size[3], uint32_t bl[3]) {
  bl[0] = bl[1] = bl[2] = 2;
  float s[3];
  for (int k = 0; k < 3; ++k) {
    s[k] = size[k] < 0 ? -size[k] : size[k];
    if (!(s[k] > 0.0f) || !(s[k] < 3.0e38f)) return;
  }
  float best = -1.0f;
  for (uint32_t a = 0; a <= 6; ++a)
    for (uint32_t b = 0; a + b <= 6; ++b) {
      const uint32_t c = 6 - a - b;
      const float e[3] = {s[0] * (float)(1u << a), s[1] * (float)(1u << b), s[2] * (float)(1u << c)};
      const float mx = e[0] > e[1] ? (e[0] > e[2] ? e[0] : e[2]) : (e[1] > e[2] ? e[1] : e[2]);
      const float mn = e[0] < e[1] ? (e[0] < e[2] ? e[0] : e[2]) : (e[1] < e[2] ? e[1] : e[2]);
      const float ratio = mx / mn;
      const bool cube = a == 2 && b == 2;
      if (best < 0.0f || ratio < best * 0.999f || (cube && ratio <= best * 1.001f)) { best = ratio; bl[0] = a; bl[1] = b; bl[2] = c; }
    }
}

// Additional whole-grid output buffers (m2s_opts.peer_out): same layout and indexing as `out`, usually on other devices.
constexpr uint32_t MAX_PEERS = 15;
struct PeerOut {
  float* p[MAX_PEERS];
  uint32_t n;
  // M2S_PEER_TRAIL: the walk counts finished packets per x-unit of 2^unit_log bricks (progress[unit], device scope release)
  // and a copy kernel that runs beside it pushes every unit to the peers as soon as it is complete
  uint32_t unit_log;
  uint32_t rows;       // brick rows (bricks along y) = row counters per unit
  uint32_t units;
  uint32_t* progress;  // [0, units): finished rows per unit;  [units + unit * rows + brick row]: finished packets of that row
};

// Device-side error flags (OR-ed into one int by kernels).
enum : int { ERRF_INDEX_OOB = 1, ERRF_NAN = 2, ERRF_TRAIL_TIMEOUT = 4, ERRF_BUILD_TIMEOUT = 8, ERRF_SPLIT_OVERFLOW = 16 };

// Result modes of the nearest search.
enum : int {
  MODE_UNSIGNED = 0,      // min unsigned distance (Raycast magnitude)                       default.rs:44-51
  MODE_NORMAL_FOLD = 1,   // compare_distances fold (None/Bvh + Normal, grid Normal)         default.rs:52-59
  MODE_NEAREST_NORMAL = 2 // sign of the single nearest triangle (Rtree)                     rtree.rs:113-125
};
// Sign sources for MODE_UNSIGNED.
enum : int {
  SIGN_NONE = 0,
  SIGN_GRID_PLANE = 1,  // grid path: majority bit plane from sign.hip
  SIGN_RAYS3 = 2,       // generic: best of three axis rays from the query, BVH candidate rule
  SIGN_XRAY_ALL = 3     // generic None(Raycast): +X ray against ALL triangles (brute force only)
};

#define M2S_HIP_CHECK(expr)                                                   \
  do {                                                                        \
    hipError_t _e = (expr);                                                   \
    if (_e != hipSuccess) {                                                   \
      ::m2s::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return M2S_ERR_HIP_INTERNAL;                                            \
    }                                                                         \
  } while (0)

constexpr int M2S_ERR_HIP_INTERNAL = -4;
void set_error(const char* fmt, ...);

// Bump allocator over one device block per device (kept between calls; see capi.hip).
struct Arena {
  char* base = nullptr;
  size_t cap = 0;
  size_t off = 0;
  template <class T>
  T* take(size_t count, size_t align = 256) {
    size_t o = (off + align - 1) / align * align;
    size_t bytes = count * sizeof(T);
    if (o + bytes > cap) return nullptr;
    off = o + bytes;
    return reinterpret_cast<T*>(base + o);
  }
};

// sortlib.hip: the rocPRIM calls (their device code lives in that translation unit alone; tmp == nullptr: the size query)
hipError_t sort_pairs_u64(void* tmp, size_t& bytes, const uint64_t* keys_in, uint64_t* keys_out, const uint32_t* vals_in, uint32_t* vals_out, size_t n,
                          unsigned begin_bit, unsigned end_bit, hipStream_t st);
hipError_t sort_pairs_u32(void* tmp, size_t& bytes, const uint32_t* keys_in, uint32_t* keys_out, const uint32_t* vals_in, uint32_t* vals_out, size_t n,
                          unsigned begin_bit, unsigned end_bit, hipStream_t st);
hipError_t select_flagged_indices(void* tmp, size_t& bytes, const uint8_t* flags, uint32_t* out, uint32_t* count, size_t n, hipStream_t st);
// ---- host launchers implemented in the .hip files ------------------------------------------
// one empty kernel per translation unit (m2s_warmup)
void warm_bvh(hipStream_t st);
void warm_sign(hipStream_t st);
void warm_distance(hipStream_t st);
void warm_serde(hipStream_t st);
void warm_client(hipStream_t st);
void warm_sortlib(hipStream_t st);
void warm_sortlib_query(hipStream_t st);
// bvh.hip: flatten topology, build triangle records + LBVH in pre-order layout.
size_t bvh_workspace_bytes(size_t n_tris);
// `after_setup` (optional) is called twice with the input-order centroid array and triangle records: with phase 0 once the kernels that fill
// it are enqueued on `st` (record an event there), and with phase 1 once the keys, the sort and the hierarchy are enqueued
// (enqueue the side work — the seed passes — on another stream behind that event; bvh.hip says why there).
int build_device_mesh(Arena& ws, hipStream_t st, const float* d_verts, size_t n_verts, const void* d_indices,
                      size_t n_indices, int index_bytes, int topology, size_t n_tris, int* d_err, DeviceMesh* out,
                      const std::function<int(const float4*, const TriRec*, int)>* after_setup = nullptr, bool records_only = false,
                      uint32_t leaf_max = 2, uint64_t job_cells = ~0ull);   // job_cells: the cells a one-shot grid call is about to walk (unknown: all ones)
bool query_walk_is_lane(size_t n_q, size_t n_tris, int sign_src);       // distance.hip: sparse query sets take the lane walk
uint32_t query_leaf_max(size_t n_q, size_t n_tris, int sign_src);       // ... and the leaf size a query call wants its tree to have
uint32_t grid_leaf_max(const GridParams& g, size_t n_tris);   // distance.hip: the leaf size a grid call wants its tree to have
// Re-marks the leaves of a resident tree (persistent meshes): the node records of a subtree of at most `leaf_max` triangles get `tri` = its first
// triangle, all others -1 — what k_emit would have written.  One small launch on `st`; nothing else of the tree depends on the leaf size.
int set_leaf_size(hipStream_t st, DeviceMesh* mesh, uint32_t leaf_max);

// sign.hip: grid-line ray parity -> one "inside" bit per voxel (bit plane in grid layout).
size_t sign_workspace_bytes(const GridParams& g, size_t n_tris);
// slab_only: the planes of the x-layers [g.xb, g.xe) (virtual slab) only — what a one-shot slab call reads; false: the whole grid
int build_grid_sign_plane(Arena& ws, hipStream_t st, const DeviceMesh& mesh, const GridParams& g,
                          const uint32_t** d_inside_plane, bool slab_only);

// distance.hip
size_t grid_distance_workspace_bytes(const GridParams& g, size_t n_tris);
// Split walk (distance.hip): a packet still walking when the launch runs dry hands the rest of its pre-order ranges to other waves.
// `cnt`: [0] suspended packets (= accumulator slots taken), [1 + r] items in the list of follow-up round r (r = 1 ..),
// [16 + x] the time (10 ns ticks, made odd) at which XCD x was handed its last packet, [24 + x] the time it was handed its first,
// [16 + 16 r + x] the "a wave of XCD x has run out of items" flag of follow-up round r; all cleared per launch by k_split_init.
constexpr uint32_t SPLIT_MAX_ROUNDS = 6, SPLIT_CNT_WORDS = 16 + 16 * (SPLIT_MAX_ROUNDS + 1);
struct SplitCtl {
  uint32_t* cnt = nullptr;          // nullptr: no splitting
  uint32_t* slot_packet = nullptr;  // accumulator slot -> packet (0xffffffff: the packet finished by itself after all)
  uint32_t* acc = nullptr;          // per slot: 64 x d2 bits, then (Normal fold) 64 x d2pos bits — merged with atomic minima
  uint4* items = nullptr;           // lists of the follow-up rounds, cap_items each: (packet, first byte, end byte, slot)
  uint32_t cap_slots = 0, cap_items = 0;
  uint32_t grace = 0;               // work units a walk does before it first looks at the flag (rounds: and between flag and suspension)
  uint32_t patience_q8 = 0;         // k_packet: (ordinary packet times a walk may outlast the flag) / (rounds before the flag), in 1/256
  uint32_t idle_below = 0;          // 1: forced — the stamps are all time 0 (flags up, no patience)
  uint32_t emit_min = 0, emit_max = 0;   // bytes of records: a suspended walk hands over the surviving subtrees of this size range
  uint32_t rounds = 3;              // follow-up launches; the last one walks to the end
};
// What a grid walk needs besides the mesh: the seed lattice and the cut lists of a slab (device pointers into the call's arena).
struct GridWalkPlan {
  const uint32_t* seeds = nullptr;
  uint32_t seed_shift = 0, seed_ny = 0, seed_nz = 0;
  const uint32_t* cut_lists = nullptr;
  uint32_t cut_log = 0, cut_ny = 0, cut_nz = 0;
  bool lane_walk = false;
  SplitCtl split;                  // packet walk only
  bool split_forced = false;       // M2S_SPLIT=2: the flags start raised (tests)
  int defer = 0;                   // packet walk: 1 exact evaluations queued and run densely, 2 + direct where most lanes are reached (distance.hip DeferQueue)
  const uint2* group_top = nullptr;   // packet groups (k_packet_group): the tree's top subtrees (k_tree_top); nullptr: one wave per packet
  uint32_t group_waves = 0;           // ... waves per packet (2, 4, 8 or 16)
  uint32_t* brute_acc = nullptr;   // tiny problems (grid_is_tiny): per-voxel minima of k_brute_split; no seeds, no lists, no tree
};
bool grid_is_tiny(const GridParams& g, size_t n_tris, int algorithm, bool raycast);   // raycast: the call's sign rule (the limits differ)
// Seed lattice of a slab: one triangle id per packet brick (ids index the centroid array it was computed from).
struct SeedLattice {
  uint32_t* ids = nullptr;
  uint32_t ny = 0, nz = 0;
  size_t points = 0;
  uint32_t shift = 0;   // 2^shift packet bricks per axis share one lattice point
};
bool grid_walk_wants_seeds(const GridParams& g, size_t n_tris, int algorithm);
uint32_t host_packet_bricks(const GridParams& g);   // packet bricks of the slab (= points of its seed lattice), padded to super-bricks
int launch_grid_seeds(Arena& ws, hipStream_t st, const float4* cen, uint32_t n_tris, const GridParams& g, SeedLattice* out);
int prepare_grid_walk(Arena& ws, hipStream_t st, const DeviceMesh& mesh, const GridParams& g, int algorithm, bool pipelined,
                      GridWalkPlan* plan, const SeedLattice* raw_seeds = nullptr);
// `g` may be an x-piece of the slab the plan was prepared for, starting bx_off bricks into it (a multiple of 2 bricks).
// `peers` (optional): buffers that receive the same values in the walk's epilogue (M2S_PEER_STORE).
int launch_grid_walk(hipStream_t st, const DeviceMesh& mesh, const GridParams& g, int mode, const uint32_t* d_inside_plane,
                     int algorithm, const GridWalkPlan& plan, uint32_t bx_off, float* d_out, int* d_err,
                     const PeerOut* peers = nullptr);
// M2S_PEER_PUSH: copies the cells [first, first + count) of the whole-grid buffer `src` to the same range of every peer
// (16 B per lane where the range allows).
int launch_push_cells(hipStream_t st, const float* src, const PeerOut& peers, uint64_t first, uint64_t count);
// M2S_PEER_TRAIL: bricks per progress unit (log2), number of units of the slab, and the copy kernel that trails the walk.
uint32_t trail_unit_log(const GridParams& g);
uint32_t trail_units(const GridParams& g);
uint32_t trail_rows(const GridParams& g);
int launch_push_trailing(hipStream_t st, const float* src, const PeerOut& peers, const GridParams& g, int* d_err);
// Records `ev_before_final` (if non-null) between the seed passes and the final k_packet launch.
int launch_grid_distance(Arena& ws, hipStream_t st, const DeviceMesh& mesh, const GridParams& g, int mode,
                         const uint32_t* d_inside_plane, int algorithm, float* d_out, int* d_err,
                         hipEvent_t ev_before_final, hipEvent_t wait_before_final = nullptr, bool pipelined = false,
                         const SeedLattice* raw_seeds = nullptr, hipEvent_t wait_raw_seeds = nullptr,
                         const PeerOut* peers = nullptr);
void cut_word_roundtrip(uint32_t n_nodes, uint32_t start, uint32_t len, uint32_t* word, uint32_t* first, uint32_t* end);   // test hook
size_t query_workspace_bytes(size_t n_q);
bool query_is_tiny(size_t n_q, size_t n_tris, int algorithm, int sign_src);   // small query sets: all queries x all triangles, no tree
int launch_query_brute_split(Arena& ws, hipStream_t st, const DeviceMesh& mesh, const float* d_queries, size_t n_q, int mode, int sign_src,
                             float* d_out, int* d_err);
// What the walk of a generic query set needs besides the mesh (prepare_query_walk; device pointers into the call's arena).
struct QueryPlan {
  size_t n_q = 0;
  const int* qb = nullptr;          // ordered-int bounding box of the queries
  const uint32_t* perm = nullptr;   // sorted position -> input index
  const float4* sorted = nullptr;   // queries in Morton order
  const uint32_t* table = nullptr;  // packet table (k_qcells), nullptr: 64 consecutive queries per packet
  const float4* centres = nullptr;  // (centre, radius) per packet when cut lists will be used
  const GridParams* lat = nullptr;  // seed lattice description (QL^3 cells over the queries' bounding box) when seeds will be used
  uint32_t launched = 0;
  bool lane_walk = false, seeds = false;
};
struct QuerySeeds {
  uint32_t* ids = nullptr;          // one triangle per lattice cell
  bool raw = false;                 // ids name input triangles (computed while the mesh was being built): translate through slot_of
};
// `after_lattice` (optional): recorded on `st` once the bounding box and the lattice description are enqueued (what launch_query_seeds needs)
int prepare_query_walk(Arena& ws, hipStream_t st, const float* d_queries, size_t n_q, size_t n_tris, int sign_src, int algorithm, QueryPlan* plan,
                       hipEvent_t after_lattice = nullptr);
int launch_query_seeds(Arena& ws, hipStream_t st, const float4* cen, uint32_t n_tris, const QueryPlan& plan, bool raw, QuerySeeds* out);
int launch_query_walk(Arena& ws, hipStream_t st, const DeviceMesh& mesh, const float* d_queries, const QueryPlan& plan,
                      int mode, int sign_src, int algorithm, float* d_out, int* d_err, const QuerySeeds* pre = nullptr);
int launch_query_distance(Arena& ws, hipStream_t st, const DeviceMesh& mesh, const float* d_queries, size_t n_q,
                          int mode, int sign_src, int algorithm, float* d_out, int* d_err);


// serde.hip: payload arrays of the V1 container (fixed-width MessagePack records), any byte alignment.
struct wire_bytes { uint8_t b[120]; uint32_t n; };   // envelope bytes passed by value to a 1-block kernel
int launch_write_bytes(hipStream_t st, uint8_t* d_dst, const uint8_t* bytes, uint32_t n);
int launch_encode_f32(hipStream_t st, const float* d_values, uint64_t n, uint8_t* d_dst);      // 5 B per value
int launch_encode_points(hipStream_t st, const float* d_points, uint64_t n, uint8_t* d_dst);   // 16 B per point
int launch_decode_f32(hipStream_t st, const uint8_t* d_src, uint64_t n, float* d_out, int* d_err);
int launch_decode_points(hipStream_t st, const uint8_t* d_src, uint64_t n, float* d_out, int* d_err);

// client.hip: stable argsort of f32 by total_cmp + itertools-style minmax (the client's voxel ordering),
// and the instance merge in front of the generator.
size_t order_workspace_bytes(size_t n);
int launch_order_cells(Arena& ws, hipStream_t st, const float* d_dist, size_t n, uint32_t* d_ordered, float* d_limits);
struct InstanceDev {        // one model instance, device pointers
  const void* vertices;     // first position; consecutive positions are `stride` bytes apart
  const uint32_t* indices;
  uint64_t stride;
  float m[16];              // glam Mat4, column-major: x_axis, y_axis, z_axis, w_axis
};
size_t merge_workspace_bytes();
int launch_merge_instances(Arena& ws, hipStream_t st, const InstanceDev* d_inst, const uint64_t* d_vfirst, const uint64_t* d_ifirst,
                           uint32_t n_inst, uint64_t n_vertices, uint64_t n_indices, float* d_vertices, uint32_t* d_indices,
                           float* d_bbox);

}  // namespace m2s
