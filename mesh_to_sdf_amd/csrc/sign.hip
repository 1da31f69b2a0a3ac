// sign.hip — SignMethod::Raycast for the grid path (generate/grid.rs:568-684), gfx950.
//
// Reference rule: for every grid line along +X, +Y, +Z (origin = centre of cell 0 of the line,
// grid.rs:570,648-684) every triangle hit with t > 0 (geo.rs:165-216) increments the cells
// 0..=k of that line, k = min(floor(t / cell_size[axis]) as usize, n_axis - 1) (grid.rs:605-617);
// a cell is inside iff at least two of its three per-axis counts are odd (grid.rs:622-639).
//
// Restated for the GPU as bit planes in the grid's own layout (bit z of word (x*ny + y)*nzw + z/32):
//   k_ray_mark : TRIANGLE-parallel.  Each triangle visits only the grid lines whose cell-0 centre
//                lies in its 1e-4-padded box (exactly the candidates bvh.traverse hands the
//                reference), runs the exact ray test and XORs ONE marker bit at bucket k.
//   k_scan_xy  : suffix XOR of the markers along x / y  ==  parity of "increment cells 0..=k" (both planes in one launch).
//   k_scan_z_combine : suffix XOR along z inside the words, then majority of the three planes.
// Work is O(T * lines-per-triangle + N^3 / 32) instead of O(N^2 log T + hits * N) atomics.
#include <algorithm>

#include "common.h"
#include "geo.hip.h"

namespace m2s {

namespace {

constexpr uint32_t SMALL_WINDOW = 64;  // lines a lane handles alone; larger windows are spread over the wave
#ifndef M2S_RAY_SHARED_BELOW
#define M2S_RAY_SHARED_BELOW 16384
#endif
constexpr uint32_t RAY_MARK_SHARED_BELOW = M2S_RAY_SHARED_BELOW;   // meshes of up to this many triangles: sixteen lanes per triangle (k_ray_mark<16>)

struct Window {
  uint32_t ulo, uhi, wlo, whi;  // inclusive index ranges on the two free axes; empty if ulo > uhi
};

__device__ __forceinline__ void axis_range(float bmn, float bmx, float first, float cs, uint32_t n, uint32_t* lo,
                                           uint32_t* hi) {
  // conservative index range of cells whose centre first + i*cs can lie in [bmn, bmx]
  float i0 = (bmn - first) / cs, i1 = (bmx - first) / cs;
  if (i0 > i1) { float t = i0; i0 = i1; i1 = t; }
  const bool bad = !(i0 == i0) || !(i1 == i1) || !(fabsf(i0) < 3.0e38f) || !(fabsf(i1) < 3.0e38f);
  if (bad) { *lo = 0; *hi = n - 1; return; }
  // cells whose centre index lies in [i0, i1], widened by a tolerance that covers the f32 rounding of
  // the index computation (the exact closed-interval test on the true centres decides membership)
  const float tol = 1.0e-3f + 4.0e-6f * fmaxf(fabsf(i0), fabsf(i1)) + 1.0e-6f * (fabsf(bmn) + fabsf(bmx) + fabsf(first)) / fabsf(cs);
  const float l = ceilf(i0 - tol), h = floorf(i1 + tol);
  if (h < 0.0f || l > (float)(n - 1) || h < l) { *lo = 1; *hi = 0; return; }
  *lo = l < 0.0f ? 0u : (uint32_t)l;
  *hi = h >= (float)(n - 1) ? n - 1 : (uint32_t)h;
}

template <int AXIS>
struct FreeAxes {
  static constexpr int U = AXIS == 0 ? 1 : 0;
  static constexpr int W = AXIS == 2 ? 1 : 2;
};

// A call that computes an x-slab only needs the planes of the slab's layers: Y- and Z-lines lie inside one x-layer,
// so a line outside the slab is never looked at (its markers would only be read by cells the call does not compute); X-lines cross
// all layers and are all traced — their hits are slab-independent, t is measured from cell 0 — but the suffix scan along x stops
// at the slab's first layer.  SlabX: the real x-layers of the slab ([lo, hi) is their hull; an interleaved slab owns the chunks
// [lo + j * period, + 2^chunk_log) inside it); all == true: the whole grid (persistent meshes keep their planes per grid, not per slab).
struct SlabX {
  uint32_t lo, hi, chunk_log, period;
  bool all;
};
__device__ __forceinline__ bool x_in_slab(const SlabX& sx, uint32_t x) {
  if (sx.all) return true;
  if (x < sx.lo || x >= sx.hi) return false;
  return sx.chunk_log >= 31u || ((x - sx.lo) % sx.period) < (1u << sx.chunk_log);
}
template <int AXIS>
__device__ __forceinline__ Window make_window(f3 mn, f3 mx, const GridParams& g, const SlabX& sx) {
  constexpr int U = FreeAxes<AXIS>::U, W = FreeAxes<AXIS>::W;
  const float bmn[3] = {mn.x, mn.y, mn.z}, bmx[3] = {mx.x, mx.y, mx.z};
  Window w;
  axis_range(bmn[U], bmx[U], g.first[U], g.size[U], g.n[U], &w.ulo, &w.uhi);
  axis_range(bmn[W], bmx[W], g.first[W], g.size[W], g.n[W], &w.wlo, &w.whi);
  if (AXIS != 0 && !sx.all && w.ulo <= w.uhi) {      // U is x for the Y- and Z-lines: only the slab's layers
    w.ulo = max(w.ulo, sx.lo);
    w.uhi = min(w.uhi, sx.hi - 1u);
  }
  if (w.wlo > w.whi) { w.ulo = 1; w.uhi = 0; }
  return w;
}
__device__ __forceinline__ uint32_t window_count(const Window& w) {
  return w.ulo > w.uhi ? 0u : (w.uhi - w.ulo + 1u) * (w.whi - w.wlo + 1u);
}

// One grid line (free-axis cell iu, iw) against one triangle.
template <int AXIS>
__device__ __forceinline__ void mark_line(f3 a, f3 b, f3 c, f3 mn, f3 mx, const GridParams& g, const SlabX& sx, uint32_t iu,
                                          uint32_t iw, uint32_t* __restrict__ plane) {
  constexpr int U = FreeAxes<AXIS>::U, W = FreeAxes<AXIS>::W;
  if (AXIS != 0 && !x_in_slab(sx, iu)) return;        // (an interleaved slab: the layers between its chunks)
  uint32_t cell[3];
  cell[AXIS] = 0; cell[U] = iu; cell[W] = iw;
  const f3 o = {cell_center(g.first[0], g.size[0], cell[0]), cell_center(g.first[1], g.size[1], cell[1]),
                cell_center(g.first[2], g.size[2], cell[2])};                     // grid.rs:570
  if (!ray_meets_box<AXIS>(o, mn, mx)) return;                                   // bvh.traverse candidate rule
  float t;
  if (!ray_triangle_aligned<AXIS>(o, a, b, c, &t)) return;                       // grid.rs:601-603
  const float f = floorf(t / g.size[AXIS]);                                      // grid.rs:605-606
  const uint32_t n = g.n[AXIS];
  uint32_t k = 0;                                                                // `as usize` saturates, NaN -> 0
  if (f == f && f > 0.0f) k = f >= (float)n ? n - 1 : min((uint32_t)f, n - 1);   // .min(n - 1), grid.rs:607
  cell[AXIS] = k;
  const size_t word = ((size_t)cell[0] * g.n[1] + cell[1]) * g.nzw + (cell[2] >> 5);
  atomicXor(&plane[word], 1u << (cell[2] & 31u));
}

constexpr uint32_t BIG_CHUNK = 2048;   // lines of a big window one workgroup pass handles (256 threads x 8)
struct BigList {
  unsigned long long* counter;   // (items << 32) | chunks
  uint2* items;                  // x = triangle | axis << 30, y = first chunk;  nullptr: no list (small grids)
};

// G lanes share a triangle (sub = this lane's place among them): the lines of a window of up to SMALL_WINDOW x G lines are dealt out to them
template <int AXIS, uint32_t G>
__device__ __forceinline__ void mark_axis(bool valid, uint32_t tri, uint32_t sub, f3 a, f3 b, f3 c, f3 mn, f3 mx, const GridParams& g, const SlabX& sx,
                                          uint32_t* __restrict__ plane, const BigList& list) {
  Window w = {1, 0, 1, 0};
  if (valid) w = make_window<AXIS>(mn, mx, g, sx);
  const uint32_t cnt = window_count(w);
  bool small = cnt <= SMALL_WINDOW * G;
  if (small) {
    if (G == 1u) {
      for (uint32_t iu = w.ulo; iu <= w.uhi && cnt; ++iu)
        for (uint32_t iw = w.wlo; iw <= w.whi; ++iw) mark_line<AXIS>(a, b, c, mn, mx, g, sx, iu, iw, plane);
    } else {
      const uint32_t nw = w.whi - w.wlo + 1u;
      for (uint32_t i = sub; i < cnt; i += G) mark_line<AXIS>(a, b, c, mn, mx, g, sx, w.ulo + i / nw, w.wlo + i % nw, plane);
    }
  }
  if (sub != 0u) small = true;                       // a large window is its first lane's business
  if (list.items != nullptr) {
    // Windows too large for one lane go onto a work list; k_ray_mark_big spreads their lines over the whole
    // chip.  (A low-poly mesh in a fine grid has FEW triangles with HUGE windows: walking them with the owner's
    // wave alone left the chip idle — suzanne in 512^3: 1.8 ms of marking on 16 waves.)
    // One packed 64-bit counter hands out the item slot (high word) and the first chunk (low word) together,
    // so item order == chunk order and the consumer can binary-search chunk -> item.
    if (!small) {
      const uint32_t nchunks = (cnt + BIG_CHUNK - 1u) / BIG_CHUNK;
      const unsigned long long old = atomicAdd(list.counter, (1ull << 32) | (unsigned long long)nchunks);
      const uint32_t slot = (uint32_t)(old >> 32);
      list.items[slot] = make_uint2((uint32_t)tri | ((uint32_t)AXIS << 30), (uint32_t)old);
    }
    return;
  }
  // small grids (no list): the whole wave walks a big window, one owner lane at a time
  unsigned long long big = __ballot(!small);
  const int lane = threadIdx.x & 63;
  while (big) {
    const int src = __ffsll((long long)big) - 1;
    big &= big - 1;
    const f3 sa = {__shfl(a.x, src), __shfl(a.y, src), __shfl(a.z, src)};
    const f3 sb = {__shfl(b.x, src), __shfl(b.y, src), __shfl(b.z, src)};
    const f3 sc = {__shfl(c.x, src), __shfl(c.y, src), __shfl(c.z, src)};
    const f3 smn = {__shfl(mn.x, src), __shfl(mn.y, src), __shfl(mn.z, src)};
    const f3 smx = {__shfl(mx.x, src), __shfl(mx.y, src), __shfl(mx.z, src)};
    const uint32_t ulo = __shfl(w.ulo, src), uhi = __shfl(w.uhi, src), wlo = __shfl(w.wlo, src),
                   whi = __shfl(w.whi, src);
    const uint32_t nw = whi - wlo + 1u;
    const uint64_t total = (uint64_t)(uhi - ulo + 1u) * nw;
    for (uint64_t i = lane; i < total; i += 64)
      mark_line<AXIS>(sa, sb, sc, smn, smx, g, sx, ulo + (uint32_t)(i / nw), wlo + (uint32_t)(i % nw), plane);
  }
}

// G = 1: a lane per triangle (meshes that fill the chip by themselves).  G = 16: sixteen lanes per triangle — a low-poly mesh has few
// triangles with wide windows, and a lane that walks 3 x 64 lines alone is the whole sign phase (suzanne, 968 triangles, in 64^3:
// 100 us of marking on 16 waves where everything else of the call's sign planes takes 20).
template <uint32_t G>
__global__ __launch_bounds__(256) void k_ray_mark(DeviceMesh mesh, GridParams g, SlabX sx, uint32_t* __restrict__ px,
                                                  uint32_t* __restrict__ py, uint32_t* __restrict__ pz, BigList list) {
  const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x, t = gid / G, sub = gid % G;
  const bool valid = t < mesh.n_tris;
  f3 a = {0, 0, 0}, b = {0, 0, 0}, c = {0, 0, 0}, mn = {0, 0, 0}, mx = {0, 0, 0};
  if (valid) {
    const TriRec r = mesh.tris[t];
    a = mk3(r.ax, r.ay, r.az);
    b = mk3(r.bx, r.by, r.bz);
    c = mk3(r.cx, r.cy, r.cz);
    triangle_bounding_box(a, b, c, &mn, &mx);
  }
  mark_axis<0, G>(valid, t, sub, a, b, c, mn, mx, g, sx, px, list);
  mark_axis<1, G>(valid, t, sub, a, b, c, mn, mx, g, sx, py, list);
  mark_axis<2, G>(valid, t, sub, a, b, c, mn, mx, g, sx, pz, list);
}

template <int AXIS>
__device__ __forceinline__ void mark_chunk(const TriRec& r, const GridParams& g, const SlabX& sx, uint32_t chunk, uint32_t* __restrict__ plane) {
  const f3 a = mk3(r.ax, r.ay, r.az), b = mk3(r.bx, r.by, r.bz), c = mk3(r.cx, r.cy, r.cz);
  f3 mn, mx;
  triangle_bounding_box(a, b, c, &mn, &mx);
  const Window w = make_window<AXIS>(mn, mx, g, sx);
  const uint32_t nw = w.whi - w.wlo + 1u, total = window_count(w);
  const uint32_t i0 = chunk * BIG_CHUNK, i1 = min(total, i0 + BIG_CHUNK);
  for (uint32_t i = i0 + threadIdx.x; i < i1; i += blockDim.x)
    mark_line<AXIS>(a, b, c, mn, mx, g, sx, w.ulo + i / nw, w.wlo + i % nw, plane);
}

// Chunks of the big windows, one per workgroup pass, over a fixed grid (the totals are only known on the device).
__global__ __launch_bounds__(256) void k_ray_mark_big(DeviceMesh mesh, GridParams g, SlabX sx, uint32_t* __restrict__ px,
                                                      uint32_t* __restrict__ py, uint32_t* __restrict__ pz, BigList list) {
  const unsigned long long ctr = *list.counter;
  const uint32_t n_items = (uint32_t)(ctr >> 32), n_chunks = (uint32_t)ctr;
  for (uint32_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
    uint32_t lo = 0, hi = n_items;           // last item whose first chunk <= chunk (first chunks ascend with the slot)
    while (hi - lo > 1) {
      const uint32_t mid = (lo + hi) >> 1;
      if (list.items[mid].y <= chunk) lo = mid; else hi = mid;
    }
    const uint2 it = list.items[lo];
    const TriRec r = mesh.tris[it.x & 0x3fffffffu];
    const uint32_t local = chunk - it.y, axis = it.x >> 30;
    if (axis == 0) mark_chunk<0>(r, g, sx, local, px);
    else if (axis == 1) mark_chunk<1>(r, g, sx, local, py);
    else mark_chunk<2>(r, g, sx, local, pz);
  }
}

// Suffix XOR along x: thread = one (y, zw) word column, walking x from nx-1 down to 0.
// (down to layer x_stop only: a slab call never reads the layers in front of its first)
__device__ __forceinline__ void scan_x_column(uint32_t* __restrict__ p, uint32_t nx, size_t row_words /*ny*nzw*/, uint32_t x_stop, size_t col) {
  if (col >= row_words) return;
  uint32_t run = 0;
  int64_t x = (int64_t)nx - 1;
  const int64_t stop = (int64_t)x_stop;
  for (; x >= stop + 7; x -= 8) {
    uint32_t v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = p[(size_t)(x - k) * row_words + col];
#pragma unroll
    for (int k = 0; k < 8; ++k) { run ^= v[k]; p[(size_t)(x - k) * row_words + col] = run; }
  }
  for (; x >= stop; --x) { run ^= p[(size_t)x * row_words + col]; p[(size_t)x * row_words + col] = run; }
}

// Suffix XOR along y: thread = one (layer, zw) pair; `layers` x-layers of the (virtual) slab g, or of the whole grid.
__device__ __forceinline__ void scan_y_column(uint32_t* __restrict__ p, const GridParams& g, uint32_t layers, bool whole, uint32_t ny, uint32_t nzw, size_t id) {
  if (id >= (size_t)layers * nzw) return;
  const size_t v = id / nzw, zw = id % nzw;
  const size_t x = whole ? v : (size_t)slab_x(g, (uint32_t)v);
  uint32_t* base = p + x * ny * nzw + zw;
  uint32_t run = 0;
  int64_t y = (int64_t)ny - 1;
  for (; y >= 7; y -= 8) {
    uint32_t v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = base[(size_t)(y - k) * nzw];
#pragma unroll
    for (int k = 0; k < 8; ++k) { run ^= v[k]; base[(size_t)(y - k) * nzw] = run; }
  }
  for (; y >= 0; --y) { run ^= base[(size_t)y * nzw]; base[(size_t)y * nzw] = run; }
}

// Both scans in one launch (round 6): they work on different planes, and each is a column walk as long as its axis whatever the grid — one after
// the other they were 39 + 47 us of the 512^3 call's sign planes.  Blocks [0, x_blocks) take the x columns, the rest the y columns.
__global__ __launch_bounds__(256) void k_scan_xy(uint32_t* __restrict__ px, uint32_t* __restrict__ py, GridParams g, uint32_t layers, bool whole, size_t row_words,
                                                 uint32_t x_stop, uint32_t x_blocks) {
  if (blockIdx.x < x_blocks) scan_x_column(px, g.n[0], row_words, x_stop, (size_t)blockIdx.x * 256u + threadIdx.x);
  else scan_y_column(py, g, layers, whole, g.n[1], g.nzw, (size_t)(blockIdx.x - x_blocks) * 256u + threadIdx.x);
}

// Suffix XOR along z inside each (x,y) row, then "at least two of three odd" (grid.rs:630-636).
// The majority plane overwrites pz.
__global__ __launch_bounds__(256) void k_scan_z_combine(const uint32_t* __restrict__ px, const uint32_t* __restrict__ py,
                                                        uint32_t* __restrict__ pz, GridParams g, bool whole, size_t rows, uint32_t nzw) {
  const size_t vrow = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // row of the (virtual) slab: layer * ny + y
  if (vrow >= rows) return;
  const size_t row = whole ? vrow : (size_t)slab_x(g, (uint32_t)(vrow / g.n[1])) * g.n[1] + vrow % g.n[1];
  uint32_t carry = 0;
  for (int64_t zw = (int64_t)nzw - 1; zw >= 0; --zw) {
    const size_t i = row * nzw + (size_t)zw;
    uint32_t z = pz[i];
    z ^= z >> 1; z ^= z >> 2; z ^= z >> 4; z ^= z >> 8; z ^= z >> 16;   // bit i = XOR of marker bits >= i
    z ^= carry;
    carry = (z & 1u) ? 0xffffffffu : 0u;
    const uint32_t x = px[i], y = py[i];
    pz[i] = (x & y) | (x & z) | (y & z);
  }
}

// Same as k_scan_z_combine with one LANE per 32-bit word (coalesced): the nzw words of a row sit in
// nzw consecutive lanes (nzw a power of two <= 64), the carry from the higher words of the row is a
// segmented suffix-XOR of the word parities done with shuffles.
__global__ __launch_bounds__(256) void k_scan_z_combine_rows(const uint32_t* __restrict__ px,
                                                             const uint32_t* __restrict__ py, uint32_t* __restrict__ pz,
                                                             GridParams g, bool whole, size_t words, uint32_t nzw) {
  const size_t vi = (size_t)blockIdx.x * blockDim.x + threadIdx.x;     // word of the (virtual) slab; a layer is a whole number of rows
  const bool valid = vi < words;
  const size_t layer_words = (size_t)g.n[1] * nzw;
  const size_t i = (whole || !valid) ? vi : (size_t)slab_x(g, (uint32_t)(vi / layer_words)) * layer_words + vi % layer_words;
  uint32_t z = valid ? pz[i] : 0u;
  z ^= z >> 1; z ^= z >> 2; z ^= z >> 4; z ^= z >> 8; z ^= z >> 16;     // bit b = XOR of marker bits >= b
  const uint32_t zw = (uint32_t)(i & (nzw - 1));                          // word index inside the row
  uint32_t par = z & 1u;                                                  // parity of the whole word
  uint32_t c = par;                                                       // inclusive suffix XOR over the row
  for (uint32_t off = 1; off < nzw; off <<= 1) {
    const uint32_t t = __shfl_down(c, off);
    if (zw + off < nzw) c ^= t;
  }
  if ((c ^ par) & 1u) z = ~z;                                             // exclusive: words above this one
  if (valid) {
    const uint32_t x = px[i], y = py[i];
    pz[i] = (x & y) | (x & z) | (y & z);
  }
}

}  // namespace

// Grids whose largest possible window (a whole face of the grid) exceeds this use the work list for big windows.
constexpr uint64_t LIST_ABOVE_LINES = 4096;
static bool use_big_list(const GridParams& g) {
  const uint64_t a = (uint64_t)g.n[1] * g.n[2], b = (uint64_t)g.n[0] * g.n[2], c = (uint64_t)g.n[0] * g.n[1];
  return std::max(a, std::max(b, c)) > LIST_ABOVE_LINES;
}

size_t sign_workspace_bytes(const GridParams& g, size_t n_tris) {
  return 3 * ((size_t)g.n[0] * g.n[1] * g.nzw * 4 + 256) + 2048 + (use_big_list(g) ? 3 * n_tris * sizeof(uint2) + 512 : 0);
}

int build_grid_sign_plane(Arena& ws, hipStream_t st, const DeviceMesh& mesh, const GridParams& g,
                          const uint32_t** d_inside_plane, bool slab_only) {
  const size_t words = (size_t)g.n[0] * g.n[1] * g.nzw;
  // the three planes and the work-list counter in ONE block: one memset instead of four launches (a launch costs the host ~8 us
  // and this sequence sits in front of the build's on the calling thread)
  const size_t wpad = (words + 63) / 64 * 64;                  // planes 256-byte aligned
  uint32_t* block = ws.take<uint32_t>(3 * wpad + 16);
  if (!block) {
    set_error("internal: sign workspace too small");
    return M2S_ERR_HIP_INTERNAL;
  }
  uint32_t *px = block, *py = block + wpad, *pz = block + 2 * wpad;
  *d_inside_plane = pz;
  if (words == 0) return 0;
  M2S_HIP_CHECK(hipMemsetAsync(block, 0, (3 * wpad + 16) * 4, st));
  const unsigned B = 256;
  if (mesh.n_tris) {
    BigList list{nullptr, nullptr};
    if (use_big_list(g)) {
      list.counter = reinterpret_cast<unsigned long long*>(block + 3 * wpad);
      list.items = ws.take<uint2>(3 * (size_t)mesh.n_tris);
      if (!list.items) {
        set_error("internal: sign workspace too small");
        return M2S_ERR_HIP_INTERNAL;
      }
    }
    // the slab's real x-layers (see SlabX): hull [lo, hi), chunks of an interleaved slab inside it
    const uint32_t vlayers = g.xe - g.xb;
    const bool whole = !slab_only || vlayers == 0 || (g.xb == 0 && vlayers == g.n[0] && g.chunk_log >= 31u);
    SlabX sx{0u, g.n[0], 31u, 0u, true};
    if (!whole) {
      sx.all = false;
      sx.lo = g.xb;
      sx.hi = slab_x(g, vlayers - 1u) + 1u;
      sx.chunk_log = g.chunk_log;
      sx.period = g.period ? g.period : 1u;
    }
    const uint32_t layers = whole ? g.n[0] : vlayers;
    if (mesh.n_tris <= RAY_MARK_SHARED_BELOW) hipLaunchKernelGGL(k_ray_mark<16>, dim3((mesh.n_tris * 16u + B - 1) / B), dim3(B), 0, st, mesh, g, sx, px, py, pz, list);
    else hipLaunchKernelGGL(k_ray_mark<1>, dim3((mesh.n_tris + B - 1) / B), dim3(B), 0, st, mesh, g, sx, px, py, pz, list);
    if (list.items) hipLaunchKernelGGL(k_ray_mark_big, dim3(2048), dim3(B), 0, st, mesh, g, sx, px, py, pz, list);
    const size_t row_words = (size_t)g.n[1] * g.nzw;
    const size_t ycols = (size_t)layers * g.nzw;
    const unsigned x_blocks = (unsigned)((row_words + B - 1) / B), y_blocks = (unsigned)((ycols + B - 1) / B);
    hipLaunchKernelGGL(k_scan_xy, dim3(x_blocks + y_blocks), dim3(B), 0, st, px, py, g, layers, whole, row_words, whole ? 0u : sx.lo, x_blocks);
    const size_t rows = (size_t)layers * g.n[1], slab_words = rows * g.nzw;
    if (g.nzw <= 64 && (g.nzw & (g.nzw - 1)) == 0)
      hipLaunchKernelGGL(k_scan_z_combine_rows, dim3((unsigned)((slab_words + B - 1) / B)), dim3(B), 0, st, px, py, pz, g, whole, slab_words, g.nzw);
    else
      hipLaunchKernelGGL(k_scan_z_combine, dim3((unsigned)((rows + B - 1) / B)), dim3(B), 0, st, px, py, pz, g, whole, rows, g.nzw);
  }
  M2S_HIP_CHECK(hipGetLastError());
  return 0;
}

// m2s_warmup: the first launch of a kernel of this translation unit makes the runtime load its code object (all its kernels).
__global__ void k_warm_sign() {}
void warm_sign(hipStream_t st) {
  hipLaunchKernelGGL(k_warm_sign, dim3(1), dim3(64), 0, st);
  // ... and resolves a kernel FUNCTION at its own first launch (~0.3 ms each): ask for the attributes of the ones a first call uses
  const void* fns[] = {
      (const void*)k_ray_mark<1>,
      (const void*)k_ray_mark<16>,
      (const void*)k_ray_mark_big,
      (const void*)k_scan_xy};
  hipFuncAttributes attr;
  for (const void* f : fns) (void)hipFuncGetAttributes(&attr, f);
  (void)hipGetLastError();
}

}  // namespace m2s
