// lbvh_sort.hip.h — the sort of the build's (Morton key, triangle) pairs for meshes of up to SS_MAX_N triangles: a sample sort in
// three launches (rocPRIM's radix_sort_pairs is a block sort + 7 merge launches at 100 k pairs: 60 us of the build's 234).
// Included by bvh.hip inside its anonymous namespace.
//
// The pairs are sorted as COMPOSITES (key, triangle index): all distinct, so the sorted order is unique — the order a stable sort of
// the keys gives (values 0, 1, 2, ... in input order), whatever the algorithm.  tests/test_gpu_build.py pins the trees.
//
//   k_sort_tiles   one workgroup per tile of W pairs: folds the scene partials, computes the keys of its tile (what k_morton_keys
//                  does on the rocPRIM path), sorts them in LDS (bitonic network, four elements per thread and round) and writes the
//                  sorted tile and every G-th element of it as a SAMPLE.  A mesh of at most W triangles is finished here.
//   k_sort_rank    one workgroup per tile: all samples into LDS, the global rank of its own samples by binary searches over the other
//                  tiles' (sorted) sample runs.  The samples of rank q, 2q, ... are the SPLITTERS of B = 4 x tiles buckets; and for
//                  every splitter the tile records how many of its own samples lie at or below it — the G-element window of the
//                  tile the splitter falls into.
//                  (The samples of tile t are its elements o_t + j G with o_t = t G / tiles: staggered, see sample_offset.)
//   k_sort_buckets one workgroup per bucket: the exact boundary in every tile's window (one load per element of the windows), the
//                  bucket's pieces gathered into LDS, sorted, written at the bucket's global position (the sum of the lower boundaries).
//
// Regular sampling bounds a bucket: a piece of `len` consecutive elements of a tile holds at least (len - (G - 1)) / G of its
// samples, all the samples inside a bucket lie between two consecutive splitters, so
//     bucket size <= G * q + tiles * (G - 1)           (q = samples per tile / 4 = samples between two splitters)
// which ss_max_tiles() keeps within SS_CAP by construction (the kernel still checks and raises ERRF_BUILD_TIMEOUT).

constexpr uint32_t SS_G = 32;                  // sampling gap
constexpr uint32_t SS_CAP = 4096;              // pairs a bucket workgroup holds in LDS
constexpr uint32_t SS_BUCKET_THREADS = 512;
constexpr uint32_t SS_RANK_THREADS = 256;
constexpr uint32_t SS_PER_TILE = 4;            // buckets per tile

__host__ __device__ constexpr uint32_t ss_max_tiles(uint32_t w) { return SS_CAP / SS_G - (w / SS_G) / SS_PER_TILE; }   // G (q + p) <= CAP

// The samples of tile t are its elements o_t, o_t + G, o_t + 2 G, ...: staggered over the tiles, so that tiles with the same
// distribution of keys (triangles in random order) do not all leave their first G - 1 elements to the first bucket
__host__ __device__ __forceinline__ uint32_t sample_offset(uint32_t t, uint32_t p) { return (t * SS_G) / p; }

__device__ __forceinline__ bool pair_less(uint64_t ka, uint32_t ia, uint64_t kb, uint32_t ib) { return ka < kb || (ka == kb && ia < ib); }

// compare-exchange towards ascending (asc) or descending order
__device__ __forceinline__ void pair_cx(uint64_t& ka, uint32_t& ia, uint64_t& kb, uint32_t& ib, bool asc) {
  const bool a_gt_b = pair_less(kb, ib, ka, ia);
  if (a_gt_b == asc) {
    const uint64_t tk = ka; ka = kb; kb = tk;
    const uint32_t ti = ia; ia = ib; ib = ti;
  }
}

// The phases k_first, 2 k_first, ..., n of the bitonic sorting network over sk / si [0, n) (n a power of two; every aligned run of
// k_first / 2 elements already sorted, directions alternating — k_first = 2: any input).  A thread takes FOUR elements per round and
// does two sub-stages (distances j and j / 2) on them in registers: half the LDS round trips of the textbook form.
template <uint32_t THREADS>
__device__ __forceinline__ void bitonic_rounds(uint64_t* __restrict__ sk, uint32_t* __restrict__ si, uint32_t n, uint32_t k_first, uint32_t tid) {
  for (uint32_t k = k_first; k <= n; k <<= 1) {
    uint32_t j = k >> 1;
    for (; j >= 2; j >>= 2) {
      const uint32_t h = j >> 1;
      for (uint32_t q = tid; q < (n >> 2); q += THREADS) {
        const uint32_t low = q & (h - 1u);
        const uint32_t i0 = ((q - low) << 2) | low, i1 = i0 + h, i2 = i0 + j, i3 = i2 + h;
        const bool asc = (i0 & k) == 0;
        uint64_t k0 = sk[i0], k1 = sk[i1], k2 = sk[i2], k3 = sk[i3];
        uint32_t v0 = si[i0], v1 = si[i1], v2 = si[i2], v3 = si[i3];
        pair_cx(k0, v0, k2, v2, asc);
        pair_cx(k1, v1, k3, v3, asc);
        pair_cx(k0, v0, k1, v1, asc);
        pair_cx(k2, v2, k3, v3, asc);
        sk[i0] = k0; sk[i1] = k1; sk[i2] = k2; sk[i3] = k3;
        si[i0] = v0; si[i1] = v1; si[i2] = v2; si[i3] = v3;
      }
      __syncthreads();
    }
    if (j == 1) {
      for (uint32_t q = tid; q < (n >> 1); q += THREADS) {
        const uint32_t i0 = q << 1, i1 = i0 + 1u;
        const bool asc = (i0 & k) == 0;
        uint64_t k0 = sk[i0], k1 = sk[i1];
        uint32_t v0 = si[i0], v1 = si[i1];
        pair_cx(k0, v0, k1, v1, asc);
        sk[i0] = k0; sk[i1] = k1;
        si[i0] = v0; si[i1] = v1;
      }
      __syncthreads();
    }
  }
}

// number of elements of the sorted run [0, LEN) (LEN a power of two) that are < (kq, iq) [OR_EQUAL: <=]
template <uint32_t LEN, bool OR_EQUAL>
__device__ __forceinline__ uint32_t run_rank(const uint64_t* __restrict__ rk, const uint32_t* __restrict__ ri, uint64_t kq, uint32_t iq) {
  uint32_t pos = 0;
#pragma unroll
  for (uint32_t s = LEN >> 1; s >= 1; s >>= 1) {
    const uint64_t ke = rk[pos + s - 1u];
    const uint32_t ie = ri[pos + s - 1u];
    const bool below = OR_EQUAL ? !pair_less(kq, iq, ke, ie) : pair_less(ke, ie, kq, iq);
    pos += below ? s : 0u;
  }
  {
    const uint64_t ke = rk[pos];
    const uint32_t ie = ri[pos];
    const bool below = OR_EQUAL ? !pair_less(kq, iq, ke, ie) : pair_less(ke, ie, kq, iq);
    pos += below ? 1u : 0u;
  }
  return pos;
}

// 63-bit Morton key of a triangle's box centre over the scene's box of centres (k_morton_keys computes the same)
__device__ __forceinline__ uint64_t morton_key_of_box(const Box& bx, const float slo[3], const float shi[3]) {
  const float c[3] = {0.5f * (bx.mnx + bx.mxx), 0.5f * (bx.mny + bx.mxy), 0.5f * (bx.mnz + bx.mxz)};
  uint32_t q[3];
  for (int k = 0; k < 3; ++k) {
    float u = (c[k] - slo[k]) / (shi[k] - slo[k]);
    u = (u == u) ? fminf(fmaxf(u, 0.0f), 1.0f) : 0.0f;
    q[k] = min((uint32_t)(u * 2097152.0f), 2097151u);
  }
  return (expand21(q[0]) << 2) | (expand21(q[1]) << 1) | expand21(q[2]);
}

// Folds the per-block scene partials of k_tri_setup (every workgroup does it for itself: a few KB out of L2); block 0 publishes
// the final values (scene[0..5] order-encoded min xyz / max xyz, scene[6] the largest finite |coordinate| as float bits, scene[7] the
// treelet root counter, cleared).
template <uint32_t THREADS>
__device__ __forceinline__ void fold_scene(int* __restrict__ scene, uint32_t n_partials, uint32_t tid, float slo[3], float shi[3]) {
  __shared__ int s_part[6][THREADS / 64];
  __shared__ int s_scene[6];
  int lo[3] = {INT32_MAX, INT32_MAX, INT32_MAX}, hi[3] = {INT32_MIN, INT32_MIN, INT32_MIN};
  for (uint32_t b = tid; b < n_partials; b += THREADS)
    for (int k = 0; k < 3; ++k) {
      lo[k] = min(lo[k], scene[8 + b * 6 + k]);
      hi[k] = max(hi[k], scene[8 + b * 6 + 3 + k]);
    }
  for (int k = 0; k < 3; ++k) {
    int l = lo[k], h = hi[k];
    for (int off = 32; off > 0; off >>= 1) {
      l = min(l, __shfl_xor(l, off));
      h = max(h, __shfl_xor(h, off));
    }
    if ((tid & 63u) == 0) { s_part[k][tid >> 6] = l; s_part[3 + k][tid >> 6] = h; }
  }
  __syncthreads();
  if (tid < 6) {
    int v = s_part[tid][0];
    for (uint32_t w = 1; w < THREADS / 64; ++w) v = tid < 3 ? min(v, s_part[tid][w]) : max(v, s_part[tid][w]);
    s_scene[tid] = v;
  }
  __syncthreads();
  if (blockIdx.x == 0 && tid == 0) {
    float s = 0.0f;
    for (int k = 0; k < 6; ++k) {
      scene[k] = s_scene[k];
      const float f = unord(s_scene[k]);
      if (f == f && fabsf(f) < 3.0e38f) s = fmaxf(s, fabsf(f));
    }
    scene[6] = __float_as_int(s);
    scene[7] = 0;
  }
  for (int k = 0; k < 3; ++k) { slo[k] = unord(s_scene[k]); shi[k] = unord(s_scene[3 + k]); }
}

template <uint32_t W>
__global__ __launch_bounds__(W / 4) void k_sort_tiles(const Box* __restrict__ boxes, uint32_t n, int* __restrict__ scene, uint32_t n_partials,
                                                      uint4* __restrict__ tiles, uint4* __restrict__ samples,
                                                      uint64_t* __restrict__ keys_out, uint32_t* __restrict__ order_out) {
  constexpr uint32_t T = W / 4;
  __shared__ uint64_t sk[W];
  __shared__ uint32_t si[W];
  const uint32_t tid = threadIdx.x;
  const uint32_t e0 = 4u * tid, base = blockIdx.x * W + e0;
  uint64_t k[4];
  uint32_t v[4];
  Box bx[4];
  for (uint32_t u = 0; u < 4; ++u)          // on their way while the scene is folded
    if (base + u < n) bx[u] = boxes[base + u];
  float slo[3], shi[3];
  fold_scene<T>(scene, n_partials, tid, slo, shi);
  for (uint32_t u = 0; u < 4; ++u) {
    if (base + u < n) { k[u] = morton_key_of_box(bx[u], slo, shi); v[u] = base + u; }
    else { k[u] = ~0ull; v[u] = 0x80000000u | (e0 + u); }            // padding of the last tile: behind every pair, still distinct
  }
  // the thread's four consecutive elements as the network leaves them after its phases 2 and 4
  pair_cx(k[0], v[0], k[1], v[1], true);
  pair_cx(k[2], v[2], k[3], v[3], true);
  pair_cx(k[0], v[0], k[2], v[2], true);
  pair_cx(k[1], v[1], k[3], v[3], true);
  pair_cx(k[1], v[1], k[2], v[2], true);
  if (e0 & 4u) {
    uint64_t tk = k[0]; k[0] = k[3]; k[3] = tk; tk = k[1]; k[1] = k[2]; k[2] = tk;
    uint32_t tv = v[0]; v[0] = v[3]; v[3] = tv; tv = v[1]; v[1] = v[2]; v[2] = tv;
  }
  for (uint32_t u = 0; u < 4; ++u) { sk[e0 + u] = k[u]; si[e0 + u] = v[u]; }
  __syncthreads();
  bitonic_rounds<T>(sk, si, W, 8u, tid);
  if (keys_out) {                                   // the whole mesh in one tile: these are the sorted arrays
    for (uint32_t e = tid; e < W && e < n; e += T) { keys_out[e] = sk[e]; order_out[e] = si[e]; }
    return;
  }
  uint4* dst = tiles + (size_t)blockIdx.x * W;
  for (uint32_t e = tid; e < W; e += T) dst[e] = make_uint4((uint32_t)sk[e], (uint32_t)(sk[e] >> 32), si[e], 0u);
  constexpr uint32_t SP = W / SS_G;
  if (tid < SP) {
    const uint32_t e = tid * SS_G + sample_offset(blockIdx.x, gridDim.x);
    samples[blockIdx.x * SP + tid] = make_uint4((uint32_t)sk[e], (uint32_t)(sk[e] >> 32), si[e], 0u);
  }
}

template <uint32_t W>
__global__ __launch_bounds__(SS_RANK_THREADS) void k_sort_rank(const uint4* __restrict__ samples, uint32_t p, uint4* __restrict__ splitters,
                                                               uint8_t* __restrict__ cmat) {
  constexpr uint32_t SP = W / SS_G, Q = SP / SS_PER_TILE, PARTS = SS_RANK_THREADS / SP, MAXS = ss_max_tiles(W) * SP;
  // all samples of all tiles in LDS: 12 bytes each, 86 KB for W = 2048 — gfx950's 160 KB of LDS per CU (one workgroup per CU there); this
  // library is gfx950-only (Makefile: ARCH), a 64 KB-LDS target would have to stream the other tiles' runs through a window instead
  static_assert(MAXS * 12u + SP * 4u <= 160u * 1024u, "k_sort_rank keeps every sample in LDS: needs gfx950's 160 KB");
  __shared__ uint64_t sk[MAXS];
  __shared__ uint32_t si[MAXS];
  __shared__ uint32_t s_rank[SP];
  const uint32_t tid = threadIdx.x, t = blockIdx.x, S = p * SP, B = p * SS_PER_TILE;
  for (uint32_t i = tid; i < S; i += SS_RANK_THREADS) {
    const uint4 e = samples[i];
    sk[i] = (uint64_t)e.x | ((uint64_t)e.y << 32);
    si[i] = e.z;
  }
  if (tid < SP) s_rank[tid] = 0u;
  __syncthreads();
  {
    const uint32_t j = tid % SP, part = tid / SP;
    const uint64_t kq = sk[t * SP + j];
    const uint32_t iq = si[t * SP + j];
    uint32_t cnt = 0;
    for (uint32_t t0 = part; t0 < p; t0 += 4u * PARTS) {            // four runs at a time: their searches are independent chains
      uint32_t pos[4] = {0, 0, 0, 0};
#pragma unroll
      for (uint32_t s = SP >> 1; s >= 1; s >>= 1)
#pragma unroll
        for (uint32_t u = 0; u < 4; ++u) {
          const uint32_t t2 = min(t0 + u * PARTS, p - 1u);
          const uint32_t at = t2 * SP + pos[u] + s - 1u;
          pos[u] += pair_less(sk[at], si[at], kq, iq) ? s : 0u;
        }
#pragma unroll
      for (uint32_t u = 0; u < 4; ++u) {
        const uint32_t t2 = t0 + u * PARTS;
        const uint32_t at = min(t2, p - 1u) * SP + pos[u];
        pos[u] += pair_less(sk[at], si[at], kq, iq) ? 1u : 0u;
        if (t2 < p) cnt += t2 == t ? j : pos[u];
      }
    }
    atomicAdd(&s_rank[j], cnt);
  }
  __syncthreads();
  if (tid < SP) {
    const uint32_t r = s_rank[tid] + 1u;            // samples at or below this one
    if (r % Q == 0u && r / Q < B) splitters[r / Q - 1u] = make_uint4((uint32_t)sk[t * SP + tid], (uint32_t)(sk[t * SP + tid] >> 32), si[t * SP + tid], 0u);
  }
  // splitter b is the sample of rank (b + 1) Q - 1: my samples at or below it
  for (uint32_t b = tid; b + 1u < B; b += SS_RANK_THREADS) {
    const uint32_t rb = (b + 1u) * Q - 1u;
    uint32_t pos = 0;
#pragma unroll
    for (uint32_t s = SP >> 1; s >= 1; s >>= 1) pos += s_rank[pos + s - 1u] <= rb ? s : 0u;
    pos += s_rank[pos] <= rb ? 1u : 0u;
    cmat[(size_t)t * B + b] = (uint8_t)pos;
  }
}

template <uint32_t W>
__global__ __launch_bounds__(SS_BUCKET_THREADS) void k_sort_buckets(const uint4* __restrict__ tiles, uint32_t n, uint32_t p,
                                                                    const uint4* __restrict__ splitters, const uint8_t* __restrict__ cmat,
                                                                    uint64_t* __restrict__ keys_out, uint32_t* __restrict__ order_out, int* __restrict__ err) {
  constexpr uint32_t SP = W / SS_G, T = SS_BUCKET_THREADS, MAXP = ss_max_tiles(W);
  __shared__ uint64_t sk[SS_CAP];
  __shared__ uint32_t si[SS_CAP];
  __shared__ uint32_t s_c[2 * MAXP], s_lo[MAXP], s_hi[MAXP], s_off[MAXP + 1], s_goff;
  const uint32_t tid = threadIdx.x, b = blockIdx.x, B = gridDim.x;
  const bool have_lo = b > 0u, have_hi = b + 1u < B;
  if (tid < 2u * p) {
    const uint32_t t = tid >> 1, side = tid & 1u;
    uint32_t c = side ? SP : 0u;
    if (side ? have_hi : have_lo) c = cmat[(size_t)t * B + (side ? b : b - 1u)];
    s_c[tid] = c;
  }
  uint4 sp[2] = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};
  if (have_lo) sp[0] = splitters[b - 1u];
  if (have_hi) sp[1] = splitters[b];
  __syncthreads();
  // the boundary inside the window: one half-wave per (tile, side), one element per lane
  {
    const uint32_t hw = tid >> 5, l = tid & 31u, n_hw = T / 32u;
    for (uint32_t w0 = hw; w0 < 2u * p; w0 += 4u * n_hw) {
      uint4 e[4];
      bool use[4];
#pragma unroll
      for (uint32_t u = 0; u < 4; ++u) {
        const uint32_t wdx = w0 + u * n_hw;
        use[u] = false;
        if (wdx < 2u * p) {
          const uint32_t t = wdx >> 1, side = wdx & 1u, c = s_c[wdx];
          // c of the tile's samples are <= the splitter: the last of them sits at ws - 1, the next G - 1 elements decide
          const uint32_t ws = c ? (c - 1u) * SS_G + sample_offset(t, p) + 1u : 0u;
          use[u] = (side ? have_hi : have_lo) && l < SS_G - 1u && ws + l < W;
          if (use[u]) e[u] = tiles[(size_t)t * W + ws + l];
        }
      }
#pragma unroll
      for (uint32_t u = 0; u < 4; ++u) {
        const uint32_t wdx = w0 + u * n_hw;
        if (wdx >= 2u * p) break;                              // uniform over the half-wave... and over the wave: n_hw is even
        const uint32_t t = wdx >> 1, side = wdx & 1u, c = s_c[wdx];
        const uint4 s = sp[side];
        const uint64_t kq = (uint64_t)s.x | ((uint64_t)s.y << 32);
        const bool le = use[u] && !pair_less(kq, s.z, (uint64_t)e[u].x | ((uint64_t)e[u].y << 32), e[u].z);
        const unsigned long long bal = __ballot(le);
        const uint32_t bits = (uint32_t)(bal >> (tid & 32u));
        uint32_t cnt = (c ? (c - 1u) * SS_G + sample_offset(t, p) + 1u : 0u) + __popc(bits);
        const uint32_t n_t = min(W, n - t * W);
        if (side == 0u && !have_lo) cnt = 0u;
        if (side == 1u && !have_hi) cnt = n_t;
        cnt = min(cnt, n_t);
        if (l == 0u) (side ? s_hi : s_lo)[t] = cnt;
      }
    }
  }
  __syncthreads();
  if (tid < 64u) {                                               // exclusive prefix of the piece lengths, the bucket's global position
    uint32_t run = 0, goff = 0;
    for (uint32_t t0 = 0; t0 < p; t0 += 64u) {
      const uint32_t t = t0 + tid;
      const uint32_t len = t < p ? s_hi[t] - s_lo[t] : 0u, lo = t < p ? s_lo[t] : 0u;
      uint32_t inc = len, g = lo;
      for (uint32_t off = 1; off < 64u; off <<= 1) {
        const uint32_t o = __shfl_up(inc, off);
        if (tid >= off) inc += o;
      }
      for (uint32_t off = 32; off > 0; off >>= 1) g += __shfl_xor(g, off);
      if (t < p) s_off[t] = run + inc - len;
      run += __shfl(inc, 63);
      goff += g;
    }
    if (tid == 0u) { s_off[p] = run; s_goff = goff; }
  }
  __syncthreads();
  const uint32_t m = s_off[p], goff = s_goff;
  if (m > SS_CAP) { if (tid == 0u) atomicOr(err, ERRF_BUILD_TIMEOUT); return; }   // excluded by ss_max_tiles
  uint32_t N = m;
  if (m > 1u) N = 1u << (32 - __clz(m - 1u));
  for (uint32_t i = tid; i < N; i += T) {
    uint64_t kk = ~0ull;
    uint32_t vv = 0xffffffffu;
    if (i < m) {
      uint32_t lo = 0, hi = p;                                   // the piece that holds element i: last t with s_off[t] <= i
      while (hi - lo > 1u) {
        const uint32_t mid = (lo + hi) >> 1;
        if (s_off[mid] <= i) lo = mid; else hi = mid;
      }
      const uint4 e = tiles[(size_t)lo * W + s_lo[lo] + (i - s_off[lo])];
      kk = (uint64_t)e.x | ((uint64_t)e.y << 32);
      vv = e.z;
    }
    sk[i] = kk;
    si[i] = vv;
  }
  __syncthreads();
  if (N > 1u) bitonic_rounds<T>(sk, si, N, 2u, tid);
  for (uint32_t i = tid; i < m; i += T) { keys_out[goff + i] = sk[i]; order_out[goff + i] = si[i]; }
}

// Which form sorts `n` pairs: tile size 0 = rocPRIM.
inline uint32_t sort_tile_size(size_t n) {
  const int forced = tuning().sort_tile;                          // M2S_SORT_TILE: -1 automatic, 0 rocPRIM, 1024 / 2048
  const uint32_t w = forced == 1024 || forced == 2048 ? (uint32_t)forced : (n <= (size_t)ss_max_tiles(1024) * 1024 ? 1024u : 2048u);
  if (forced == 0 || n > (size_t)ss_max_tiles(w) * w) return 0u;
  return w;
}
struct SortBufs {
  uint4* tiles = nullptr;
  uint4* samples = nullptr;
  uint4* splitters = nullptr;
  uint8_t* cmat = nullptr;
};
inline size_t sample_sort_bytes(size_t n) {                       // nothing where rocPRIM sorts (the tile x bucket matrix is quadratic in n:
  if (sort_tile_size(n) == 0u) return 0;                          // 380 MB at 10 M triangles, 9.5 GB at 50 M); else an upper estimate for either tile size
  const size_t p = (n + 1023) / 1024;
  return (n + 2048) * 16 + p * 64 * 16 + p * SS_PER_TILE * 16 + p * p * SS_PER_TILE + 4 * 256;
}

template <uint32_t W>
void launch_sample_sort(hipStream_t st, const Box* boxes, uint32_t n, int* scene, uint32_t n_partials, const SortBufs& sb,
                        uint64_t* keys_out, uint32_t* order_out, int* d_err) {
  const uint32_t p = (n + W - 1u) / W;
  if (p == 1u) {
    hipLaunchKernelGGL(k_sort_tiles<W>, dim3(1), dim3(W / 4), 0, st, boxes, n, scene, n_partials, (uint4*)nullptr, (uint4*)nullptr, keys_out, order_out);
    return;
  }
  hipLaunchKernelGGL(k_sort_tiles<W>, dim3(p), dim3(W / 4), 0, st, boxes, n, scene, n_partials, sb.tiles, sb.samples, (uint64_t*)nullptr, (uint32_t*)nullptr);
  hipLaunchKernelGGL(k_sort_rank<W>, dim3(p), dim3(SS_RANK_THREADS), 0, st, (const uint4*)sb.samples, p, sb.splitters, sb.cmat);
  hipLaunchKernelGGL(k_sort_buckets<W>, dim3(p * SS_PER_TILE), dim3(SS_BUCKET_THREADS), 0, st, (const uint4*)sb.tiles, n, p, (const uint4*)sb.splitters,
                     (const uint8_t*)sb.cmat, keys_out, order_out, d_err);
}
