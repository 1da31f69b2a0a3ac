// VALU issue-rate probe for gfx950: v_fma_f32 vs v_pk_fma_f32 (is packed fp32 twice the work per issue slot?)
// hipcc --offload-arch=gfx950 -O3 pk_rate.hip -o pk_rate && ./pk_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int PK>
__global__ __launch_bounds__(256) void k(float* out, float a, float b, int iters) {
  f2 x0 = {a + threadIdx.x, a}, x1 = {b, a}, x2 = {a, b}, x3 = {b, b + 1}, x4 = {a + 2, b}, x5 = {a, b + 3}, x6 = {1, a}, x7 = {b, 2};
  const f2 m = {a, b}, c = {b, a};
  for (int i = 0; i < iters; ++i) {
    if (PK) {
#define STEP(q) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(q) : "v"(m), "v"(c));
      STEP(x0) STEP(x1) STEP(x2) STEP(x3) STEP(x4) STEP(x5) STEP(x6) STEP(x7)
#undef STEP
    } else {
#define STEP(q) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(q.x) : "v"(m.x), "v"(c.x));
      STEP(x0) STEP(x1) STEP(x2) STEP(x3) STEP(x4) STEP(x5) STEP(x6) STEP(x7)
#undef STEP
    }
  }
  const f2 s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}
int main() {
  float* d; hipMalloc(&d, 1024 * 8 * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000, blocks = 256 * 8;   // 8 workgroups of 4 waves per CU = 8 waves per SIMD
  for (int pk = 0; pk < 2; ++pk)
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (pk) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d, 1.0f, 0.5f, iters);
      else hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d, 1.0f, 0.5f, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double insts = (double)blocks * 4 * iters * 8;   // wave-level instructions
      if (rep) printf("%s: %.3f ms, %.2f G wave-instructions/s, %.1f TFLOP/s\n", pk ? "v_pk_fma_f32" : "v_fma_f32   ", ms, insts / ms / 1e6, insts * 64 * 2 * (pk ? 2 : 1) / ms / 1e9);
    }
  return 0;
}
