// VALU issue rate on gfx950 (dev probe): independent v_fma_f32 / v_fma_f64 / v_cmp+v_cndmask streams at 1, 2, 4, 8 waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  float a[8]; double d[8];
  for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 1e-3f + i; d[i] = a[i]; }
  const float m = 1.0000001f, c = 1e-9f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (MODE == 0) a[i] = __builtin_fmaf(a[i], m, c);
        else if (MODE == 1) d[i] = __builtin_fma(d[i], (double)m, (double)c);
        else a[i] = (a[i] > (float)it) ? a[i] * m : a[i] + c;
      }
    }
  }
  float s = 0; for (int i = 0; i < 8; ++i) s += a[i] + (float)d[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE> void run(const char* name, int wps) {
  const int iters = 20000, blocks = 256 * wps;          // 256-thread blocks = 4 waves = 1 per SIMD; wps blocks per CU
  float* out; hipMalloc(&out, (size_t)blocks * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 100);
  hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters); hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double instr_per_wave = (double)iters * 64 * (MODE == 2 ? 2.0 : 1.0);
  const double per_simd = instr_per_wave * wps;          // wave-instructions each SIMD issued
  printf("%-14s %d waves/SIMD: %.3f ms -> %.2f cycles per wave-instruction per SIMD at 2.4 GHz\n", name, wps, ms, ms * 1e-3 * 2.4e9 / per_simd);
  hipFree(out);
}
int main() {
  for (int w : {1, 2, 4, 8}) run<0>("v_fma_f32", w);
  for (int w : {1, 2, 4, 8}) run<1>("v_fma_f64", w);
  for (int w : {1, 2, 4}) run<2>("cmp+cndmask", w);
  return 0;
}
