// Attainable matrix-core rate on this box: a pure chain-free MFMA loop (v_mfma_f32_32x32x16_bf16, four independent
// accumulators per wave), W waves per SIMD on every CU, for short (~0.3 ms) and long (~5 ms) launches -- the clock the
// card sustains under matrix load is what the "dense peak" of the roofline has to be read against.
// build: hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o tools/bin/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, float seed) {
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + threadIdx.x * 1e-3f); b[i] = (__bf16)(seed * 0.5f); }
  f32x16 acc[NACC];
  for (int n = 0; n < NACC; ++n) for (int i = 0; i < 16; ++i) acc[n][i] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[n], 0, 0, 0);
  }
  float s = 0.f;
  for (int n = 0; n < NACC; ++n) for (int i = 0; i < 16; ++i) s += acc[n][i];
  if (s == 12345.678f) out[0] = s;
}

template <int NACC>
void run(const char* tag, int blocks_per_cu, int iters) {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  float* out; hipMalloc(&out, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = cus * blocks_per_cu;
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(mfma_loop<NACC>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f);
  hipDeviceSynchronize();
  const int reps = 5;
  hipEventRecord(e0);
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(mfma_loop<NACC>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
  const double flops = (double)grid * 4 /*waves*/ * iters * 4.0 * NACC * 2.0 * 32 * 32 * 16;
  printf("%-28s CUs %d  %d wave(s)/SIMD  %8.3f ms  %7.1f TFLOP/s  (= %.2f GHz at 1024 flop/clk/SIMD)\n", tag, cus, blocks_per_cu, ms,
         flops / ms / 1e9, flops / ms / 1e9 * 1e12 / (cus * 4.0 * 1024.0) / 1e9);
  hipFree(out);
}

int main() {
  run<4>("short, 4 acc", 1, 600);
  run<4>("short, 4 acc", 2, 300);
  run<4>("long, 4 acc", 1, 12000);
  run<4>("long, 4 acc", 2, 6000);
  run<2>("long, 2 acc (dependent)", 1, 24000);
  run<4>("very long, 4 acc", 1, 120000);
  return 0;
}
