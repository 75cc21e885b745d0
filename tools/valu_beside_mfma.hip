// What does a vector-ALU wave get to issue next to a wave that issues back-to-back MFMAs on the same SIMD?
// One 512-thread workgroup per CU: four "matrix" waves (one per SIMD) run a chain-free v_mfma_f32_32x32x16_bf16 loop,
// four "vector" waves run independent v_fma_f32 chains (the shape of tconv's memory role: BatchNorm + ReLU transform of
// the staged chunk).  Cycles (s_memtime) of one wave of each role, alone and together, with
//   PAD   idle cycles (s_nop) in the matrix wave after every MFMA,
//   PRIO  s_setprio 3 for the vector waves (1) or the matrix waves (2),
//   SWAP  the vector role on waves 0-3 (the older half), the matrix role on waves 4-7.
// build: hipcc --offload-arch=gfx950 -O3 tools/valu_beside_mfma.hip -o tools/bin/valu_beside_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;

template <int PAD, int PRIO, bool SWAP, int FILL, int OP = 0>
__global__ __launch_bounds__(512, 2) void k(unsigned long long* out, int mfma_iters, int valu_iters, float seed) {
  extern __shared__ float lds[];
  const int wave = threadIdx.x >> 6;
  const bool matrix = SWAP ? wave >= 4 : wave < 4;
  const bool active_m = mfma_iters > 0, active_v = valu_iters > 0;
  if (PRIO == 1 && !matrix) __builtin_amdgcn_s_setprio(3);
  if (PRIO == 2 && matrix) __builtin_amdgcn_s_setprio(3);
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (matrix) {
    if (!active_m) return;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + threadIdx.x * 1e-3f); b[i] = (__bf16)(seed * 0.5f); }
    f32x16 acc[4];
    for (int n = 0; n < 4; ++n) for (int i = 0; i < 16; ++i) acc[n][i] = 0.f;
    float* lp = lds + threadIdx.x * 4;
    float4 f = {0, 0, 0, 0};
    for (int it = 0; it < mfma_iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int n = 0; n < 4; ++n) {
          // the MFMA and what follows it in ONE asm statement: the placement is what is being measured
          if constexpr (FILL == 1)        // an LDS read after each MFMA (what the real loop has in its gaps)
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n\tds_read_b128 %1, %4" : "+v"(acc[n]), "=v"(f) : "v"(a), "v"(b), "v"((unsigned)(size_t)lp));
          else
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[n]) : "v"(a), "v"(b));
          if constexpr (PAD == 1) asm volatile("s_nop 3");
          if constexpr (PAD == 2) asm volatile("s_nop 7");
          if constexpr (PAD == 3) asm volatile("s_nop 7\n\ts_nop 3");
          if constexpr (PAD == 4) asm volatile("s_nop 7\n\ts_nop 7");
          if constexpr (PAD == 5) asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3");
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = f.x;
    for (int n = 0; n < 4; ++n) for (int i = 0; i < 16; ++i) s += acc[n][i];
    if (s == 12345.678f) out[15] = 1;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0 && (wave & 3) == 0) { out[0] = t1 - t0; out[1] = (unsigned long long)mfma_iters * 16; }
  } else {
    if (!active_v) return;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = seed + i + threadIdx.x;
    f32x2 vv[4];
    for (int i = 0; i < 4; ++i) vv[i] = f32x2{seed + i, seed - i};
    typedef float f32x4v __attribute__((ext_vector_type(4)));
    f32x4v w4 = {seed, seed, seed, seed};
    const float c0 = seed * 0.999f, c1 = seed * 0.001f;
    const f32x2 cc0 = {c0, c0}, cc1 = {c1, c1};
    for (int it = 0; it < valu_iters; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          if constexpr (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(c0), "v"(c1));
          if constexpr (OP == 1) { if (i & 1) continue; asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(vv[i >> 1]) : "v"(cc0), "v"(cc1)); }
          if constexpr (OP == 2) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c0));
          if constexpr (OP == 3) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(v[i]) : "v"(c0));
          if constexpr (OP == 4) asm volatile("v_lshlrev_b32 %0, 16, %0" : "+v"(v[i]));
          if constexpr (OP == 5) { if (i & 1) continue; asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(vv[i >> 1]) : "v"(cc0)); }
          if constexpr (OP == 6) { if (i & 3) continue; asm volatile("ds_write_b128 %0, %1" :: "v"((unsigned)((threadIdx.x & 255) * 16)), "v"(w4)); }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += v[i];
    for (int i = 0; i < 4; ++i) s += vv[i][0] + vv[i][1];
    if (s == 12345.678f) out[15] = 2;
    asm volatile("s_waitcnt lgkmcnt(0)");
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0 && (wave & 3) == 0) { out[2] = __builtin_amdgcn_s_memtime() - t0; out[3] = (unsigned long long)valu_iters * (OP == 1 || OP == 5 ? 32 : OP == 6 ? 16 : 64); }
  }
}

template <int PAD, int PRIO, bool SWAP, int FILL, int OP = 0>
void run(const char* tag, int mi, int vi) {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  unsigned long long* out; hipMalloc(&out, 16 * 8); hipMemset(out, 0, 16 * 8);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<PAD, PRIO, SWAP, FILL, OP>), dim3(p.multiProcessorCount), dim3(512), 8192, 0, out, mi, vi, 1.0f);
  hipDeviceSynchronize();
  unsigned long long h[16]; hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-44s", tag);
  if (mi) printf(" matrix wave: %7.1f cycles/MFMA", (double)h[0] / h[1]); else printf(" %34s", "");
  if (vi) printf("   vector wave: %6.2f cycles/VALU", (double)h[2] / h[3]);
  printf("\n");
  hipFree(out);
}

// 12 waves per workgroup: one matrix wave and TWO vector waves per SIMD (does a second vector wave add issue slots?)
__global__ __launch_bounds__(768, 3) void k12(unsigned long long* out, int mfma_iters, int valu_iters, float seed) {
  const int wave = threadIdx.x >> 6;
  const bool matrix = wave < 4;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (matrix) {
    if (mfma_iters <= 0) return;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + threadIdx.x * 1e-3f); b[i] = (__bf16)(seed * 0.5f); }
    f32x16 acc[4];
    for (int n = 0; n < 4; ++n) for (int i = 0; i < 16; ++i) acc[n][i] = 0.f;
    for (int it = 0; it < mfma_iters; ++it)
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int n = 0; n < 4; ++n) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[n]) : "v"(a), "v"(b));
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int n = 0; n < 4; ++n) for (int i = 0; i < 16; ++i) s += acc[n][i];
    if (s == 12345.678f) out[15] = 1;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0 && wave == 0) { out[0] = t1 - t0; out[1] = (unsigned long long)mfma_iters * 16; }
  } else {
    if (valu_iters <= 0) return;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = seed + i + threadIdx.x;
    const float c0 = seed * 0.999f, c1 = seed * 0.001f;
    for (int it = 0; it < valu_iters; ++it)
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(c0), "v"(c1));
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += v[i];
    if (s == 12345.678f) out[15] = 2;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0 && wave == 4) { out[2] = t1 - t0; out[3] = (unsigned long long)valu_iters * 64; }
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0 && wave == 8) { out[4] = t1 - t0; }
  }
}

void run12(const char* tag, int mi, int vi) {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  unsigned long long* out; hipMalloc(&out, 16 * 8); hipMemset(out, 0, 16 * 8);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k12, dim3(p.multiProcessorCount), dim3(768), 0, 0, out, mi, vi, 1.0f);
  hipDeviceSynchronize();
  unsigned long long h[16]; hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-44s", tag);
  if (mi) printf(" matrix wave: %7.1f cycles/MFMA", (double)h[0] / h[1]); else printf(" %34s", "");
  if (vi) printf("   vector wave 4: %6.2f, wave 8: %6.2f cycles/VALU", (double)h[2] / h[3], (double)h[4] / h[3]);
  printf("\n");
  hipFree(out);
}

int main() {
  run12("12 waves: two vector waves per SIMD, alone", 0, 4000);
  run12("12 waves: matrix + two vector waves per SIMD", 4000, 4000);

  const int MI = 4000, VI = 4000;     // 64000 MFMAs = 2.05 M cycles alone; 256000 VALU = 1.0 M cycles alone at 4 cycles each
  run<0, 0, false, 0>("matrix waves alone", MI, 0);
  run<0, 0, false, 0>("vector waves alone", 0, VI);
  run<0, 0, false, 0>("both, no padding", MI, VI);
  run<0, 1, false, 0>("both, vector waves at s_setprio 3", MI, VI);
  run<0, 2, false, 0>("both, matrix waves at s_setprio 3", MI, VI);
  run<0, 0, true, 0>("both, vector role on waves 0-3 (older)", MI, VI);
  run<1, 0, false, 0>("both, s_nop 3 after each MFMA", MI, VI);
  run<2, 0, false, 0>("both, s_nop 7 after each MFMA", MI, VI);
  run<3, 0, false, 0>("both, 12 idle cycles after each MFMA", MI, VI);
  run<4, 0, false, 0>("both, 16 idle cycles after each MFMA", MI, VI);
  run<5, 0, false, 0>("both, 20 idle cycles after each MFMA", MI, VI);
  run<4, 0, false, 0>("matrix alone, 16 idle cycles", MI, 0);
  run<5, 0, false, 0>("matrix alone, 20 idle cycles", MI, 0);
  run<0, 0, false, 1>("both, an LDS read after each MFMA", MI, VI);
  run<2, 0, false, 1>("both, LDS read + s_nop 7 after each MFMA", MI, VI);
  run<4, 1, false, 0>("both, 16 idle cycles + vector prio", MI, VI);
  // which vector instructions: cycles per INSTRUCTION of the vector wave (younger half), alone and beside the MFMA wave
  run<0, 0, false, 0, 1>("v_pk_fma_f32 alone", 0, VI);
  run<0, 0, false, 0, 1>("v_pk_fma_f32 beside MFMAs", MI, VI);
  run<0, 0, true, 0, 1>("v_pk_fma_f32 beside MFMAs, vector waves older", MI, VI);
  run<0, 0, false, 0, 5>("v_pk_add_f32 alone", 0, VI);
  run<0, 0, false, 0, 5>("v_pk_add_f32 beside MFMAs", MI, VI);
  run<0, 0, false, 0, 2>("v_cvt_pk_bf16_f32 alone", 0, VI);
  run<0, 0, false, 0, 2>("v_cvt_pk_bf16_f32 beside MFMAs", MI, VI);
  run<0, 0, false, 0, 3>("v_pk_max_i16 alone", 0, VI);
  run<0, 0, false, 0, 3>("v_pk_max_i16 beside MFMAs", MI, VI);
  run<0, 0, false, 0, 4>("v_lshlrev_b32 alone", 0, VI);
  run<0, 0, false, 0, 4>("v_lshlrev_b32 beside MFMAs", MI, VI);
  run<0, 0, false, 0, 6>("ds_write_b128 alone", 0, VI);
  run<0, 0, false, 0, 6>("ds_write_b128 beside MFMAs", MI, VI);
  return 0;
}
