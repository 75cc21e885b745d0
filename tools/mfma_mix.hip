// What keeps ONE wave per SIMD from issuing back-to-back MFMAs?  The compute role of `tconv` (csrc/tconv.hip) in isolation:
// 2 x 4 accumulator tiles per wave, weight fragments through a 6-deep register ring from global memory, activation
// fragments through a 3-deep ring from LDS, running (tap, k-group) positions in scalar registers -- each ingredient behind a
// switch, at one and at two waves per SIMD.  No barriers, no memory role: what is measured is the steady state of the loop.
// build: hipcc --offload-arch=gfx950 -O3 tools/mfma_mix.hip -o tools/bin/mfma_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;
typedef __attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned u32x4;

constexpr int DA = 6, DB = 3, PD = DB - 1, MTW = 2, NTW = 4;
constexpr int ROWS = 456, ROWB = 80, LDS_BYTES = ROWS * ROWB;       // staged chunk of a 256-row tile with the 9-tap halo, V = 25
constexpr int NIT = 18;                                             // steps per item: 9 taps x 2 k-groups

template <bool GLD, bool LDSR, bool POS, int WPS, int RT = 0, bool BAR = false>   // RT: 1 = loop counts at run time, 2 = opaque row offsets, 3 = both
__global__ __launch_bounds__(256, WPS) void mix(const u32x4* __restrict__ W, int nfrag_steps, float* out, int chunks, int V, int nit_rt, int nkg_rt, int stag) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  for (int i = threadIdx.x; i < LDS_BYTES / 16; i += 256) {
    const unsigned v = 0x3c003c00u + (unsigned)(i & 0xff);
    reinterpret_cast<u32x4*>(smem)[i] = u32x4{v, v + 1, v + 2, v + 3};
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wr = wave >> 1, wm = wave & 1;
  f32x16 acc[MTW][NTW];
  for (int m = 0; m < MTW; ++m) for (int t = 0; t < NTW; ++t) for (int i = 0; i < 16; ++i) acc[m][t][i] = 0.f;
  int brow[NTW];
  for (int t = 0; t < NTW; ++t) {
    brow[t] = (wr * 128 + t * 32 + (lane & 31)) * ROWB + (lane >> 5) * 16;
    if (RT & 2) asm volatile("" : "+v"(brow[t]));      // as in the kernel: per-row table lookups, not base + constant
  }
  const int nit = (RT & 1) ? nit_rt : NIT, nkg = (RT & 1) ? nkg_rt : 2;
  const u32x4* abase = W + (size_t)(wm * MTW) * 64 + lane;
  const size_t astride = 4 * 64;                                     // four channel tiles per step
  const size_t alimit = (size_t)nfrag_steps * astride;
  size_t aoff = (size_t)((blockIdx.x * (unsigned)stag) % (unsigned)nfrag_steps) * astride;   // stag > 0: CUs walk the weights out of phase
  u32x4 a[DA][MTW], b[DB][NTW];
  for (int d = 0; d < DA; ++d) for (int m = 0; m < MTW; ++m) a[d][m] = u32x4{0x3c003c00u, 0x3c013c00u, 0x3c003c02u, 0x3c003c00u};
  for (int d = 0; d < DB; ++d) for (int t = 0; t < NTW; ++t) b[d][t] = u32x4{0x3c003c00u, 0x3c013c00u, 0x3c003c02u, 0x3c003c00u};
  auto load_a = [&](u32x4 (&dst)[MTW]) __attribute__((always_inline)) {
    if (GLD) {
#pragma unroll
      for (int m = 0; m < MTW; ++m) dst[m] = abase[aoff + (size_t)m * 64];
    }
    if (POS) { const size_t an = aoff + astride; aoff = an == alimit ? 0 : an; }
  };
  const int tapstep = V * ROWB - 32;
  int sb = 0, kgb = 0, soffb = 0;
  auto load_b = [&](u32x4 (&dst)[NTW]) __attribute__((always_inline)) {
    if (LDSR) {
#pragma unroll
      for (int t = 0; t < NTW; ++t) dst[t] = *reinterpret_cast<const u32x4*>(smem + brow[t] + soffb);
    }
    if (POS) {
      const bool adv = sb + 1 < nit;
      const bool wrap = kgb + 1 == nkg;
      soffb += adv ? (wrap ? tapstep : 32) : 0;
      kgb = adv ? (wrap ? 0 : kgb + 1) : kgb;
      sb += adv ? 1 : 0;
    }
  };
#pragma unroll
  for (int d = 0; d < DA - 1; ++d) load_a(a[d]);
  for (int item = 0; item < chunks; ++item) {
    sb = 0; kgb = 0; soffb = 0;
#pragma unroll
    for (int d = 0; d < PD; ++d) load_b(b[d]);
#define STEP(D)                                                                                          \
    {                                                                                                    \
      load_a(a[((D) + DA - 1) % DA]);                                                                    \
      load_b(b[((D) + PD) % DB]);                                                                        \
      _Pragma("unroll") for (int m = 0; m < MTW; ++m)                                                    \
        _Pragma("unroll") for (int t = 0; t < NTW; ++t)                                                  \
          acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[D][m]),       \
                                                              __builtin_bit_cast(bf16x8, b[(D) % DB][t]), acc[m][t], 0, 0, 0); \
      _Pragma("unroll") for (int i_ = 0; i_ < MTW * NTW; ++i_) {                                         \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                               \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                               \
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                               \
        __builtin_amdgcn_sched_group_barrier(0x006, 5, 0);                                               \
      }                                                                                                  \
      __builtin_amdgcn_sched_barrier(0);                                                                 \
    }
    for (int c = 0; c < nit / DA; ++c) { STEP(0) STEP(1) STEP(2) STEP(3) STEP(4) STEP(5) }
    if (BAR) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#undef STEP
  }
  float s = 0.f;
  for (int m = 0; m < MTW; ++m) for (int t = 0; t < NTW; ++t) for (int i = 0; i < 16; ++i) s += acc[m][t][i];
  if (s == 12345.678f) out[0] = s;
}

static u32x4* g_W;
static float* g_out;
static int g_cus;

template <bool GLD, bool LDSR, bool POS, int WPS, int RT = 0, bool BAR = false>
void run(const char* tag, int items, int stag = 0) {
  auto k = mix<GLD, LDSR, POS, WPS, RT, BAR>;
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  const int nfs = 8 * NIT;                       // 8 chunks x 18 steps of 4 KB: the 590 KB of a 256-channel layer's block
  const int grid = g_cus * WPS;
  const int it = items / WPS;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k, dim3(grid), dim3(256), LDS_BYTES, 0, g_W, nfs, g_out, it, 25, NIT, 2, stag);
  hipDeviceSynchronize();
  const int reps = 5;
  hipEventRecord(e0);
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k, dim3(grid), dim3(256), LDS_BYTES, 0, g_W, nfs, g_out, it, 25, NIT, 2, stag);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
  const double flops = (double)grid * 4 * it * NIT * 8.0 * 2.0 * 32 * 32 * 16;
  const double tf = flops / ms / 1e9;
  printf("%-44s %d wave/SIMD  %8.3f ms  %7.1f TFLOP/s   %5.1f cycles/MFMA at 2.4 GHz\n", tag, WPS, ms, tf,
         32.0 * (g_cus * 4.0 * 1024.0 * 2.4e9 / 1e12) / tf);
}

int main(int argc, char** argv) {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  g_cus = p.multiProcessorCount;
  const size_t wbytes = (size_t)8 * NIT * 4 * 64 * 16 + 4096;
  hipMalloc(&g_W, wbytes);
  unsigned* h = (unsigned*)malloc(wbytes);
  for (size_t i = 0; i < wbytes / 4; ++i) h[i] = 0x3c003c00u + (unsigned)(i & 0x7f);
  hipMemcpy(g_W, h, wbytes, hipMemcpyHostToDevice);
  hipMalloc(&g_out, 4);
  const int items = argc > 1 ? atoi(argv[1]) : 64;     // 64 items = 8 tiles of a 256-channel layer (~125 us of MFMA time)
  run<false, false, false, 1>("MFMAs only", items);
  run<false, false, true, 1>("+ running positions (SALU)", items);
  run<true, false, false, 1>("+ weight ring, fragment 0 (L1)", items);
  run<true, false, true, 1>("+ weight ring, walking 590 KB (L2)", items);
  run<false, true, false, 1>("+ activation ring, offset 0 (LDS)", items);
  run<false, true, true, 1>("+ activation ring, walking taps (LDS)", items);
  run<true, true, true, 1>("everything (the tconv compute loop)", items);
  run<true, false, true, 1>("+ weight ring (L2), CUs out of phase", items, 7);
  run<true, true, true, 1>("everything, CUs out of phase", items, 7);
  run<true, true, true, 1, 1>("everything, run-time loop counts", items);
  run<true, true, true, 1, 2>("everything, opaque row offsets", items);
  run<true, true, true, 1, 3>("everything, run-time loop parameters", items);
  run<true, true, true, 1, 0, true>("everything + barrier per item", items);
  run<true, true, true, 1, 3, true>("everything, run-time parameters + barrier", items);
  run<false, false, false, 2>("MFMAs only", items);
  run<true, false, true, 2>("+ weight ring, walking 590 KB (L2)", items);
  run<false, true, true, 2>("+ activation ring, walking taps (LDS)", items);
  run<true, true, true, 2>("everything (the tconv compute loop)", items);
  return 0;
}
