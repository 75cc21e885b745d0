// Shared device helpers for the IST-GCN gfx950 kernels (CDNA4 only: wave64, MFMA 32x32).
//
// Activation layout everywhere in this library ("NTVC"): x[n][t][v][c], c innermost, one
// sequence n = one (clip, person) pair.  A frame (t) is V*C contiguous elements, so a run of
// frames is one contiguous HBM stream and the channel axis sits on the MFMA k / lane axes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>

// Wave-specialised kernels (one 8-wave workgroup per CU, two roles): which HALF of the workgroup gets which role.  The two
// waves of a SIMD compete for its vector-issue port and the port goes to the OLDER wave (waves 0-3; s_setprio does not
// change it: tools/valu_beside_mfma.hip).  256 = the role code written for waves 4-7 runs on waves 0-3 and vice versa
// (a relabelling of the thread index, nothing else changes).
#ifndef ISTGCN_ROLE_FLIP
#define ISTGCN_ROLE_FLIP 0
#endif

#define ISTGCN_OK 0
#define ISTGCN_EINVAL 1
#define ISTGCN_ELAUNCH 2

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// ---------------------------------------------------------------------------------------
// Element traits.  One "k-group" = the 16 bytes of contraction index one lane feeds to the
// matrix core: 8 bf16 / fp16 (one v_mfma_f32_32x32x16_{bf16,f16}) or 4 f32 (four v_mfma_f32_32x32x2_f32,
// lane-half h contributing k = 4h+s to step s).  Both operands use the same slot -> k map,
// so the contraction is exact whatever the order.  dtype codes of the C ABI: 0 = f32, 1 = bf16, 2 = fp16;
// the two 16-bit types share every tiling constant (geometry functions treat dtype != 0 alike).
// ---------------------------------------------------------------------------------------
template <typename T> struct Elem;
template <> struct Elem<float> {
  static constexpr int EPL = 4;        // elements per 16-byte lane fragment
  static constexpr int KGS = 8;        // contraction indices consumed per k-group (2 halves x EPL)
  static constexpr int CC = 32;        // input channels per LDS chunk
  typedef f32x4 frag;
  __device__ static inline float to_f(float v) { return v; }
  __device__ static inline float from_f(float v) { return v; }
};
template <> struct Elem<__bf16> {
  static constexpr int EPL = 8;
  static constexpr int KGS = 16;
  static constexpr int CC = 64;
  typedef bf16x8 frag;
  typedef bf16x4 frag4;
  __device__ static inline float to_f(__bf16 v) { return (float)v; }
  __device__ static inline __bf16 from_f(float v) { return (__bf16)v; }
};
template <> struct Elem<_Float16> {
  static constexpr int EPL = 8;
  static constexpr int KGS = 16;
  static constexpr int CC = 64;
  typedef f16x8 frag;
  typedef f16x4 frag4;
  __device__ static inline float to_f(_Float16 v) { return (float)v; }
  __device__ static inline _Float16 from_f(float v) { return (_Float16)v; }
};

static inline bool istgcn_dtype_ok(int dtype) { return dtype >= 0 && dtype <= 2; }

__device__ static inline void mma_kgroup(f32x16& acc, const f32x4& a, const f32x4& b) {
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], b[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], b[2], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], b[3], acc, 0, 0, 0);
}
__device__ static inline void mma_kgroup(f32x16& acc, const bf16x8& a, const bf16x8& b) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
}
__device__ static inline void mma_kgroup(f32x16& acc, const f16x8& a, const f16x8& b) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
}

// 16-bit element types: transposed LDS read (ds_read_b64_tr_b16, identical for both 16-bit formats) of the two 4-row
// blocks a lane addresses -> one MFMA operand fragment; and an 8-element dot product on v_dot2 (no conversions).
template <typename T>
__device__ static inline typename Elem<T>::frag tr_pair(const T* lo_addr, const T* hi_addr) {
  typedef typename Elem<T>::frag4 frag4;
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)lo_addr);
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)hi_addr);
  frag4 l4 = __builtin_bit_cast(frag4, lo), h4 = __builtin_bit_cast(frag4, hi);
  typename Elem<T>::frag o;
  o[0] = l4[0]; o[1] = l4[1]; o[2] = l4[2]; o[3] = l4[3];
  o[4] = h4[0]; o[5] = h4[1]; o[6] = h4[2]; o[7] = h4[3];
  return o;
}
__device__ static inline float dot8(const bf16x8& a, const bf16x8& b, float s) {
#pragma unroll
  for (int e = 0; e < 8; e += 2) {
    const bf16x2 a2 = {a[e], a[e + 1]}, b2 = {b[e], b[e + 1]};
    s = __builtin_amdgcn_fdot2_f32_bf16(a2, b2, s, false);
  }
  return s;
}
__device__ static inline float dot8(const f16x8& a, const f16x8& b, float s) {
#pragma unroll
  for (int e = 0; e < 8; e += 2) {
    const f16x2 a2 = {a[e], a[e + 1]}, b2 = {b[e], b[e + 1]};
    s = __builtin_amdgcn_fdot2(a2, b2, s, false);
  }
  return s;
}

// D tile of a 32x32 MFMA: lane l holds column (l & 31); register r holds row
//   (r & 3) + 8 * (r >> 2) + 4 * (l >> 5).
__device__ static inline int mfma_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

template <typename T> __device__ static inline void zero_frag(typename Elem<T>::frag& f) {
#pragma unroll
  for (int j = 0; j < Elem<T>::EPL; ++j) f[j] = Elem<T>::from_f(0.f);
}

// 4 consecutive channels of one row -> LDS (one ds_write_b128 / ds_write_b64)
__device__ static inline void store4(float* dst, const float (&v)[4]) {
  f32x4 o = {v[0], v[1], v[2], v[3]};
  *reinterpret_cast<f32x4*>(dst) = o;
}
__device__ static inline void store4(__bf16* dst, const float (&v)[4]) {
  bf16x4 o = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
  *reinterpret_cast<bf16x4*>(dst) = o;
}
__device__ static inline void store4(_Float16* dst, const float (&v)[4]) {
  f16x4 o = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
  *reinterpret_cast<f16x4*>(dst) = o;
}

// ---------------------------------------------------------------------------------------------------------
// Tile staging HBM/L2 -> LDS.  Copies an [R rows] x [Q 16-byte vectors] block whose global rows are `gstride`
// elements apart into LDS rows `lstride` elements apart.  All U loads of a batch are issued before the first one
// is consumed, so a 256-thread workgroup keeps U*4 KiB in flight (the serial load->use loop this replaces kept 4 KiB
// and was pure memory latency).  Rows outside [r_lo, r_hi) and channels >= c_lim are written as zeros; an optional
// per-channel affine (+ReLU) -- the BatchNorm in front of the consumer -- is applied on the way in.
// VEC = rows are 16-byte aligned and whole vectors are either inside or outside c_lim.
// ---------------------------------------------------------------------------------------------------------
template <typename T, int U, bool VEC>
__device__ static inline void stage_block(const T* __restrict__ g, size_t gstride, int c_lim, T* lds, int lstride,
                                          int R, int r_lo, int r_hi, int Q, const float* __restrict__ sc,
                                          const float* __restrict__ sh, int relu, int tid, int nthreads) {
  using E = Elem<T>;
  constexpr int EPL = E::EPL;
  typedef typename E::frag frag_t;
  const bool pow2 = (Q & (Q - 1)) == 0;
  const int lq = 31 - __builtin_clz(Q);
  if (pow2 && Q <= nthreads) {
    // a thread keeps one channel vector q and walks rows r0, r0+RS, ...: one add per item instead of a div/mod,
    // and the per-channel affine lives in registers
    const int q = tid & (Q - 1), r0 = tid >> lq, RS = nthreads >> lq;
    const bool qlive = q * EPL < c_lim;
    float scv[EPL], shv[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) { scv[e] = 0.f; shv[e] = 0.f; }
    if (sc) {
      if (VEC) {
        if (qlive) {                         // whole 16-byte loads: one branch, not one per element
#pragma unroll
          for (int e4 = 0; e4 < EPL; e4 += 4) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(sc + q * EPL + e4);
            const f32x4 b = *reinterpret_cast<const f32x4*>(sh + q * EPL + e4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { scv[e4 + e] = a[e]; shv[e4 + e] = b[e]; }
          }
        }
      } else {
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
          if (q * EPL + e < c_lim) { scv[e] = sc[q * EPL + e]; shv[e] = sh[q * EPL + e]; }
        }
      }
    }
    const T* src = g + (size_t)r0 * gstride + q * EPL;
    T* dst = lds + r0 * lstride + q * EPL;
    const size_t gstep = (size_t)RS * gstride;
    const int lstep = RS * lstride;
    for (int r = r0; r < R; r += RS * U) {
      frag_t v[U];
      bool live[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int rr = r + u * RS;
        live[u] = qlive && rr >= r_lo && rr < r_hi && rr < R;
        zero_frag<T>(v[u]);
        if (live[u]) {
          if (VEC) v[u] = *reinterpret_cast<const frag_t*>(src + u * gstep);
          else {
#pragma unroll
            for (int e = 0; e < EPL; ++e) if (q * EPL + e < c_lim) v[u][e] = src[u * gstep + e];
          }
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (r + u * RS < R) {
          if (sc && live[u]) {
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
              if (VEC || q * EPL + e < c_lim) {
                float fv = E::to_f(v[u][e]) * scv[e] + shv[e];
                if (relu) fv = fmaxf(fv, 0.f);
                v[u][e] = E::from_f(fv);
              }
            }
          }
          *reinterpret_cast<frag_t*>(dst + u * lstep) = v[u];
        }
      }
      src += (size_t)U * gstep;
      dst += U * lstep;
    }
    return;
  }
  const int tot = R * Q;
  for (int base = tid; base < tot; base += nthreads * U) {
    frag_t v[U];
    int rr[U], qq[U];
    bool live[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int it = base + u * nthreads;
      const int r = it / Q, q = it - r * Q;
      rr[u] = r; qq[u] = q;
      live[u] = it < tot && r >= r_lo && r < r_hi && q * EPL < c_lim;
      zero_frag<T>(v[u]);
      if (live[u]) {
        const T* src = g + (size_t)r * gstride + q * EPL;
        if (VEC) v[u] = *reinterpret_cast<const frag_t*>(src);
        else {
#pragma unroll
          for (int e = 0; e < EPL; ++e) if (q * EPL + e < c_lim) v[u][e] = src[e];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int it = base + u * nthreads;
      if (it < tot) {
        if (sc && live[u]) {
#pragma unroll
          for (int e = 0; e < EPL; ++e) {
            if (VEC || qq[u] * EPL + e < c_lim) {
              float fv = E::to_f(v[u][e]) * sc[qq[u] * EPL + e] + sh[qq[u] * EPL + e];
              if (relu) fv = fmaxf(fv, 0.f);
              v[u][e] = E::from_f(fv);
            }
          }
        }
        *reinterpret_cast<frag_t*>(lds + rr[u] * lstride + qq[u] * EPL) = v[u];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// Register-ring software pipeline for an MFMA loop whose operand fragments come from L2/LDS: the loads of step
// it + D - 1 are issued before the MFMAs of step it, so D - 1 steps of matrix work cover the load latency (one bf16
// step is only 128-256 MFMA cycles, an L2 hit 500-900).  `load(it, a, b)` fills one ring slot, `mma(a, b)` consumes
// it.  No data-dependent control flow surrounds the loads (out-of-range steps re-load the last step), so the
// compiler's vmcnt/lgkmcnt bookkeeping is exact; sched_barrier pins the issue order.
// ---------------------------------------------------------------------------------------------------------
template <int D, int NA, int NB, typename FragT, typename LoadF, typename MmaF>
__device__ static inline void mfma_ring(int nit, LoadF load, MmaF mma) {
  FragT a[D][NA], b[D][NB];
#pragma unroll
  for (int d = 0; d < D - 1; ++d) load(min(d, nit - 1), a[d], b[d]);
  for (int it0 = 0; it0 < nit; it0 += D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      load(min(it0 + d + D - 1, nit - 1), a[(d + D - 1) % D], b[(d + D - 1) % D]);
      __builtin_amdgcn_sched_barrier(0);
      if (it0 + d < nit) mma(a[d], b[d]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// The same ring with the two operand streams split: `loadA` (weights, global/L2 -- independent of the LDS tile) can be
// primed BEFORE the phase that produces the LDS operand (ring_prime_a), so the ring-fill latency (an L2 round trip
// before the first MFMA -- most of a 12-step loop) overlaps that phase; `loadB` reads LDS.
template <int D, int NA, int NB, typename FragT>
struct MfmaRing {
  FragT a[D][NA], b[D][NB];
};

template <int D, int NA, int NB, typename FragT, typename LoadA>
__device__ static inline void ring_prime_a(MfmaRing<D, NA, NB, FragT>& R, int nit, LoadA loadA) {
#pragma unroll
  for (int d = 0; d < D - 1; ++d) loadA(min(d, nit - 1), R.a[d]);
}

template <int D, int NA, int NB, typename FragT, typename LoadA, typename LoadB, typename MmaF>
__device__ static inline void ring_run(MfmaRing<D, NA, NB, FragT>& R, int nit, LoadA loadA, LoadB loadB, MmaF mma) {
#pragma unroll
  for (int d = 0; d < D - 1; ++d) loadB(min(d, nit - 1), R.b[d]);
  for (int it0 = 0; it0 < nit; it0 += D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const int nx = min(it0 + d + D - 1, nit - 1);
      loadA(nx, R.a[(d + D - 1) % D]);
      loadB(nx, R.b[(d + D - 1) % D]);
      __builtin_amdgcn_sched_barrier(0);
      if (it0 + d < nit) mma(R.a[d], R.b[d]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

__device__ static inline void atomic_add_f64(double* p, double v) { unsafeAtomicAdd(p, v); }

static inline int round_up(int a, int b) { return (a + b - 1) / b * b; }
static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// hipGetLastError() is per-thread and sticky across *any* HIP call of the host process (PyTorch's allocator polls
// events and leaves hipErrorNotReady behind): clear it before our launch so the check below sees only our own.
// Dispatch trace (trace.hip; test instrumentation, off unless istgcn_trace(1) switched it on): every launch of the library
// goes through this macro, so with the trace on the set of kernel symbols a call really launched can be read back --
// tests assert which variant served a shape instead of trusting the dispatch predicates.
extern "C" int istgcn_trace_on();
extern "C" void istgcn_trace_note(const void* kfn, const char* launcher);
#define ISTGCN_LAUNCH(k, ...)                                                        \
  do {                                                                               \
    (void)hipGetLastError();                                                         \
    if (istgcn_trace_on()) istgcn_trace_note((const void*)(k), __PRETTY_FUNCTION__); \
    hipLaunchKernelGGL(k, __VA_ARGS__);                                              \
  } while (0)

// Workgroups of `kfn` resident on the whole device at once (occupancy x CUs).  The persistent kernels launch exactly
// that many: every extra workgroup repeats the per-workgroup prologue (adjacency tables) / flush and waits for a slot
// anyway.  Cached per (kernel, block size, LDS bytes); thread_local so DataParallel's per-device threads never race.
inline int istgcn_resident_blocks(const void* kfn, int threads, size_t lds) {
  struct Entry { const void* k; int threads; int dev; size_t lds; int blocks; };
  thread_local Entry cache[64];
  thread_local int n = 0;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  for (int i = 0; i < n; ++i)
    if (cache[i].k == kfn && cache[i].threads == threads && cache[i].lds == lds && cache[i].dev == dev)
      return cache[i].blocks;
  int cus = 0, occ = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kfn, threads, lds) != hipSuccess || occ < 1) occ = 1;
  (void)hipGetLastError();
  const int blocks = occ * cus;
  if (n < 64) cache[n++] = Entry{kfn, threads, dev, lds, blocks};
  return blocks;
}

// Opt a kernel into > 64 KiB of dynamic LDS.  The attribute is per DEVICE (a single-process multi-GPU caller --
// nn.DataParallel -- launches the same kernel on several devices from several threads), so the "done" flag is one bit
// per device ordinal in an atomic mask owned by the call site.  Returns 0 or 2000 + hipError.
inline int istgcn_lds_optin(const void* kfn, std::atomic<unsigned long long>& done) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  const unsigned long long bit = 1ull << (dev & 63);
  if (done.load(std::memory_order_acquire) & bit) return 0;
  hipError_t e = hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e != hipSuccess) return 2000 + (int)e;
  done.fetch_or(bit, std::memory_order_release);
  return 0;
}

#define ISTGCN_CHECK_LAUNCH()                         \
  do {                                                \
    hipError_t e_ = hipGetLastError();                \
    if (e_ != hipSuccess) return 1000 + (int)e_;      \
  } while (0)
