// Shared device helpers for the IST-GCN gfx950 kernels (CDNA4 only: wave64, MFMA 32x32).
//
// Activation layout everywhere in this library ("NTVC"): x[n][t][v][c], c innermost, one
// sequence n = one (clip, person) pair.  A frame (t) is V*C contiguous elements, so a run of
// frames is one contiguous HBM stream and the channel axis sits on the MFMA k / lane axes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define ISTGCN_OK 0
#define ISTGCN_EINVAL 1
#define ISTGCN_ELAUNCH 2

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// ---------------------------------------------------------------------------------------
// Element traits.  One "k-group" = the 16 bytes of contraction index one lane feeds to the
// matrix core: 8 bf16 (one v_mfma_f32_32x32x16_bf16) or 4 f32 (four v_mfma_f32_32x32x2_f32,
// lane-half h contributing k = 4h+s to step s).  Both operands use the same slot -> k map,
// so the contraction is exact whatever the order.
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
  __device__ static inline float to_f(__bf16 v) { return (float)v; }
  __device__ static inline __bf16 from_f(float v) { return (__bf16)v; }
};

__device__ static inline void mma_kgroup(f32x16& acc, const f32x4& a, const f32x4& b) {
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], b[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], b[2], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], b[3], acc, 0, 0, 0);
}
__device__ static inline void mma_kgroup(f32x16& acc, const bf16x8& a, const bf16x8& b) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
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

__device__ static inline void atomic_add_f64(double* p, double v) { unsafeAtomicAdd(p, v); }

static inline int round_up(int a, int b) { return (a + b - 1) / b * b; }
static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// hipGetLastError() is per-thread and sticky across *any* HIP call of the host process (PyTorch's allocator polls
// events and leaves hipErrorNotReady behind): clear it before our launch so the check below sees only our own.
#define ISTGCN_LAUNCH(...)            \
  do {                                \
    (void)hipGetLastError();          \
    hipLaunchKernelGGL(__VA_ARGS__);  \
  } while (0)

#define ISTGCN_CHECK_LAUNCH()                         \
  do {                                                \
    hipError_t e_ = hipGetLastError();                \
    if (e_ != hipSuccess) return 1000 + (int)e_;      \
  } while (0)
