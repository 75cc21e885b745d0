// Shared pieces of the register-chained graph-conv kernels (gcn_rc.hip: forward, gcn_rc_bwd.hip: data gradient + adjacency
// gradient, gcn_rc_wgrad.hip: weight gradient).  See gcn_rc.hip for the scheme.
#pragma once
#include "common.hpp"

constexpr int RC_NTH = 512;
constexpr int IMG_RS = 68;                    // dwords per pair-row of a wave's output image (64 channels + 4: rows 4 banks apart)
constexpr int IMG_BYTES = 16 * IMG_RS * 4;    // 16 pair-rows

// Memory operations go through buffer descriptors that cover exactly ONE frame (V rows): rows >= V of the padded 32-row
// tile fall outside the descriptor and the hardware's bounds check makes their loads return zero and drops their stores.
// No lane predicate, no branch around a memory operation: the number of loads and stores per frame is a constant, so the
// compiler's s_waitcnt vmcnt(N) for the prefetched frame counts exactly the operations issued since (a predicated store
// would turn it into "wait for everything", i.e. one store round trip per frame -- DESIGN.md, compiler pitfalls 4 and 11).
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ static inline rsrc_t make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}

template <typename T> __device__ static inline uint32_t pack2(float a, float b);
template <> __device__ inline uint32_t pack2<__bf16>(float a, float b) {
  bf16x2 p = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(uint32_t, p);
}
template <> __device__ inline uint32_t pack2<_Float16>(float a, float b) {
  f16x2 p = {(_Float16)a, (_Float16)b};
  return __builtin_bit_cast(uint32_t, p);
}
template <typename T> __device__ static inline void unpack2(uint32_t p, float& lo, float& hi);
template <> __device__ inline void unpack2<__bf16>(uint32_t p, float& lo, float& hi) {
  lo = __builtin_bit_cast(float, p << 16);
  hi = __builtin_bit_cast(float, p & 0xffff0000u);
}
template <> __device__ inline void unpack2<_Float16>(uint32_t p, float& lo, float& hi) {
  const f16x2 v = __builtin_bit_cast(f16x2, p);
  lo = (float)v[0];
  hi = (float)v[1];
}

