// Counter-based dropout shared by the glue kernels (pointwise.hip) and the fused BatchNorm-backward input of the bottleneck
// chain (bneck_rc.hip): Philox4x32 (7 rounds since round 5) keyed by (seed, element index / 8); the backward pass regenerates
// the forward mask.
#pragma once
#include "common.hpp"

// Philox4x32 (Salmon et al., SC'11) with PHILOX_ROUNDS rounds: 7 is the paper's fastest Crush-resistant variant (10 its
// conservative default, which rounds 1-4 of this repository used); a round's hi / lo word pairs come from ONE 32 x 32 -> 64-bit
// multiply each (v_mad_u64_u32) instead of a v_mul_hi_u32 + v_mul_lo_u32 pair -- both multiplies are quarter-rate instructions
// and the generator, not HBM, bounded every kernel that draws a mask (affine2: 62.8 us with dropout, 45.7 without; round 5).
#ifndef PHILOX_ROUNDS
#define PHILOX_ROUNDS 7
#endif
__device__ static inline void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t k0, uint32_t k1, uint32_t (&out)[4]) {
  uint32_t c[4] = {c0, c1, 0x9E3779B9u, 0xBB67AE85u};
#pragma unroll
  for (int r = 0; r < PHILOX_ROUNDS; ++r) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c[0], p1 = (unsigned long long)0xCD9E8D57u * c[2];
    const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    const uint32_t n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
    c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
}

// keep-mask scale factors (0 or 1/(1-p)) for VW consecutive elements starting at flat element index e0 (e0 % VW == 0,
// VW in {4, 8}).  One Philox call yields EIGHT 16-bit uniform draws: element e uses draw e & 7 of block e >> 3 (the
// probability is resolved to 2^-16; a 32-bit draw per element doubled the integer work of the bf16 kernels).
template <int VW>
__device__ static inline void drop_scales(float (&m)[VW], size_t e0, uint32_t thr, float inv_keep, uint32_t s0, uint32_t s1) {
#pragma unroll
  for (int b = 0; b < (VW + 7) / 8; ++b) {
    uint32_t rnd[4];
    const size_t blk = e0 / 8 + b;
    philox4x32_10((uint32_t)blk, (uint32_t)(blk >> 32), s0, s1, rnd);
    const int j0 = VW >= 8 ? 0 : (int)(e0 & 7);            // 0, or 4 for the upper half of a block (VW == 4)
#pragma unroll
    for (int j = 0; j < (VW < 8 ? VW : 8); ++j) {
      const int d = j0 + j;                                // draw index within the block
      uint32_t w = rnd[0];
      if ((d >> 1) == 1) w = rnd[1];
      if ((d >> 1) == 2) w = rnd[2];
      if ((d >> 1) == 3) w = rnd[3];
      const uint32_t v = (d & 1) ? (w >> 16) : (w & 0xFFFFu);
      if (8 * b + j < VW) m[8 * b + j] = v >= thr ? inv_keep : 0.f;
    }
  }
}
__device__ static inline float drop_scale1(size_t e, uint32_t thr, float inv_keep, uint32_t s0, uint32_t s1) {
  uint32_t rnd[4];
  const size_t blk = e / 8;
  philox4x32_10((uint32_t)blk, (uint32_t)(blk >> 32), s0, s1, rnd);
  const int d = (int)(e & 7);
  const uint32_t w = rnd[d >> 1];
  const uint32_t v = (d & 1) ? (w >> 16) : (w & 0xFFFFu);
  return v >= thr ? inv_keep : 0.f;
}

struct DropCfg {
  uint32_t thr; float inv_keep; uint32_t s0, s1; int on;
  // optional device-resident offset added to the seed when the kernel runs: lets a launch recorded ONCE in a hipGraph
  // draw a fresh mask on every replay (the host increments the counter -- or records its increment in the same graph)
  const unsigned long long* epoch;
};
__device__ static inline void drop_key(const DropCfg& D, uint32_t& k0, uint32_t& k1) {
  unsigned long long s = ((unsigned long long)D.s1 << 32) | D.s0;
  if (D.on && D.epoch) s += *D.epoch;
  k0 = (uint32_t)s; k1 = (uint32_t)(s >> 32);
}

static inline DropCfg make_drop(float p, unsigned long long seed, const unsigned long long* epoch) {
  DropCfg d{};
  d.on = p > 0.f;
  d.epoch = epoch;
  if (d.on) {
    double t = (double)p * 65536.0 + 0.5;                  // 16-bit draws: drop when draw < thr
    d.thr = t >= 65536.0 ? 65536u : (uint32_t)t;
    d.inv_keep = p < 1.f ? 1.f / (1.f - p) : 0.f;
    d.s0 = (uint32_t)seed; d.s1 = (uint32_t)(seed >> 32);
  }
  return d;
}

