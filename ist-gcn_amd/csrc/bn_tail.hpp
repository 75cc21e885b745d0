// "Last workgroup finalises": the per-channel arithmetic of istgcn_bn_finalize / istgcn_bn_bwd_coef (5 us kernels that sit
// strictly between two dependent passes, 46 launches per step) as the TAIL of the kernel that produces their batch sums.
// Protocol (the classic threadfence reduction): every workgroup adds its partial sums with device-scope atomics, fences,
// and takes a ticket; the workgroup that draws the last ticket -- all sums are then complete and, being atomics' results,
// live at the device's coherence point -- reads them with device-scope atomic loads, computes the coefficients, updates
// the running statistics, zeroes the sums for the next producer and resets the ticket.  One source for the arithmetic:
// the stand-alone kernels of pointwise.hip call the same two functions.
//
// Host side: a tail is ARMED per host thread (istgcn_bn_tail_arm_*), the next launch of a supporting producer whose
// `stats` pointer matches takes it (istgcn_bn_tail_take), and istgcn_bn_tail_disarm() tells the caller whether anybody
// did -- if not (a kernel variant without the tail), the caller launches the stand-alone kernel.
#pragma once
#include "common.hpp"

struct BnTail {
  int kind;                  // 0: none, 1: forward finalize, 2: backward coefficients
  int rep, C, training;
  double* stats;             // [rep][2][C], zeroed again by the tail
  unsigned* ticket;          // zero between launches
  double count;
  const float* gamma; const float* beta;      // kind 1 (beta), both kinds (gamma)
  float* rmean; float* rvar;                  // kind 1: running statistics (may be null)
  float momentum, eps;
  const float* coef_in;                       // kind 2: [4][C] of the forward
  float* out0;                                // kind 1: coef [4][C]; kind 2: abc [3][C]
  float* out1; float* out2;                   // kind 2: dgamma, dbeta (may be null)
};

__device__ static inline double bn_tail_load(const double* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// forward: sums (s, ss) of one channel -> coef rows scale, shift, mean, rstd (+ running statistics); the arithmetic of
// nn.BatchNorm2d in training mode (net/st_gcnold.py:165,174,192)
__device__ static inline void bn_finalize_channel(double s, double ss, double count, const float* gamma, const float* beta,
                                                  float* rmean, float* rvar, float momentum, float eps, float* coef, int C, int c) {
  const double m = s / count;
  double var = ss / count - m * m;
  if (var < 0) var = 0;
  const float mean = (float)m;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  if (rmean) {
    const double unb = count > 1 ? var * count / (count - 1) : var;
    rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)m;
    rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unb;
  }
  const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
  coef[c] = g * rstd;
  coef[C + c] = b - mean * g * rstd;
  coef[2 * C + c] = mean;
  coef[3 * C + c] = rstd;
}

// backward: sums (s = sum d, sx = sum d * xhat) of one channel -> (a, b, c) of dx = a*d + b*x + c, dgamma, dbeta
__device__ static inline void bn_bwd_coef_channel(double s, double sx, double count, const float* gamma, const float* coef,
                                                  int training, float* abc, float* dgamma, float* dbeta, int C, int c) {
  const float g = gamma ? gamma[c] : 1.f, mean = coef[2 * C + c], rstd = coef[3 * C + c];
  if (dgamma) dgamma[c] = (float)sx;
  if (dbeta) dbeta[c] = (float)s;
  if (training) {
    const double m1 = s / count, m2 = sx / count;
    abc[c] = g * rstd;
    abc[C + c] = (float)(-(double)g * rstd * rstd * m2);
    abc[2 * C + c] = (float)(-(double)g * rstd * m1 + (double)g * rstd * rstd * m2 * mean);
  } else {
    abc[c] = g * rstd; abc[C + c] = 0.f; abc[2 * C + c] = 0.f;
  }
}

// Call at the very end of a producer kernel, by ALL threads of every workgroup, after the workgroup's own atomics into
// t.stats have been issued.  nwg = workgroups of the launch; lds_word = any LDS word the kernel no longer needs (the
// kernels opt in to the full 160 KB of DYNAMIC LDS, which leaves no room for a static __shared__ variable here).
__device__ static inline void bn_tail_run(const BnTail& t, unsigned nwg, unsigned* lds_word) {
  if (t.kind == 0) return;                                   // (uniform over the launch)
  // The only data the workgroups exchange are results of device-scope ATOMICS (performed at the coherence point, never
  // in a per-XCD cache), read back with device-scope atomic loads: no cache write-back / invalidate is needed, only that
  // each thread's atomics have been performed before the ticket is drawn.  (__threadfence() here costs a write-back of
  // the L2's dirty lines -- the kernel's whole output -- per workgroup: measured +28 us per launch.)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this thread's atomics are acknowledged
  __syncthreads();                                           // ... and so are those of the whole workgroup (LDS is free now)
  if (threadIdx.x == 0) {
    const unsigned tk = __hip_atomic_fetch_add(t.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *lds_word = tk == nwg - 1 ? 1u : 0u;
  }
  __syncthreads();
  if (!*lds_word) return;
  const int C = t.C;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    double s = 0, q = 0;
    for (int r = 0; r < t.rep; ++r) {
      double* p0 = t.stats + (size_t)r * 2 * C + c;
      s += bn_tail_load(p0);
      q += bn_tail_load(p0 + C);
      p0[0] = 0.0;                                           // ready for the next producer (ordered before it by the kernel boundary)
      p0[C] = 0.0;
    }
    if (t.kind == 1) bn_finalize_channel(s, q, t.count, t.gamma, t.beta, t.rmean, t.rvar, t.momentum, t.eps, t.out0, C, c);
    else bn_bwd_coef_channel(s, q, t.count, t.gamma, t.coef_in, t.training, t.out0, t.out1, t.out2, C, c);
  }
  if (threadIdx.x == 0) __hip_atomic_store(t.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// host side (defined in pointwise.hip): the armed tail of this host thread, if its stats pointer is `stats`; disarms it
extern "C" int istgcn_bn_tail_take(const double* stats, BnTail* out);
