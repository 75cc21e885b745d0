// HBM-bound glue of the st_gcn block around the two MFMA kernels (all on [rows = NM*T*V][C] NTVC views):
//   bn_finalize     batch sums -> BatchNorm2d affine (train: batch stats + running-stat update; eval: running stats)
//   block_out_fwd   out = relu( dropout( z*s2 + b2 ) + residual )                 net/st_gcnold.py:174-175,201-203
//   block_out_bwd   d = dout * [out > 0]  (residual gradient), BatchNorm-backward sums of d*mask for tcn.3 and of d
//                   for the residual BatchNorm                                   (autograd of the same lines)
//   bn_bwd_coef     sums -> per-channel (a, b, c) with  dx = a*d*mask + b*x + c,  and dgamma, dbeta
//   affine2         out = a[c]*d*mask + b[c]*x + c[c]                             (BatchNorm backward, elementwise part)
// Dropout uses a counter-based Philox4x32-10 stream keyed by (seed, element index): the backward pass regenerates
// the forward mask instead of storing it.
#include "common.hpp"
#include "dropout.hpp"
#include "bn_tail.hpp"

namespace {

constexpr int NT = 256;

// Streaming (non-temporal) loads: these passes read every vector exactly once; left to the default policy they push the
// weights and halo rows of the MFMA kernels that run between them out of L2 / the Infinity Cache.  Measured over the whole
// step (config 2, same box, ms/step): default policy 14.77, every operand streaming 14.41 (-2.4 %), only the operands that
// come from far back (COLD: the forward's tensors in the backward pass, the block input in block_out_fwd) 14.52; config 5:
// -0.4 %.  Streaming STORES on top measured worse (the next kernel reads the output), and so did streaming loads in the
// bottleneck stream kernels (their inputs were just written and still sit in the Infinity Cache) and of tconv's staged
// chunks (halo rows are re-read by the neighbouring tile).  -DISTGCN_X_NT=0 none, 1 (default) every operand, 2 the cold ones.
#ifndef ISTGCN_X_NT
#define ISTGCN_X_NT 1
#endif
template <typename T, int VW, bool COLD = false>
__device__ static inline void load_vec(const T* p, float (&f)[VW]) {
  constexpr bool NTL = ISTGCN_X_NT == 1 || (ISTGCN_X_NT == 2 && COLD);
  if constexpr (VW == Elem<T>::EPL) {
    typename Elem<T>::frag r;
    if constexpr (NTL) r = __builtin_nontemporal_load(reinterpret_cast<const typename Elem<T>::frag*>(p));
    else r = *reinterpret_cast<const typename Elem<T>::frag*>(p);
#pragma unroll
    for (int j = 0; j < VW; ++j) f[j] = Elem<T>::to_f(r[j]);
  } else {
#pragma unroll
    for (int j = 0; j < VW; ++j) f[j] = Elem<T>::to_f(p[j]);
  }
}
template <typename T, int VW>
__device__ static inline void store_vec(T* p, float (&f)[VW]) {
  if constexpr (VW == Elem<T>::EPL) {
    typename Elem<T>::frag r;
#pragma unroll
    for (int j = 0; j < VW; ++j) r[j] = Elem<T>::from_f(f[j]);
    *reinterpret_cast<typename Elem<T>::frag*>(p) = r;
  } else {
#pragma unroll
    for (int j = 0; j < VW; ++j) p[j] = Elem<T>::from_f(f[j]);
  }
}

// bit j = (f[j], rounded to the storage type, is > 0): the ReLU mask of a stored activation vector in one byte
template <typename T, int VW>
__device__ static inline unsigned relu_bits(const float (&f)[VW]) {
  unsigned b = 0;
#pragma unroll
  for (int j = 0; j < VW; ++j) b |= (Elem<T>::to_f(Elem<T>::from_f(f[j])) > 0.f ? 1u : 0u) << j;
  return b;
}

// ---------------------------------------------------------------------------------------------------------
__global__ void bn_finalize_kernel(double* stats, int rep, int clear, double count, const float* gamma,
                                   const float* beta, float* rmean, float* rvar, float momentum, float eps, int training,
                                   float* coef, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  if (training) {
    double s = 0, ss = 0;
    for (int r = 0; r < rep; ++r) {
      s += stats[(size_t)r * 2 * C + c]; ss += stats[(size_t)r * 2 * C + C + c];
      if (clear) { stats[(size_t)r * 2 * C + c] = 0.0; stats[(size_t)r * 2 * C + C + c] = 0.0; }   // ready for the next producer
    }
    bn_finalize_channel(s, ss, count, gamma, beta, rmean, rvar, momentum, eps, coef, C, c);
  } else {
    const float mean = rmean[c], rstd = 1.f / sqrtf(rvar[c] + eps);
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    coef[c] = g * rstd;
    coef[C + c] = b - mean * g * rstd;
    coef[2 * C + c] = mean;
    coef[3 * C + c] = rstd;
  }
}

__global__ void bn_bwd_coef_kernel(double* stats, int rep, int clear, double count, const float* gamma,
                                   const float* coef, int training, float* abc, float* dgamma, float* dbeta, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double s = 0, sx = 0;
  for (int r = 0; r < rep; ++r) {
    s += stats[(size_t)r * 2 * C + c]; sx += stats[(size_t)r * 2 * C + C + c];
    if (clear) { stats[(size_t)r * 2 * C + c] = 0.0; stats[(size_t)r * 2 * C + C + c] = 0.0; }
  }
  bn_bwd_coef_channel(s, sx, count, gamma, coef, training, abc, dgamma, dbeta, C, c);
}

// ---------------------------------------------------------------------------------------------------------
// In the element-wise kernels the grid stride (gridDim.x * 256 threads) is a multiple of the vectors per row whenever
// QC divides 256, so a thread always lands on the same channel vector: its per-channel coefficients are loaded once
// (`fixed_q`); otherwise they are re-read per element (the generic path for odd channel counts).
template <typename T, int VW>
__global__ __launch_bounds__(NT) void block_out_fwd_kernel(const T* z, const float* coef2, const T* res, const float* coefr,
                                                          T* out, unsigned char* rmask, size_t rows, int C, DropCfg D) {
  uint32_t dk0, dk1;
  drop_key(D, dk0, dk1);
  const int QC = C / VW;
  const size_t total = rows * QC;
  const bool fixed_q = (NT % QC) == 0;
  float s2[VW], b2[VW], sr[VW], br[VW];
  if (fixed_q) {
    const int c0 = (threadIdx.x % QC) * VW;
#pragma unroll
    for (int j = 0; j < VW; ++j) {
      s2[j] = coef2[c0 + j]; b2[j] = coef2[C + c0 + j];
      sr[j] = coefr ? coefr[c0 + j] : 1.f; br[j] = coefr ? coefr[C + c0 + j] : 0.f;
    }
  }
  for (size_t idx = (size_t)blockIdx.x * NT + threadIdx.x; idx < total; idx += (size_t)gridDim.x * NT) {
    const size_t e0 = idx * VW;
    if (!fixed_q) {
      const int c0 = (int)(idx % QC) * VW;
#pragma unroll
      for (int j = 0; j < VW; ++j) {
        s2[j] = coef2[c0 + j]; b2[j] = coef2[C + c0 + j];
        sr[j] = coefr ? coefr[c0 + j] : 1.f; br[j] = coefr ? coefr[C + c0 + j] : 0.f;
      }
    }
    float zv[VW], rv[VW], m[VW];
    load_vec<T, VW>(z + e0, zv);
    if (res) load_vec<T, VW, true>(res + e0, rv);
    if (D.on) {
      if constexpr (VW % 4 == 0) drop_scales<VW>(m, e0, D.thr, D.inv_keep, dk0, dk1);
      else {
#pragma unroll
        for (int j = 0; j < VW; ++j) m[j] = drop_scale1(e0 + j, D.thr, D.inv_keep, dk0, dk1);
      }
    }
#pragma unroll
    for (int j = 0; j < VW; ++j) {
      float y = zv[j] * s2[j] + b2[j];
      if (D.on) y *= m[j];
      if (res) y += rv[j] * sr[j] + br[j];
      zv[j] = fmaxf(y, 0.f);
    }
    store_vec<T, VW>(out + e0, zv);
    if (rmask) rmask[idx] = (unsigned char)relu_bits<T, VW>(zv);      // one byte per vector: the backward's [out > 0]
  }
}

// d = dout*[out>0] -> dres (or nowhere: dres == NULL) ; stats2 += (sum d*mask, sum d*mask*zhat) ; statsr += (sum d, sum d*rhat)
// HASR: the residual branch has its own BatchNorm (strided 1x1 conv, 2 of 10 blocks); a compile-time switch because the
// second set of per-channel constants and sums costs 32 registers = one resident wave per SIMD (126 -> 94 VGPRs)
template <typename T, int VW, bool HASR>
__global__ __launch_bounds__(NT) void block_out_bwd_kernel(const T* dout, const T* out, const unsigned char* rmask, const T* z,
                                                          const float* coef2, const T* r, const float* coefr, T* dres,
                                                          double* stats2, double* statsr, int rep, size_t rows, int C,
                                                          DropCfg D, BnTail tail) {
  uint32_t dk0, dk1;
  drop_key(D, dk0, dk1);
  __shared__ float red[4][NT];
  const int QC = C / VW;                 // vectors per row; NT % QC == 0 is guaranteed by the launcher
  const int q = threadIdx.x % QC;
  const int c0 = q * VW;
  const size_t rstep = (size_t)gridDim.x * (NT / QC);
  float a1[VW], a2[VW], b1[VW], b2[VW], mz[VW], rz[VW], mr[VW], rr_[VW];
#pragma unroll
  for (int j = 0; j < VW; ++j) {
    a1[j] = a2[j] = b1[j] = b2[j] = 0.f;
    mz[j] = coef2[2 * C + c0 + j]; rz[j] = coef2[3 * C + c0 + j];
    mr[j] = HASR ? coefr[2 * C + c0 + j] : 0.f; rr_[j] = HASR ? coefr[3 * C + c0 + j] : 0.f;
  }
  for (size_t row = (size_t)blockIdx.x * (NT / QC) + threadIdx.x / QC; row < rows; row += rstep) {
    const size_t e0 = row * C + c0;
    float dv[VW], ov[VW], zv[VW], rv[VW], m[VW];
    load_vec<T, VW>(dout + e0, dv);
    if (rmask) {
      // the forward's one-byte ReLU mask of this vector instead of the 16-byte read of `out`
      const unsigned bits = rmask[row * QC + q];
#pragma unroll
      for (int j = 0; j < VW; ++j) ov[j] = (bits >> j) & 1u ? 1.f : 0.f;
    } else {
      load_vec<T, VW>(out + e0, ov);
    }
    load_vec<T, VW, true>(z + e0, zv);
    if (HASR) load_vec<T, VW, true>(r + e0, rv);
    if (D.on) {
      if constexpr (VW % 4 == 0) drop_scales<VW>(m, e0, D.thr, D.inv_keep, dk0, dk1);
      else {
#pragma unroll
        for (int j = 0; j < VW; ++j) m[j] = drop_scale1(e0 + j, D.thr, D.inv_keep, dk0, dk1);
      }
    }
#pragma unroll
    for (int j = 0; j < VW; ++j) {
      const float d = ov[j] > 0.f ? dv[j] : 0.f;
      dv[j] = d;
      const float dm = D.on ? d * m[j] : d;
      a1[j] += dm;
      a2[j] += dm * (zv[j] - mz[j]) * rz[j];
      if (HASR) {
        b1[j] += d;
        b2[j] += d * (rv[j] - mr[j]) * rr_[j];
      }
    }
    if (dres) store_vec<T, VW>(dres + e0, dv);           // (round 5: NULL when every consumer takes dout + the ReLU mask itself)
  }
  // block reduction over the NT/QC threads that share a channel vector
#pragma unroll
  for (int j = 0; j < VW; ++j) {
    red[0][threadIdx.x] = a1[j]; red[1][threadIdx.x] = a2[j]; red[2][threadIdx.x] = b1[j]; red[3][threadIdx.x] = b2[j];
    __syncthreads();
    if (threadIdx.x < QC) {
      float s0 = 0, s1 = 0, s2 = 0, s3 = 0;
      for (int t = threadIdx.x; t < NT; t += QC) { s0 += red[0][t]; s1 += red[1][t]; s2 += red[2][t]; s3 += red[3][t]; }
      const int c = c0 + j;
      double* d2 = stats2 + (size_t)(blockIdx.x % rep) * 2 * C;
      atomic_add_f64(d2 + c, (double)s0);
      atomic_add_f64(d2 + C + c, (double)s1);
      if (HASR) {
        double* dr = statsr + (size_t)(blockIdx.x % rep) * 2 * C;
        atomic_add_f64(dr + c, (double)s2);
        atomic_add_f64(dr + C + c, (double)s3);
      }
    }
    __syncthreads();
  }
  bn_tail_run(tail, gridDim.x, reinterpret_cast<unsigned*>(&red[0][0]));                              // tcn.3's backward coefficients, when the caller armed them
}

// rmask (or NULL): the forward's ReLU byte mask of the tensor d is the gradient of -- d := d * [bit] first, i.e. the kernel reads
// dout itself where block_out_bwd used to write dres = dout * [out > 0] for it (one byte per VW-element vector, VW = 16 bytes)
template <typename T, int VW>
__global__ __launch_bounds__(NT) void affine2_kernel(const T* d, const unsigned char* rmask, const T* x, const float* abc, T* out,
                                                    size_t rows, int C, DropCfg D) {
  uint32_t dk0, dk1;
  drop_key(D, dk0, dk1);
  const int QC = C / VW;
  const size_t total = rows * QC;
  const bool fixed_q = (NT % QC) == 0;
  float ca[VW], cb[VW], cc[VW];
  if (fixed_q) {
    const int c0 = (threadIdx.x % QC) * VW;
#pragma unroll
    for (int j = 0; j < VW; ++j) { ca[j] = abc[c0 + j]; cb[j] = abc[C + c0 + j]; cc[j] = abc[2 * C + c0 + j]; }
  }
  for (size_t idx = (size_t)blockIdx.x * NT + threadIdx.x; idx < total; idx += (size_t)gridDim.x * NT) {
    const size_t e0 = idx * VW;
    if (!fixed_q) {
      const int c0 = (int)(idx % QC) * VW;
#pragma unroll
      for (int j = 0; j < VW; ++j) { ca[j] = abc[c0 + j]; cb[j] = abc[C + c0 + j]; cc[j] = abc[2 * C + c0 + j]; }
    }
    float dv[VW], xv[VW], m[VW];
    load_vec<T, VW>(d + e0, dv);
    if (x) load_vec<T, VW, true>(x + e0, xv);
    if (rmask) {
      const unsigned bits = rmask[idx];
#pragma unroll
      for (int j = 0; j < VW; ++j) dv[j] = (bits >> j) & 1u ? dv[j] : 0.f;
    }
    if (D.on) {
      if constexpr (VW % 4 == 0) drop_scales<VW>(m, e0, D.thr, D.inv_keep, dk0, dk1);
      else {
#pragma unroll
        for (int j = 0; j < VW; ++j) m[j] = drop_scale1(e0 + j, D.thr, D.inv_keep, dk0, dk1);
      }
    }
#pragma unroll
    for (int j = 0; j < VW; ++j) {
      float v = dv[j];
      if (D.on) v *= m[j];
      v = v * ca[j];
      if (x) v += xv[j] * cb[j] + cc[j];
      dv[j] = v;
    }
    store_vec<T, VW>(out + e0, dv);
  }
}


// Global average pooling of the trunk output (net/st_gcnold.py:89-91: F.avg_pool2d over (T, V), then the mean over the M
// persons of a clip): partial SUMS over row slices, out[s][rows of slice][c] -> psum [S][seqs][C]; the caller adds the M*S
// partial rows of a clip.  One workgroup = one (sequence, slice): lanes walk the channel vectors, row groups the rows.
template <typename T, int VW>
__global__ __launch_bounds__(NT) void pool_fwd_kernel(const T* y, float* psum, int P, int C, int S) {
  __shared__ float red[NT * VW];
  const int QC = C / VW;                       // channel vectors per row (<= NT, checked on the host)
  const int nm = blockIdx.x, sl = blockIdx.y;
  const int rg = threadIdx.x / QC, q = threadIdx.x - rg * QC, RG = NT / QC;
  const int r_lo = (int)((long long)P * sl / S), r_hi = (int)((long long)P * (sl + 1) / S);
  float acc[VW];
#pragma unroll
  for (int j = 0; j < VW; ++j) acc[j] = 0.f;
  if (rg < RG) {
    const T* base = y + ((size_t)nm * P) * C + q * VW;
    for (int r = r_lo + rg; r < r_hi; r += RG) {
      float v[VW];
      load_vec<T, VW>(base + (size_t)r * C, v);
#pragma unroll
      for (int j = 0; j < VW; ++j) acc[j] += v[j];
    }
  }
#pragma unroll
  for (int j = 0; j < VW; ++j) red[threadIdx.x * VW + j] = acc[j];
  __syncthreads();
  if (threadIdx.x < QC) {
    float tot[VW];
#pragma unroll
    for (int j = 0; j < VW; ++j) tot[j] = 0.f;
    for (int g = 0; g < RG; ++g)
#pragma unroll
      for (int j = 0; j < VW; ++j) tot[j] += red[(g * QC + threadIdx.x) * VW + j];
    float* o = psum + ((size_t)nm * S + sl) * C + threadIdx.x * VW;
#pragma unroll
    for (int j = 0; j < VW; ++j) o[j] = tot[j];
  }
}

// backward of the pooling: dy[nm][p][c] = scale * dfeat[nm / M][c] for every position p (broadcast store)
template <typename T, int VW>
__global__ __launch_bounds__(NT) void pool_bwd_kernel(const float* dfeat, T* dy, size_t rows, int P, int C, int M, float scale) {
  const int QC = C / VW;
  const size_t total = rows * QC;
  for (size_t idx = (size_t)blockIdx.x * NT + threadIdx.x; idx < total; idx += (size_t)gridDim.x * NT) {
    const size_t row = idx / QC;
    const int q = (int)(idx - row * QC);
    const size_t n = row / ((size_t)P * M);
    float v[VW];
#pragma unroll
    for (int j = 0; j < VW; ++j) v[j] = scale * dfeat[n * C + q * VW + j];
    store_vec<T, VW>(dy + idx * VW, v);
  }
}

// Persistent-grid caps (workgroups of 256 threads) of the streaming passes, measured over the whole step (config 2, ms/step
// of the family; the cap decides how many concurrent sequential streams HBM sees): affine2 512: 1.66, 768: 1.51, 1024: 1.50,
// 1536: 1.52, 2048: 1.61, 4096: 1.49, 8192: 1.46 (but block_out_fwd 0.94 there); block_out_fwd 512: 0.83, 768 / 1024: 0.73,
// 2048: 0.78; block_out_bwd 512: 1.07, 768: 0.92, 1024: 0.98, 1536: 1.05, 3072: 1.57.  -DISTGCN_X_*CAP override (experiments).
#ifndef ISTGCN_X_EWCAP
#define ISTGCN_X_EWCAP 2048
#endif
#ifndef ISTGCN_X_BOFCAP
#define ISTGCN_X_BOFCAP 1024
#endif
#ifndef ISTGCN_X_AFFCAP
#define ISTGCN_X_AFFCAP 1024
#endif
static inline int ew_grid(size_t items, size_t cap = ISTGCN_X_EWCAP) {
  size_t g = (items + NT - 1) / NT;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

// vector width: the 16-byte fragment when the channel count allows it and the per-channel thread map divides the block
template <typename T> static inline int pick_vw(int C, bool need_div) {
  const int epl = Elem<T>::EPL;
  if (C % epl == 0 && (!need_div || (NT % (C / epl) == 0 && C / epl <= NT))) return epl;
  return 1;
}

}  // namespace

// ---- armed BatchNorm tail of this host thread (bn_tail.hpp) ----
static thread_local BnTail g_bn_tail{};

extern "C" int istgcn_bn_tail_take(const double* stats, BnTail* out) {
  if (g_bn_tail.kind == 0 || g_bn_tail.stats != stats) return 0;
  *out = g_bn_tail;
  g_bn_tail.kind = 0;
  return 1;
}

extern "C" int istgcn_bn_tail_arm_finalize(double* stats, int stats_rep, double count, const float* gamma, const float* beta,
                                           float* running_mean, float* running_var, float momentum, float eps, float* coef,
                                           int C, unsigned* ticket) {
  if (!stats || !coef || !ticket || stats_rep < 1 || C < 1 || count <= 0) return ISTGCN_EINVAL;
  if ((running_mean == nullptr) != (running_var == nullptr)) return ISTGCN_EINVAL;
  BnTail t{};
  t.kind = 1; t.rep = stats_rep; t.C = C; t.training = 1; t.stats = stats; t.ticket = ticket; t.count = count;
  t.gamma = gamma; t.beta = beta; t.rmean = running_mean; t.rvar = running_var; t.momentum = momentum; t.eps = eps; t.out0 = coef;
  g_bn_tail = t;
  return ISTGCN_OK;
}

extern "C" int istgcn_bn_tail_arm_bwd(double* stats, int stats_rep, double count, const float* gamma, const float* coef,
                                      int training, float* abc, float* dgamma, float* dbeta, int C, unsigned* ticket) {
  if (!stats || !coef || !abc || !ticket || stats_rep < 1 || C < 1 || count <= 0) return ISTGCN_EINVAL;
  BnTail t{};
  t.kind = 2; t.rep = stats_rep; t.C = C; t.training = training; t.stats = stats; t.ticket = ticket; t.count = count;
  t.gamma = gamma; t.coef_in = coef; t.out0 = abc; t.out1 = dgamma; t.out2 = dbeta;
  g_bn_tail = t;
  return ISTGCN_OK;
}

// 1 if a tail was still armed (no producer took it: the caller runs the stand-alone kernel), 0 otherwise
extern "C" int istgcn_bn_tail_disarm() {
  const int was = g_bn_tail.kind != 0;
  g_bn_tail.kind = 0;
  return was;
}

extern "C" int istgcn_bn_finalize(double* stats, int stats_rep, int clear, double count, const float* gamma,
                                  const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                                  int training, float* coef, int C, void* stream) {
  if (!coef || C < 1) return ISTGCN_EINVAL;
  if (training && (!stats || stats_rep < 1 || count <= 0)) return ISTGCN_EINVAL;
  if (!training && (!running_mean || !running_var)) return ISTGCN_EINVAL;
  ISTGCN_LAUNCH(bn_finalize_kernel, dim3((C + 127) / 128), dim3(128), 0, (hipStream_t)stream, stats, stats_rep, clear,
                     count, gamma, beta, running_mean, running_var, momentum, eps, training, coef, C);
  ISTGCN_CHECK_LAUNCH();
  return ISTGCN_OK;
}

extern "C" int istgcn_bn_bwd_coef(double* stats, int stats_rep, int clear, double count, const float* gamma, const float* coef,
                                  int training, float* abc, float* dgamma, float* dbeta, int C, void* stream) {
  if (!stats || !coef || !abc || C < 1 || stats_rep < 1 || count <= 0) return ISTGCN_EINVAL;
  ISTGCN_LAUNCH(bn_bwd_coef_kernel, dim3((C + 127) / 128), dim3(128), 0, (hipStream_t)stream, stats, stats_rep, clear,
                     count, gamma, coef, training, abc, dgamma, dbeta, C);
  ISTGCN_CHECK_LAUNCH();
  return ISTGCN_OK;
}

// 1 when the one-byte-per-vector ReLU mask is available for (C, dtype): forward and backward both use whole 16-byte
// vectors (C a multiple of the vector width and the per-channel thread map of the backward divides the block)
extern "C" int istgcn_relu_mask_ok(int C, int dtype) {
  if (!istgcn_dtype_ok(dtype) || C < 1) return 0;
  const int epl = dtype == 0 ? 4 : 8;
  return (C % epl == 0 && C / epl <= NT && NT % (C / epl) == 0) ? 1 : 0;
}

// dtype dispatch of the element-wise kernels: cast every activation pointer to the element type and launch
#define EW_CASES(BODY)                                           \
  do {                                                           \
    if (dtype == 0) { typedef float ET; constexpr int VWB = 4; BODY; }        \
    else if (dtype == 1) { typedef __bf16 ET; constexpr int VWB = 8; BODY; }  \
    else { typedef _Float16 ET; constexpr int VWB = 8; BODY; }                \
  } while (0)
#define DISPATCH_VW(KERNEL, TYPE, VWBIG, grid, ...)                                                                   \
  do {                                                                                                            \
    if (vw == VWBIG) ISTGCN_LAUNCH((KERNEL<TYPE, VWBIG>), grid, dim3(NT), 0, (hipStream_t)stream, __VA_ARGS__); \
    else ISTGCN_LAUNCH((KERNEL<TYPE, 1>), grid, dim3(NT), 0, (hipStream_t)stream, __VA_ARGS__);              \
  } while (0)

extern "C" int istgcn_block_out_fwd(const void* z, const float* coef2, const void* res, const float* coefr, void* out,
                                    unsigned char* relu_mask, long long rows, int C, float p_drop, unsigned long long seed,
                                    const unsigned long long* seed_epoch, int dtype, void* stream) {
  if (!z || !coef2 || !out || rows < 0 || C < 1 || !istgcn_dtype_ok(dtype) || p_drop < 0.f || p_drop > 1.f)
    return ISTGCN_EINVAL;
  if (rows == 0) return ISTGCN_OK;
  const int vw = dtype == 0 ? pick_vw<float>(C, false) : pick_vw<__bf16>(C, false);
  // the mask is one byte per WHOLE vector: both directions must use the vector map (istgcn_relu_mask_ok)
  if (relu_mask && !istgcn_relu_mask_ok(C, dtype)) return ISTGCN_EINVAL;
  const DropCfg D = make_drop(p_drop, seed, seed_epoch);
  const dim3 grid(ew_grid((size_t)rows * (C / vw), ISTGCN_X_BOFCAP));
  EW_CASES(DISPATCH_VW(block_out_fwd_kernel, ET, VWB, grid, (const ET*)z, coef2, (const ET*)res, coefr, (ET*)out,
                       relu_mask, (size_t)rows, C, D));
  ISTGCN_CHECK_LAUNCH();
  return ISTGCN_OK;
}

extern "C" int istgcn_block_out_bwd(const void* dout, const void* out, const unsigned char* relu_mask, const void* z,
                                    const float* coef2, const void* r, const float* coefr, void* dres, double* stats2,
                                    double* statsr, int stats_rep, long long rows, int C, float p_drop,
                                    unsigned long long seed, const unsigned long long* seed_epoch, int dtype, void* stream) {
  if (!dout || (!out && !relu_mask) || !z || !coef2 || !stats2 || stats_rep < 1 || rows < 0 || C < 1) return ISTGCN_EINVAL;
  if (!dres && !relu_mask) return ISTGCN_EINVAL;       // without dres the consumers re-derive it from dout and the byte mask
  if (relu_mask && !istgcn_relu_mask_ok(C, dtype)) return ISTGCN_EINVAL;
  if ((r != nullptr) != (coefr != nullptr) || (r && !statsr)) return ISTGCN_EINVAL;
  if (!istgcn_dtype_ok(dtype) || p_drop < 0.f || p_drop > 1.f) return ISTGCN_EINVAL;
  if (rows == 0) return ISTGCN_OK;
  int vw = dtype == 0 ? pick_vw<float>(C, true) : pick_vw<__bf16>(C, true);
  if (vw == 1 && (C > NT || NT % C != 0)) return ISTGCN_EINVAL;   // scalar map needs C | 256
  const DropCfg D = make_drop(p_drop, seed, seed_epoch);
  const int rpb = NT / (C / vw);
  size_t g = ((size_t)rows + rpb - 1) / rpb;
  // persistent grid = the resident workgroups: 4 per CU with the residual BatchNorm's constants in registers (120 VGPRs),
  // 6 per CU without them (75)
#ifdef ISTGCN_X_BOBCAP
  const size_t gcap = ISTGCN_X_BOBCAP;
#else
  const size_t gcap = 768;               // (measured, above; was 1536 / 1024 with the residual BatchNorm)
#endif
  if (g > gcap) g = gcap;
  const dim3 grid((int)g);
  BnTail tail{};
  istgcn_bn_tail_take(stats2, &tail);
#define BOB_LAUNCH(VWv, HR)                                                                                              \
  ISTGCN_LAUNCH((block_out_bwd_kernel<ET, VWv, HR>), grid, dim3(NT), 0, (hipStream_t)stream, (const ET*)dout, (const ET*)out,  \
                relu_mask, (const ET*)z, coef2, (const ET*)r, coefr, (ET*)dres, stats2, statsr, stats_rep, (size_t)rows, C, D, tail)
  EW_CASES({
    if (vw == VWB) { if (r) BOB_LAUNCH(VWB, true); else BOB_LAUNCH(VWB, false); }
    else { if (r) BOB_LAUNCH(1, true); else BOB_LAUNCH(1, false); }
  });
#undef BOB_LAUNCH
  ISTGCN_CHECK_LAUNCH();
  return ISTGCN_OK;
}

extern "C" int istgcn_affine2m(const void* d, const unsigned char* relu_mask, const void* x, const float* abc, void* out,
                               long long rows, int C, float p_drop, unsigned long long seed,
                               const unsigned long long* seed_epoch, int dtype, void* stream) {
  if (!d || !abc || !out || rows < 0 || C < 1 || !istgcn_dtype_ok(dtype) || p_drop < 0.f || p_drop > 1.f)
    return ISTGCN_EINVAL;
  if (relu_mask && !istgcn_relu_mask_ok(C, dtype)) return ISTGCN_EINVAL;      // (the mask is one byte per 16-byte vector)
  if (rows == 0) return ISTGCN_OK;
  const int vw = dtype == 0 ? pick_vw<float>(C, false) : pick_vw<__bf16>(C, false);
  const DropCfg D = make_drop(p_drop, seed, seed_epoch);
  const dim3 grid(ew_grid((size_t)rows * (C / vw), ISTGCN_X_AFFCAP));
  EW_CASES(DISPATCH_VW(affine2_kernel, ET, VWB, grid, (const ET*)d, relu_mask, (const ET*)x, abc, (ET*)out, (size_t)rows, C, D));
  ISTGCN_CHECK_LAUNCH();
  return ISTGCN_OK;
}

extern "C" int istgcn_affine2(const void* d, const void* x, const float* abc, void* out, long long rows, int C,
                              float p_drop, unsigned long long seed, const unsigned long long* seed_epoch, int dtype,
                              void* stream) {
  return istgcn_affine2m(d, nullptr, x, abc, out, rows, C, p_drop, seed, seed_epoch, dtype, stream);
}

extern "C" int istgcn_pool_fwd(const void* y, float* psum, int NM, int P, int C, int S, int dtype, void* stream) {
  if (!y || !psum || NM < 0 || P < 1 || C < 1 || S < 1 || S > P || !istgcn_dtype_ok(dtype)) return ISTGCN_EINVAL;
  if (NM == 0) return ISTGCN_OK;
  int vw = dtype == 0 ? pick_vw<float>(C, false) : pick_vw<__bf16>(C, false);
  if (C / vw > NT) return ISTGCN_EINVAL;
  const dim3 grid(NM, S);
  EW_CASES(DISPATCH_VW(pool_fwd_kernel, ET, VWB, grid, (const ET*)y, psum, P, C, S));
  ISTGCN_CHECK_LAUNCH();
  return ISTGCN_OK;
}

extern "C" int istgcn_pool_bwd(const float* dfeat, void* dy, int NM, int P, int C, int M, float scale, int dtype, void* stream) {
  if (!dfeat || !dy || NM < 0 || P < 1 || C < 1 || M < 1 || (NM % M) != 0 || !istgcn_dtype_ok(dtype)) return ISTGCN_EINVAL;
  if (NM == 0) return ISTGCN_OK;
  const int vw = dtype == 0 ? pick_vw<float>(C, false) : pick_vw<__bf16>(C, false);
  const size_t rows = (size_t)NM * P;
  const dim3 grid(ew_grid(rows * (C / vw)));
  EW_CASES(DISPATCH_VW(pool_bwd_kernel, ET, VWB, grid, dfeat, (ET*)dy, rows, P, C, M, scale));
  ISTGCN_CHECK_LAUNCH();
  return ISTGCN_OK;
}
