// Weight packers: parameter tensors (fp32, any strides -- read in place from the nn.Conv2d weight) -> the fragment
// orders the MFMA loops stream (zero padded, cast to the activation dtype), one launch per weight.  These replace the
// pad / reshape / permute / cast chains of the host mirror (ops.pack_*_weight, kept as the executable specification
// the tests compare against bit for bit): a training step repacks every weight after each optimiser update, and a
// chain of 4-6 tiny launches per weight was a tenth of the bf16 step.
//
//   istgcn_pack_gcn      Wr[c][k][i]  -> [nch][MTtot][NKG][2][32][EPL]            (gcn_fwd.hip, see istgcn.h)
//   istgcn_pack_tconv    Wf[j][o][i]  -> [nch][ntaps][NKG][MTtot][2][32][EPL]     (tconv.hip); the taps are picked
//                        through `tap_sel` and o / i have independent strides, so the data-gradient weights
//                        (transposed, per-phase tap subset) come from the same parameter without a copy
//   istgcn_pack_gcn_bwd  W3[k][c][i]  -> [nchi][nchc][NKGc][MTK][2][32][EPL]      (gcn_bwd.hip)
#include "common.hpp"

extern "C" int istgcn_gcn_geometry(int Cin, int Cout, int K, int dtype, int* CCeff, int* nch, int* KKp, int* MTtot,
                                   int* EPL);
extern "C" int istgcn_tconv_geometry(int V, int Cin, int Cout, int ntaps, const int* tap_off, int in_mul, int dtype,
                                     int* CC, int* nch, int* MTtot, int* EPL);
extern "C" int istgcn_gcn_bwd_geometry(int Cin, int Cout, int K, int dtype, int* CCi, int* nchi, int* CCc, int* nchc,
                                       int* KKp, int* EPL);
extern "C" int istgcn_gcn_rc_layout(int Cin, int Cout, int K, int dtype);
extern "C" int istgcn_gcn_bwd_rc_layout(int Cin, int Cout, int K, int dtype);

namespace {

constexpr int MAX_TAPS = 16;

struct PackGcn {
  const float* src; void* dst;
  long long s_o, s_k, s_i;     // element strides of Wr[c][k][i]
  int Cin, Cout, K, CCeff, nch, NKG, MTtot;
  int rc;                      // the register-chained layout of gcn_rc.hip follows the round-2 one
};

// one thread per 16-byte output vector (h, r fixed, e = 0..EPL-1)
template <typename T>
__device__ static inline void pack_gcn_body(const PackGcn& P, int idx) {
  constexpr int EPL = Elem<T>::EPL;
  const int total = P.nch * P.MTtot * P.NKG * 2 * 32;
  if (idx >= total) {
    // register-chained section [jt][k][s][lane][8]: lane (c = lane & 31, h = lane >> 5) of fragment (jt, k, s) holds
    // Wr[32 jt + c][k][16 s + 8 h + e] -- the B operand of H_k = x W_k^T with the channels in natural k order
    if (!P.rc) return;
    if constexpr (sizeof(T) == 4) {
      // float32 section (gcn_rc_f32.hip) [jt][k][q][s4][lane][4]: lane (c, h) of vector (jt, k, q, s4) holds
      // Wr[32 jt + c][k][64 q + 32 h + 4 s4 + e]
      const int r = idx - total, NQ = P.Cin / 64;
      if (r >= P.K * (P.Cout / 32) * NQ * 8 * 64) return;
      const int lane = r & 63;
      int f = r >> 6;
      const int s4 = f & 7; f >>= 3;
      const int q = f % NQ; f /= NQ;
      const int k = f % P.K; const int jt = f / P.K;
      const int o = 32 * jt + (lane & 31), i0 = 64 * q + 32 * (lane >> 5) + 4 * s4;
      typename Elem<T>::frag v;
#pragma unroll
      for (int e = 0; e < EPL; ++e) v[e] = Elem<T>::from_f(P.src[o * P.s_o + k * P.s_k + (i0 + e) * P.s_i]);
      *reinterpret_cast<typename Elem<T>::frag*>(reinterpret_cast<T*>(P.dst) + (size_t)idx * EPL) = v;
      return;
    }
    const int r = idx - total, S = (P.Cin + 15) / 16;       // (the 3-channel first layer: one zero-padded k-step)
    if (r >= P.K * (P.Cout / 32) * S * 64) return;
    const int lane = r & 63;
    int f = r >> 6;
    const int s = f % S; f /= S;
    const int k = f % P.K; const int jt = f / P.K;
    const int o = 32 * jt + (lane & 31), i0 = 16 * s + 8 * (lane >> 5);
    typename Elem<T>::frag v;
#pragma unroll
    for (int e = 0; e < EPL; ++e) v[e] = Elem<T>::from_f((e < 8 && i0 + e < P.Cin) ? P.src[o * P.s_o + k * P.s_k + (i0 + e) * P.s_i] : 0.f);
    *reinterpret_cast<typename Elem<T>::frag*>(reinterpret_cast<T*>(P.dst) + (size_t)idx * EPL) = v;
    return;
  }
  int t = idx;
  const int r = t % 32; t /= 32;
  const int h = t % 2; t /= 2;
  const int kg = t % P.NKG; t /= P.NKG;
  const int mt = t % P.MTtot; const int ch = t / P.MTtot;
  const int o = mt * 32 + r;
  typename Elem<T>::frag v;
#pragma unroll
  for (int e = 0; e < EPL; ++e) {
    const int kk = kg * 2 * EPL + h * EPL + e;
    const int k = kk / P.CCeff, il = kk - k * P.CCeff, i = ch * P.CCeff + il;
    const bool ok = o < P.Cout && k < P.K && i < P.Cin;
    v[e] = Elem<T>::from_f(ok ? P.src[o * P.s_o + k * P.s_k + i * P.s_i] : 0.f);
  }
  *reinterpret_cast<typename Elem<T>::frag*>(reinterpret_cast<T*>(P.dst) + (size_t)idx * EPL) = v;
}

struct PackTconv {
  const float* src; void* dst;
  long long s_t, s_o, s_i;     // element strides of Wf[tap][o][i]
  int Cin, Cout, ntaps, CC, nch, NKG, MTtot;
  int tap_sel[MAX_TAPS];
};

template <typename T>
__global__ __launch_bounds__(256) void pack_gcn_kernel(const PackGcn P) { pack_gcn_body<T>(P, blockIdx.x * 256 + threadIdx.x); }

template <typename T>
__device__ static inline void pack_tconv_body(const PackTconv& P, int idx) {
  constexpr int EPL = Elem<T>::EPL;
  const int total = P.nch * P.ntaps * P.NKG * P.MTtot * 2 * 32;
  if (idx >= total) return;
  int t = idx;
  const int r = t % 32; t /= 32;
  const int h = t % 2; t /= 2;
  const int mt = t % P.MTtot; t /= P.MTtot;
  const int kg = t % P.NKG; t /= P.NKG;
  const int j = t % P.ntaps; const int ch = t / P.ntaps;
  const int o = mt * 32 + r;
  const float* base = P.src + P.tap_sel[j] * P.s_t + o * P.s_o;
  typename Elem<T>::frag v;
#pragma unroll
  for (int e = 0; e < EPL; ++e) {
    const int i = ch * P.CC + kg * 2 * EPL + h * EPL + e;
    v[e] = Elem<T>::from_f((o < P.Cout && i < P.Cin) ? base[i * P.s_i] : 0.f);
  }
  *reinterpret_cast<typename Elem<T>::frag*>(reinterpret_cast<T*>(P.dst) + (size_t)idx * EPL) = v;
}

struct PackGcnBwd {
  const float* src; void* dst;
  long long s_k, s_c, s_i;     // element strides of W3[k][c][i]
  int Cin, Cout, K, CCi, nchi, CCc, nchc, NKGc, MTK;
  int rc;                      // the register-chained layout of gcn_rc_bwd.hip follows
};

template <typename T>
__global__ __launch_bounds__(256) void pack_tconv_kernel(const PackTconv P) { pack_tconv_body<T>(P, blockIdx.x * 256 + threadIdx.x); }

template <typename T>
__device__ static inline void pack_gcn_bwd_body(const PackGcnBwd& P, int idx) {
  constexpr int EPL = Elem<T>::EPL;
  const int total = P.nchi * P.nchc * P.NKGc * P.MTK * 2 * 32;
  if (idx >= total) {
    // register-chained section [it][k][s][lane][8]: lane (c = lane & 31, h = lane >> 5) of fragment (it, k, s) holds
    // W3[k][16 s + 8 h + e][32 it + p(c)], p = c with bits 2 and 3 swapped (see gcn_rc_bwd.hip)
    if (!P.rc) return;
    const int r = idx - total, SO = P.Cout / 16;
    if (r >= P.K * ((P.Cin + 31) / 32) * SO * 64) return;   // (the 3-channel first layer: one zero-padded tile)
    const int lane = r & 63;
    int f = r >> 6;
    const int s = f % SO; f /= SO;
    const int k = f % P.K; const int it = f / P.K;
    const int cl = lane & 31;
    const int i = 32 * it + ((cl & ~12) | ((cl & 4) << 1) | ((cl & 8) >> 1)), c0 = 16 * s + 8 * (lane >> 5);
    typename Elem<T>::frag v;
#pragma unroll
    for (int e = 0; e < EPL; ++e) v[e] = Elem<T>::from_f((e < 8 && i < P.Cin) ? P.src[k * P.s_k + (c0 + e) * P.s_c + i * P.s_i] : 0.f);
    *reinterpret_cast<typename Elem<T>::frag*>(reinterpret_cast<T*>(P.dst) + (size_t)idx * EPL) = v;
    return;
  }
  int t = idx;
  const int r = t % 32; t /= 32;
  const int h = t % 2; t /= 2;
  const int m = t % P.MTK; t /= P.MTK;
  const int kg = t % P.NKGc; t /= P.NKGc;
  const int cch = t % P.nchc; const int ich = t / P.nchc;
  const int kk = m * 32 + r;
  const int k = kk / P.CCi, il = kk - k * P.CCi, i = ich * P.CCi + il;
  const bool row_ok = k < P.K && i < P.Cin;
  typename Elem<T>::frag v;
#pragma unroll
  for (int e = 0; e < EPL; ++e) {
    const int c = cch * P.CCc + kg * 2 * EPL + h * EPL + e;
    v[e] = Elem<T>::from_f((row_ok && c < P.Cout) ? P.src[k * P.s_k + c * P.s_c + i * P.s_i] : 0.f);
  }
  *reinterpret_cast<typename Elem<T>::frag*>(reinterpret_cast<T*>(P.dst) + (size_t)idx * EPL) = v;
}

template <typename T>
__global__ __launch_bounds__(256) void pack_gcn_bwd_kernel(const PackGcnBwd P) { pack_gcn_bwd_body<T>(P, blockIdx.x * 256 + threadIdx.x); }

// ---- every weight of a model in ONE launch: a table of job records in device memory (filled on the host by the
//      istgcn_pack_job_* functions, uploaded once; the parameter pointers are stable between optimiser steps), each
//      workgroup finds its job from the table of first-workgroup indices ----
struct PackJob {
  int kind;              // 0 gcn, 1 tconv, 2 gcn_bwd
  int pad_;
  union { PackGcn g; PackTconv t; PackGcnBwd b; } u;
};

template <typename T>
__global__ __launch_bounds__(256) void pack_batch_kernel(const PackJob* __restrict__ jobs, const int* __restrict__ bstart, int njobs) {
  int j = 0;
  while (j + 1 < njobs && (int)blockIdx.x >= bstart[j + 1]) ++j;          // wave-uniform scan of <= a few dozen entries
  const int idx = ((int)blockIdx.x - bstart[j]) * 256 + threadIdx.x;
  const PackJob& J = jobs[j];
  if (J.kind == 0) { const PackGcn P = J.u.g; pack_gcn_body<T>(P, idx); }
  else if (J.kind == 1) { const PackTconv P = J.u.t; pack_tconv_body<T>(P, idx); }
  else { const PackGcnBwd P = J.u.b; pack_gcn_bwd_body<T>(P, idx); }
}

}  // namespace

extern "C" int istgcn_pack_job_bytes(void) { return (int)sizeof(PackJob); }

extern "C" int istgcn_pack_job_gcn(void* rec, const float* src, long long s_o, long long s_k, long long s_i, void* dst, int Cin,
                                   int Cout, int K, int dtype) {
  if (!rec || !src || !dst || Cin < 1 || Cout < 1 || K < 1) return -1;
  int cce, nch, kkp, mttot, epl;
  if (istgcn_gcn_geometry(Cin, Cout, K, dtype, &cce, &nch, &kkp, &mttot, &epl)) return -1;
  PackJob J{};
  J.kind = 0;
  const int rc = istgcn_gcn_rc_layout(Cin, Cout, K, dtype);
  J.u.g = PackGcn{src, dst, s_o, s_k, s_i, Cin, Cout, K, cce, nch, kkp / (2 * epl), mttot, rc};
  *reinterpret_cast<PackJob*>(rec) = J;
  return ceil_div(nch * mttot * J.u.g.NKG * 2 * 32 + (rc ? K * Cout * round_up(Cin, 16) / epl : 0), 256);
}

extern "C" int istgcn_pack_job_tconv(void* rec, const float* src, long long s_t, long long s_o, long long s_i, const int* tap_sel,
                                     void* dst, int V, int Cin, int Cout, int ntaps, const int* tap_off, int in_mul, int dtype) {
  if (!rec || !src || !dst || !tap_sel || !tap_off) return -1;
  int cc, nch, mttot, epl;
  if (istgcn_tconv_geometry(V, Cin, Cout, ntaps, tap_off, in_mul, dtype, &cc, &nch, &mttot, &epl)) return -1;
  PackJob J{};
  J.kind = 1;
  PackTconv& P = J.u.t;
  P.src = src; P.dst = dst; P.s_t = s_t; P.s_o = s_o; P.s_i = s_i;
  P.Cin = Cin; P.Cout = Cout; P.ntaps = ntaps; P.CC = cc; P.nch = nch; P.NKG = cc / (2 * epl); P.MTtot = mttot;
  for (int j = 0; j < ntaps; ++j) {
    if (tap_sel[j] < 0) return -1;
    P.tap_sel[j] = tap_sel[j];
  }
  *reinterpret_cast<PackJob*>(rec) = J;
  return ceil_div(nch * ntaps * P.NKG * mttot * 2 * 32, 256);
}

extern "C" int istgcn_pack_job_gcn_bwd(void* rec, const float* src, long long s_k, long long s_c, long long s_i, void* dst, int Cin,
                                       int Cout, int K, int dtype) {
  if (!rec || !src || !dst || Cin < 1 || Cout < 1) return -1;
  int cci, nchi, ccc, nchc, kkp, epl;
  if (istgcn_gcn_bwd_geometry(Cin, Cout, K, dtype, &cci, &nchi, &ccc, &nchc, &kkp, &epl)) return -1;
  PackJob J{};
  J.kind = 2;
  const int rc = istgcn_gcn_bwd_rc_layout(Cin, Cout, K, dtype);
  J.u.b = PackGcnBwd{src, dst, s_k, s_c, s_i, Cin, Cout, K, cci, nchi, ccc, nchc, ccc / (2 * epl), kkp / 32, rc};
  *reinterpret_cast<PackJob*>(rec) = J;
  return ceil_div(nchi * nchc * J.u.b.NKGc * J.u.b.MTK * 2 * 32 + (rc ? K * Cout * round_up(Cin, 32) / 8 : 0), 256);
}

extern "C" int istgcn_pack_batch(const void* jobs_dev, const int* block_start_dev, int njobs, int total_blocks, int dtype,
                                 void* stream) {
  if (!jobs_dev || !block_start_dev || njobs < 1 || total_blocks < 1 || !istgcn_dtype_ok(dtype)) return ISTGCN_EINVAL;
  const PackJob* jobs = reinterpret_cast<const PackJob*>(jobs_dev);
  if (dtype == 0) ISTGCN_LAUNCH(pack_batch_kernel<float>, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, jobs, block_start_dev, njobs);
  else if (dtype == 2) ISTGCN_LAUNCH(pack_batch_kernel<_Float16>, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, jobs, block_start_dev, njobs);
  else ISTGCN_LAUNCH(pack_batch_kernel<__bf16>, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, jobs, block_start_dev, njobs);
  ISTGCN_CHECK_LAUNCH();
  return ISTGCN_OK;
}

extern "C" long long istgcn_pack_gcn_elems(int Cin, int Cout, int K, int dtype) {
  int cce, nch, kkp, mttot, epl;
  if (Cin < 1 || Cout < 1 || K < 1 || istgcn_gcn_geometry(Cin, Cout, K, dtype, &cce, &nch, &kkp, &mttot, &epl)) return -1;
  return (long long)nch * mttot * kkp * 32 + (istgcn_gcn_rc_layout(Cin, Cout, K, dtype) ? (long long)K * Cout * round_up(Cin, 16) : 0);
}

// Element offset of the register-chained section inside the packed graph-conv weights, or -1 if there is none.
extern "C" long long istgcn_gcn_rc_offset(int Cin, int Cout, int K, int dtype) {
  int cce, nch, kkp, mttot, epl;
  if (Cin < 1 || Cout < 1 || K < 1 || !istgcn_gcn_rc_layout(Cin, Cout, K, dtype)) return -1;
  if (istgcn_gcn_geometry(Cin, Cout, K, dtype, &cce, &nch, &kkp, &mttot, &epl)) return -1;
  return (long long)nch * mttot * kkp * 32;
}

extern "C" int istgcn_pack_gcn(const float* src, long long s_o, long long s_k, long long s_i, void* dst, int Cin,
                               int Cout, int K, int dtype, void* stream) {
  if (!src || !dst || Cin < 1 || Cout < 1 || K < 1) return ISTGCN_EINVAL;
  int cce, nch, kkp, mttot, epl;
  if (int rc = istgcn_gcn_geometry(Cin, Cout, K, dtype, &cce, &nch, &kkp, &mttot, &epl)) return rc;
  const int rc = istgcn_gcn_rc_layout(Cin, Cout, K, dtype);
  PackGcn P{src, dst, s_o, s_k, s_i, Cin, Cout, K, cce, nch, kkp / (2 * epl), mttot, rc};
  const int total = nch * mttot * P.NKG * 2 * 32 + (rc ? K * Cout * round_up(Cin, 16) / epl : 0);
  if (dtype == 0) ISTGCN_LAUNCH(pack_gcn_kernel<float>, dim3(ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, P);
  else if (dtype == 2) ISTGCN_LAUNCH(pack_gcn_kernel<_Float16>, dim3(ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, P);
  else ISTGCN_LAUNCH(pack_gcn_kernel<__bf16>, dim3(ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, P);
  ISTGCN_CHECK_LAUNCH();
  return ISTGCN_OK;
}

extern "C" long long istgcn_pack_tconv_elems(int V, int Cin, int Cout, int ntaps, const int* tap_off, int in_mul,
                                             int dtype) {
  int cc, nch, mttot, epl;
  if (istgcn_tconv_geometry(V, Cin, Cout, ntaps, tap_off, in_mul, dtype, &cc, &nch, &mttot, &epl)) return -1;
  return (long long)nch * ntaps * cc * mttot * 32;
}

extern "C" int istgcn_pack_tconv(const float* src, long long s_t, long long s_o, long long s_i, const int* tap_sel,
                                 void* dst, int V, int Cin, int Cout, int ntaps, const int* tap_off, int in_mul,
                                 int dtype, void* stream) {
  if (!src || !dst || !tap_sel || !tap_off) return ISTGCN_EINVAL;
  int cc, nch, mttot, epl;
  if (int rc = istgcn_tconv_geometry(V, Cin, Cout, ntaps, tap_off, in_mul, dtype, &cc, &nch, &mttot, &epl)) return rc;
  PackTconv P{};
  P.src = src; P.dst = dst; P.s_t = s_t; P.s_o = s_o; P.s_i = s_i;
  P.Cin = Cin; P.Cout = Cout; P.ntaps = ntaps; P.CC = cc; P.nch = nch; P.NKG = cc / (2 * epl); P.MTtot = mttot;
  for (int j = 0; j < ntaps; ++j) {
    if (tap_sel[j] < 0) return ISTGCN_EINVAL;
    P.tap_sel[j] = tap_sel[j];
  }
  const int total = nch * ntaps * P.NKG * mttot * 2 * 32;
  if (dtype == 0) ISTGCN_LAUNCH(pack_tconv_kernel<float>, dim3(ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, P);
  else if (dtype == 2) ISTGCN_LAUNCH(pack_tconv_kernel<_Float16>, dim3(ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, P);
  else ISTGCN_LAUNCH(pack_tconv_kernel<__bf16>, dim3(ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, P);
  ISTGCN_CHECK_LAUNCH();
  return ISTGCN_OK;
}

extern "C" long long istgcn_pack_gcn_bwd_elems(int Cin, int Cout, int K, int dtype) {
  int cci, nchi, ccc, nchc, kkp, epl;
  if (Cin < 1 || Cout < 1 || istgcn_gcn_bwd_geometry(Cin, Cout, K, dtype, &cci, &nchi, &ccc, &nchc, &kkp, &epl)) return -1;
  return (long long)nchi * nchc * ccc * kkp + (istgcn_gcn_bwd_rc_layout(Cin, Cout, K, dtype) ? (long long)K * Cout * round_up(Cin, 32) : 0);
}

// Element offset of the register-chained section inside the packed weights of istgcn_gcn_bwd_data, or -1 if there is none.
extern "C" long long istgcn_gcn_bwd_rc_offset(int Cin, int Cout, int K, int dtype) {
  int cci, nchi, ccc, nchc, kkp, epl;
  if (Cin < 1 || Cout < 1 || !istgcn_gcn_bwd_rc_layout(Cin, Cout, K, dtype)) return -1;
  if (istgcn_gcn_bwd_geometry(Cin, Cout, K, dtype, &cci, &nchi, &ccc, &nchc, &kkp, &epl)) return -1;
  return (long long)nchi * nchc * ccc * kkp;
}

extern "C" int istgcn_pack_gcn_bwd(const float* src, long long s_k, long long s_c, long long s_i, void* dst, int Cin,
                                   int Cout, int K, int dtype, void* stream) {
  if (!src || !dst || Cin < 1 || Cout < 1) return ISTGCN_EINVAL;
  int cci, nchi, ccc, nchc, kkp, epl;
  if (int rc = istgcn_gcn_bwd_geometry(Cin, Cout, K, dtype, &cci, &nchi, &ccc, &nchc, &kkp, &epl)) return rc;
  const int rc = istgcn_gcn_bwd_rc_layout(Cin, Cout, K, dtype);
  PackGcnBwd P{src, dst, s_k, s_c, s_i, Cin, Cout, K, cci, nchi, ccc, nchc, ccc / (2 * epl), kkp / 32, rc};
  const int total = nchi * nchc * P.NKGc * P.MTK * 2 * 32 + (rc ? K * Cout * round_up(Cin, 32) / 8 : 0);
  if (dtype == 0) ISTGCN_LAUNCH(pack_gcn_bwd_kernel<float>, dim3(ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, P);
  else if (dtype == 2) ISTGCN_LAUNCH(pack_gcn_bwd_kernel<_Float16>, dim3(ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, P);
  else ISTGCN_LAUNCH(pack_gcn_bwd_kernel<__bf16>, dim3(ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, P);
  ISTGCN_CHECK_LAUNCH();
  return ISTGCN_OK;
}
