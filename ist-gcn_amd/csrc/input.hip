// Input stage of the Model on the GPU: the feeder's augmentation and the data_bn prologue in (at most) two passes over
// the raw clip batch, writing the engine's NTVC activation directly.
//
//   reference                                                                  here
//   feeder/tools.py:31-41   auto_pading  (zero pad to the window, begin = 0)   frame shift + zero outside [0, Traw)
//   feeder/tools.py:44-57   random_choose (random crop / random pad offset)    the same shift, drawn on the host
//   feeder/tools.py:60-101  random_move  (per-frame rotation/scale/shift of    per-(clip, frame) 2x3 affine on channels
//                                         channels 0,1; Python loop over T)     0,1, coefficients computed on the host
//   net/st_gcnold.py:74-80  permute -> BatchNorm1d(V*C) -> permute             batch sums (pass 1), affine + layout
//                                                                              change to [N*M][T][V][C] (pass 2)
//   autograd of data_bn                                                        sum(dy), sum(dy*xhat) per (v,c) channel
//
// Layouts: raw / augmented clips are the reference's (N, C, T, V, M) fp32; the BatchNorm channel of (v, c) is v*C + c
// (x.permute(0,4,3,1,2).view(N*M, V*C, T), st_gcnold.py:75-76); the output is the NTVC tensor every block consumes.
// The tensors here are tiny next to the activations (C = 3: 11.5 MB in, 5.8 MB out at batch 64), so the kernels are
// written for few launches, not for the last GB/s: one workgroup = one clip x a run of TT frames, one thread = one
// (channel, joint, person) column of that run, whose global reads are contiguous across the workgroup.
#include "common.hpp"

namespace {

constexpr int TT = 16;              // frames per workgroup

struct InParams {
  const float* raw;      // [N][C][Traw][V][M]
  const int* shift;      // [N] or null: source frame = t + shift[n]   (crop: +begin; pad: -begin)
  const double* move;    // [N][T][6] or null: (m00, m01, tx, m10, m11, ty) of frame t
  int N, C, Traw, T, V, M;
};

// augmented value of element (n, c, t, vm): tools.py:88-99 computes theta . xy + t in float64 and stores it back into
// the float32 clip, so the affine runs in double and is rounded once
__device__ static inline float fetch(const InParams& P, int n, int c, int t, int vm) {
  const int VM = P.V * P.M;
  const int ts = t + (P.shift ? P.shift[n] : 0);
  const bool inb = ts >= 0 && ts < P.Traw;
  const size_t base = ((size_t)n * P.C * P.Traw + (inb ? ts : 0)) * VM + vm;
  const size_t cstride = (size_t)P.Traw * VM;
  if (P.move && c < 2 && P.C >= 2) {
    const double x0 = inb ? (double)P.raw[base] : 0.0;
    const double x1 = inb ? (double)P.raw[base + cstride] : 0.0;
    const double* mv = P.move + ((size_t)n * P.T + t) * 6 + 3 * c;
    return (float)((mv[0] * x0 + mv[1] * x1) + mv[2]);
  }
  return inb ? P.raw[base + c * cstride] : 0.f;
}

__global__ void feeder_augment_kernel(const InParams P, float* __restrict__ out) {
  const int VM = P.V * P.M;
  const size_t total = (size_t)P.N * P.C * P.T * VM;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int vm = (int)(i % VM);
    size_t r = i / VM;
    const int t = (int)(r % P.T); r /= P.T;
    const int c = (int)(r % P.C);
    const int n = (int)(r / P.C);
    out[i] = fetch(P, n, c, t, vm);
  }
}

// pass 1: per BatchNorm channel (v*C + c) sum and sum of squares over (n, m, t)
__global__ void input_stats_kernel(const InParams P, double* __restrict__ stats, int rep) {
  extern __shared__ float red[];                       // [2][C*V*M]
  const int VM = P.V * P.M, cols = P.C * VM;
  const int n = blockIdx.y, t0 = blockIdx.x * TT;
  const int tid = threadIdx.x;
  if (tid < cols) {
    const int c = tid / VM, vm = tid - c * VM;
    float s = 0.f, ss = 0.f;
    const int t1 = min(P.T, t0 + TT);
    for (int t = t0; t < t1; ++t) {
      const float v = fetch(P, n, c, t, vm);
      s += v; ss += v * v;
    }
    red[tid] = s; red[cols + tid] = ss;
  }
  __syncthreads();
  const int VC = P.V * P.C;
  if (tid < VC) {
    const int v = tid / P.C, c = tid - v * P.C;
    float s = 0.f, ss = 0.f;
    for (int m = 0; m < P.M; ++m) { s += red[c * VM + v * P.M + m]; ss += red[cols + c * VM + v * P.M + m]; }
    double* dst = stats + (size_t)((blockIdx.x + blockIdx.y) % rep) * 2 * VC;
    atomic_add_f64(dst + tid, (double)s);
    atomic_add_f64(dst + VC + tid, (double)ss);
  }
}

// pass 2: y = x*scale[v*C+c] + shift[v*C+c] written as out[n*M+m][t][v][c]
template <typename T>
__global__ void input_apply_kernel(const InParams P, const float* __restrict__ coef, T* __restrict__ out) {
  extern __shared__ float tile[];                      // [TT][M][V*C]
  const int VM = P.V * P.M, cols = P.C * VM, VC = P.V * P.C;
  const int n = blockIdx.y, t0 = blockIdx.x * TT;
  const int nt = min(TT, P.T - t0);
  const int tid = threadIdx.x;
  if (tid < cols) {
    const int c = tid / VM, vm = tid - c * VM;
    const int v = vm / P.M, m = vm - v * P.M;
    const float sc = coef[v * P.C + c], sh = coef[VC + v * P.C + c];
    for (int t = 0; t < nt; ++t) tile[(t * P.M + m) * VC + v * P.C + c] = fetch(P, n, c, t0 + t, vm) * sc + sh;
  }
  __syncthreads();
  const int per_m = nt * VC;
  for (int i = tid; i < P.M * per_m; i += blockDim.x) {
    const int m = i / per_m, r = i - m * per_m;          // r = t*VC + vc: contiguous in HBM for one person
    const int t = r / VC, vc = r - t * VC;
    out[((size_t)(n * P.M + m) * P.T + t0) * VC + r] = Elem<T>::from_f(tile[(t * P.M + m) * VC + vc]);
  }
}

// backward of data_bn: per channel sum(dy) and sum(dy * xhat), xhat = (x - mean)*rstd recomputed from the raw clip
template <typename T>
__global__ void input_bwd_kernel(const InParams P, const T* __restrict__ dout, const float* __restrict__ coef,
                                 double* __restrict__ stats, int rep) {
  extern __shared__ float red[];                       // [2][C*V*M]
  const int VM = P.V * P.M, cols = P.C * VM, VC = P.V * P.C;
  const int n = blockIdx.y, t0 = blockIdx.x * TT;
  const int tid = threadIdx.x;
  if (tid < cols) {
    const int c = tid / VM, vm = tid - c * VM;
    const int v = vm / P.M, m = vm - v * P.M;
    const float mean = coef[2 * VC + v * P.C + c], rstd = coef[3 * VC + v * P.C + c];
    float s = 0.f, sx = 0.f;
    const int t1 = min(P.T, t0 + TT);
    for (int t = t0; t < t1; ++t) {
      const float x = fetch(P, n, c, t, vm);
      const float d = Elem<T>::to_f(dout[((size_t)(n * P.M + m) * P.T + t) * VC + v * P.C + c]);
      s += d; sx += d * (x - mean) * rstd;
    }
    red[tid] = s; red[cols + tid] = sx;
  }
  __syncthreads();
  if (tid < VC) {
    const int v = tid / P.C, c = tid - v * P.C;
    float s = 0.f, sx = 0.f;
    for (int m = 0; m < P.M; ++m) { s += red[c * VM + v * P.M + m]; sx += red[cols + c * VM + v * P.M + m]; }
    double* dst = stats + (size_t)((blockIdx.x + blockIdx.y) % rep) * 2 * VC;
    atomic_add_f64(dst + tid, (double)s);
    atomic_add_f64(dst + VC + tid, (double)sx);
  }
}

static inline bool in_ok(const float* raw, int N, int C, int Traw, int T, int V, int M) {
  return raw && N >= 0 && C >= 1 && Traw >= 1 && T >= 1 && V >= 1 && M >= 1 && (long long)C * V * M <= 1024;
}

static inline InParams mk(const float* raw, const int* shift, const double* move, int N, int C, int Traw, int T, int V,
                          int M) {
  InParams P;
  P.raw = raw; P.shift = shift; P.move = move; P.N = N; P.C = C; P.Traw = Traw; P.T = T; P.V = V; P.M = M;
  return P;
}

}  // namespace

extern "C" int istgcn_feeder_augment(const float* raw, const int* shift, const double* move, float* out, int N, int C,
                                     int Traw, int T, int V, int M, void* stream) {
  if (!in_ok(raw, N, C, Traw, T, V, M) || !out) return ISTGCN_EINVAL;
  if (N == 0) return ISTGCN_OK;
  const size_t total = (size_t)N * C * T * V * M;
  size_t g = (total + 255) / 256;
  if (g > 2048) g = 2048;
  ISTGCN_LAUNCH(feeder_augment_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, mk(raw, shift, move, N, C, Traw, T, V, M),
                out);
  ISTGCN_CHECK_LAUNCH();
  return ISTGCN_OK;
}

extern "C" int istgcn_input_stats(const float* raw, const int* shift, const double* move, double* stats, int stats_rep,
                                  int N, int C, int Traw, int T, int V, int M, void* stream) {
  if (!in_ok(raw, N, C, Traw, T, V, M) || !stats || stats_rep < 1) return ISTGCN_EINVAL;
  if (N == 0) return ISTGCN_OK;
  const int cols = C * V * M, threads = round_up(cols, 64);
  ISTGCN_LAUNCH(input_stats_kernel, dim3(ceil_div(T, TT), N), dim3(threads), (size_t)2 * cols * 4, (hipStream_t)stream,
                mk(raw, shift, move, N, C, Traw, T, V, M), stats, stats_rep);
  ISTGCN_CHECK_LAUNCH();
  return ISTGCN_OK;
}

extern "C" int istgcn_input_apply(const float* raw, const int* shift, const double* move, const float* coef, void* out,
                                  int N, int C, int Traw, int T, int V, int M, int dtype, void* stream) {
  if (!in_ok(raw, N, C, Traw, T, V, M) || !coef || !out || !istgcn_dtype_ok(dtype)) return ISTGCN_EINVAL;
  if (N == 0) return ISTGCN_OK;
  const int cols = C * V * M, threads = round_up(cols, 64);
  const size_t lds = (size_t)TT * cols * 4;
  const InParams P = mk(raw, shift, move, N, C, Traw, T, V, M);
  const dim3 grid(ceil_div(T, TT), N);
  if (dtype == 0) ISTGCN_LAUNCH(input_apply_kernel<float>, grid, dim3(threads), lds, (hipStream_t)stream, P, coef, (float*)out);
  else if (dtype == 1) ISTGCN_LAUNCH(input_apply_kernel<__bf16>, grid, dim3(threads), lds, (hipStream_t)stream, P, coef, (__bf16*)out);
  else ISTGCN_LAUNCH(input_apply_kernel<_Float16>, grid, dim3(threads), lds, (hipStream_t)stream, P, coef, (_Float16*)out);
  ISTGCN_CHECK_LAUNCH();
  return ISTGCN_OK;
}

extern "C" int istgcn_input_bwd(const float* raw, const int* shift, const double* move, const void* dout,
                                const float* coef, double* stats, int stats_rep, int N, int C, int Traw, int T, int V,
                                int M, int dtype, void* stream) {
  if (!in_ok(raw, N, C, Traw, T, V, M) || !dout || !coef || !stats || stats_rep < 1 || !istgcn_dtype_ok(dtype))
    return ISTGCN_EINVAL;
  if (N == 0) return ISTGCN_OK;
  const int cols = C * V * M, threads = round_up(cols, 64);
  const size_t lds = (size_t)2 * cols * 4;
  const InParams P = mk(raw, shift, move, N, C, Traw, T, V, M);
  const dim3 grid(ceil_div(T, TT), N);
  if (dtype == 0) ISTGCN_LAUNCH(input_bwd_kernel<float>, grid, dim3(threads), lds, (hipStream_t)stream, P, (const float*)dout, coef, stats, stats_rep);
  else if (dtype == 1) ISTGCN_LAUNCH(input_bwd_kernel<__bf16>, grid, dim3(threads), lds, (hipStream_t)stream, P, (const __bf16*)dout, coef, stats, stats_rep);
  else ISTGCN_LAUNCH(input_bwd_kernel<_Float16>, grid, dim3(threads), lds, (hipStream_t)stream, P, (const _Float16*)dout, coef, stats, stats_rep);
  ISTGCN_CHECK_LAUNCH();
  return ISTGCN_OK;
}
