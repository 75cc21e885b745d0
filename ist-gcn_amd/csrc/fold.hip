// Parameter folds of the graph-conv unit, forward and backward, one small launch each (they were ~18 framework
// launches per block and step: elementwise mul/add chains, a reduction, a tiny GEMM and their autograd mirrors).
//
//   A_eff[k][v][w] = sum_j B_j[k][v][w] * imp_j[k][v][w]        J = 1: A*importance            net/st_gcnold.py:86
//                                                               J = 3: A, A2, A3 (Inception-GCN, st_gcn_msgcn.py:116-117)
//                                                                      or A, A^2, A^3 elementwise (tgcn_multi3_fix_3A.py:86-89)
//   bterm[w][c]    = sum_k bias[k*C + c] * sum_v A_eff[k][v][w]  the Conv2d bias pushed through the einsum
//                                                               (net/utils/tgcn.py:79-86: conv bias, then 'nkctv,kvw->nctw')
// Backward (dA = gradient w.r.t. A_eff from istgcn_gcn_bwd_data, S = gradient w.r.t. bterm from istgcn_gcn_wgrad):
//   dimp_j = B_j (.) (dA + 1_v (x) dcol),  dcol[k][w] = sum_c S[w][c] * bias[k*C+c],
//   dbias[k*C+c] = sum_w S[w][c] * colsum[k][w].
#include "common.hpp"

namespace {

constexpr int NT = 1024;
constexpr int MAXKV = 4 * 128;

struct FoldParams {
  const float* B;        // [J][K][V][V]
  const float* imp[3];   // J pointers, each [K][V][V]
  const float* bias;     // [K*C] or null
  float* A_eff;          // fwd out [K][V][V]
  float* bterm;          // fwd out [V][C] (null without bias)
  const float* dA;       // bwd in [K][V][V] or null
  const float* S;        // bwd in [V][C] or null
  float* dimp[3];        // bwd out
  float* dbias;          // bwd out [K*C] or null
  int J, K, V, C;
};

// A_eff = sum_j B_j (.) imp_j into LDS (and optionally HBM), then its column sums colsum[k][w] = sum_v A_eff[k][v][w]:
// coalesced element pass, then one thread per (k,w) walking LDS (stride V floats: conflict-free for odd V, 2-way else)
__device__ static inline void fold_and_sum(const FoldParams& P, float* a_l, float* colsum, float* a_out) {
  const int VV = P.V * P.V, KVV = P.K * VV, KV = P.K * P.V;
  for (int e = threadIdx.x; e < KVV; e += NT) {
    float a = 0.f;
    for (int j = 0; j < P.J; ++j) a += P.B[(size_t)j * KVV + e] * P.imp[j][e];
    a_l[e] = a;
    if (a_out) a_out[e] = a;
  }
  __syncthreads();
  for (int kw = threadIdx.x; kw < KV; kw += NT) {
    const int k = kw / P.V, w = kw - k * P.V;
    const float* col = a_l + k * VV + w;
    float s0 = 0.f, s1 = 0.f;
    int v = 0;
    for (; v + 1 < P.V; v += 2) { s0 += col[v * P.V]; s1 += col[(v + 1) * P.V]; }
    if (v < P.V) s0 += col[v * P.V];
    colsum[kw] = s0 + s1;
  }
  __syncthreads();
}

// K*V*V <= MAXA floats (48 KiB) of LDS for A_eff: V <= 64 with K = 3; the skeleton graphs have V <= 25
constexpr int MAXA = 3 * 64 * 64;

__device__ static inline void fold_fwd_body(const FoldParams& P, float* colsum, float* a_l) {
  if (!P.bias) {
    const int KVV = P.K * P.V * P.V;
    for (int e = threadIdx.x; e < KVV; e += NT) {
      float a = 0.f;
      for (int j = 0; j < P.J; ++j) a += P.B[(size_t)j * KVV + e] * P.imp[j][e];
      P.A_eff[e] = a;
    }
    return;
  }
  fold_and_sum(P, a_l, colsum, P.A_eff);
  for (int idx = threadIdx.x; idx < P.V * P.C; idx += NT) {       // consecutive threads -> consecutive c: coalesced
    const int w = idx / P.C, c = idx - w * P.C;
    float s = 0.f;
    for (int k = 0; k < P.K; ++k) s += P.bias[k * P.C + c] * colsum[k * P.V + w];
    P.bterm[idx] = s;
  }
}

__global__ __launch_bounds__(NT) void fold_fwd_kernel(const FoldParams P) {
  __shared__ float colsum[MAXKV];
  extern __shared__ float a_l[];
  fold_fwd_body(P, colsum, a_l);
}

// every block of a model in one launch: one workgroup per block (the folds only depend on parameters, so all blocks'
// A_eff / bterm can be produced at the top of the forward pass, and all their gradients when the last one is known)
constexpr int MAXB = 16;
struct FoldBatch { FoldParams p[MAXB]; };

__global__ __launch_bounds__(NT) void fold_fwd_batch_kernel(const FoldBatch PB) {
  __shared__ float colsum[MAXKV];
  extern __shared__ float a_l[];
  fold_fwd_body(PB.p[blockIdx.x], colsum, a_l);
}

__device__ static inline void fold_bwd_body(const FoldParams& P, float* colsum, float* dcol, float* a_l) {
  const int KV = P.K * P.V, VV = P.V * P.V, KVV = P.K * VV;
  const bool hb = P.bias && P.S;
  if (hb) {
    fold_and_sum(P, a_l, colsum, nullptr);
    // dcol[k][w] = sum_c S[w][c] * bias[k][c]: one WAVE per (k,w), lanes stride the channels (coalesced), shuffle-reduce
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int kw = wave; kw < KV; kw += NT / 64) {
      const int k = kw / P.V, w = kw - k * P.V;
      float s = 0.f;
      for (int c = lane; c < P.C; c += 64) s += P.S[w * P.C + c] * P.bias[k * P.C + c];
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m);
      if (lane == 0) dcol[kw] = s;
    }
    if (P.dbias) {
      // dbias[k][c] = sum_w S[w][c] * colsum[k][w]: thread per (k,c), the S reads of a wave are one contiguous row piece
      for (int kc = threadIdx.x; kc < P.K * P.C; kc += NT) {
        const int k = kc / P.C, c = kc - k * P.C;
        float s0 = 0.f, s1 = 0.f;
        int w = 0;
        for (; w + 1 < P.V; w += 2) {
          s0 += P.S[w * P.C + c] * colsum[k * P.V + w];
          s1 += P.S[(w + 1) * P.C + c] * colsum[k * P.V + w + 1];
        }
        if (w < P.V) s0 += P.S[w * P.C + c] * colsum[k * P.V + w];
        P.dbias[kc] = s0 + s1;
      }
    }
    __syncthreads();
  }
  for (int e = threadIdx.x; e < KVV; e += NT) {
    const int k = e / VV, w = e % P.V;
    const float g = (P.dA ? P.dA[e] : 0.f) + (hb ? dcol[k * P.V + w] : 0.f);
    for (int j = 0; j < P.J; ++j) P.dimp[j][e] = P.B[(size_t)j * KVV + e] * g;
  }
}

__global__ __launch_bounds__(NT) void fold_bwd_kernel(const FoldParams P) {
  __shared__ float colsum[MAXKV];
  __shared__ float dcol[MAXKV];
  extern __shared__ float a_l[];
  fold_bwd_body(P, colsum, dcol, a_l);
}

__global__ __launch_bounds__(NT) void fold_bwd_batch_kernel(const FoldBatch PB) {
  __shared__ float colsum[MAXKV];
  __shared__ float dcol[MAXKV];
  extern __shared__ float a_l[];
  fold_bwd_body(PB.p[blockIdx.x], colsum, dcol, a_l);
}

// ---- the three-branch Inception-TCN as ONE 15-tap convolution (linear in the weights):
//      taps[j][o][i] = scale * (m0*W1[o][i][j-6] + m1*W2[o][i][j-3] + m2*W3[o][i][j])   (branches of 3 / 9 / 15 taps,
//      paddings 1 / 4 / 7 -> tap offsets 6 / 3 / 0),  bias = scale * (m0*b1 + m1*b2 + m2*b3)
//      net/st_gcn_multi3_fix_3A_mstcn.py:160-180,212-215 (scale = 1), net/st_gcn_mstcn.py:189-209,242-245 (scale = 1/3)
struct TcnFold {
  const float *w1, *w2, *w3, *b1, *b2, *b3, *mst;     // W_s [Co][Ci][k_s] (k = 3, 9, 15), b_s [Co], mst [3]
  float* taps;          // fwd out [15][Co][Ci]
  float* bias;          // fwd out [Co]
  const float* dtaps;   // bwd in [15][Co][Ci]
  const float* dbias;   // bwd in [Co]
  float *dw1, *dw2, *dw3, *db1, *db2, *db3, *dmst;    // bwd out (dmst [3] accumulated: caller zeroes)
  float scale;
  int Co, Ci;
};

// Both kernels: a block owns 256 consecutive (o,i) pairs.  The [o][i][k] parameter tensors are contiguous runs of 256*k
// floats per block and move through LDS (coalesced global accesses; per-thread LDS strides 15 / 9 / 3 floats are odd:
// conflict-free); the [j][o][i] taps are read / written directly (consecutive threads = consecutive addresses).
__device__ static inline void tcn_stage_in(const float* __restrict__ g, float* l, int first, int count, int k) {
  for (int i = threadIdx.x; i < count * k; i += 256) l[i] = g[(size_t)first * k + i];
}
__device__ static inline void tcn_stage_out(float* __restrict__ g, const float* l, int first, int count, int k) {
  for (int i = threadIdx.x; i < count * k; i += 256) g[(size_t)first * k + i] = l[i];
}

__global__ __launch_bounds__(256) void tcn_fold_fwd_kernel(const TcnFold P) {
  __shared__ float s3[256 * 15], s2[256 * 9], s1[256 * 3];
  const int n = P.Co * P.Ci;
  const int first = blockIdx.x * 256, count = min(256, n - first);
  const int oi = first + threadIdx.x, l = threadIdx.x;
  const float m0 = P.mst[0] * P.scale, m1 = P.mst[1] * P.scale, m2 = P.mst[2] * P.scale;
  tcn_stage_in(P.w3, s3, first, count, 15);
  tcn_stage_in(P.w2, s2, first, count, 9);
  tcn_stage_in(P.w1, s1, first, count, 3);
  __syncthreads();
  if (oi < n) {
    float t[15];
#pragma unroll
    for (int j = 0; j < 15; ++j) t[j] = m2 * s3[l * 15 + j];
#pragma unroll
    for (int j = 0; j < 9; ++j) t[j + 3] += m1 * s2[l * 9 + j];
#pragma unroll
    for (int j = 0; j < 3; ++j) t[j + 6] += m0 * s1[l * 3 + j];
#pragma unroll
    for (int j = 0; j < 15; ++j) P.taps[(size_t)j * n + oi] = t[j];
  }
  if (oi < P.Co) P.bias[oi] = m0 * P.b1[oi] + m1 * P.b2[oi] + m2 * P.b3[oi];
}

__global__ __launch_bounds__(256) void tcn_fold_bwd_kernel(const TcnFold P) {
  __shared__ float s3[256 * 15], s2[256 * 9], s1[256 * 3];
  __shared__ float red[3][4];
  const int n = P.Co * P.Ci;
  const int first = blockIdx.x * 256, count = min(256, n - first);
  const int oi = first + threadIdx.x, l = threadIdx.x;
  const float m0 = P.mst[0] * P.scale, m1 = P.mst[1] * P.scale, m2 = P.mst[2] * P.scale;
  tcn_stage_in(P.w3, s3, first, count, 15);
  tcn_stage_in(P.w2, s2, first, count, 9);
  tcn_stage_in(P.w1, s1, first, count, 3);
  __syncthreads();
  float s0 = 0.f, sm1 = 0.f, sm2 = 0.f;
  float d[15];
#pragma unroll
  for (int j = 0; j < 15; ++j) d[j] = oi < n ? P.dtaps[(size_t)j * n + oi] : 0.f;
  if (oi < n) {
    // importance gradients from the staged weights, then the slots are overwritten with the weight gradients
#pragma unroll
    for (int j = 0; j < 15; ++j) { sm2 += d[j] * s3[l * 15 + j]; s3[l * 15 + j] = m2 * d[j]; }
#pragma unroll
    for (int j = 0; j < 9; ++j) { sm1 += d[j + 3] * s2[l * 9 + j]; s2[l * 9 + j] = m1 * d[j + 3]; }
#pragma unroll
    for (int j = 0; j < 3; ++j) { s0 += d[j + 6] * s1[l * 3 + j]; s1[l * 3 + j] = m0 * d[j + 6]; }
  }
  if (oi < P.Co) {
    const float db = P.dbias[oi];
    P.db1[oi] = m0 * db; P.db2[oi] = m1 * db; P.db3[oi] = m2 * db;
    s0 += db * P.b1[oi]; sm1 += db * P.b2[oi]; sm2 += db * P.b3[oi];
  }
  __syncthreads();
  tcn_stage_out(P.dw3, s3, first, count, 15);
  tcn_stage_out(P.dw2, s2, first, count, 9);
  tcn_stage_out(P.dw1, s1, first, count, 3);
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) { s0 += __shfl_xor(s0, m); sm1 += __shfl_xor(sm1, m); sm2 += __shfl_xor(sm2, m); }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { red[0][wave] = s0; red[1][wave] = sm1; red[2][wave] = sm2; }
  __syncthreads();
  if (threadIdx.x < 3) {
    const float v = (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
    atomicAdd(P.dmst + threadIdx.x, v * P.scale);
  }
}

}  // namespace

extern "C" int istgcn_fold_fwd(const float* B, int J, const float* imp0, const float* imp1, const float* imp2,
                               const float* bias, float* A_eff, float* bterm, int K, int V, int C, void* stream) {
  if (!B || !imp0 || !A_eff || J < 1 || J > 3 || (J > 1 && !imp1) || (J > 2 && !imp2)) return ISTGCN_EINVAL;
  if (K < 1 || V < 1 || K * V > MAXKV || K * V * V > MAXA || (bias && (!bterm || C < 1))) return ISTGCN_EINVAL;
  FoldParams P{};
  P.B = B; P.imp[0] = imp0; P.imp[1] = imp1; P.imp[2] = imp2; P.bias = bias; P.A_eff = A_eff; P.bterm = bterm;
  P.J = J; P.K = K; P.V = V; P.C = C;
  ISTGCN_LAUNCH(fold_fwd_kernel, dim3(1), dim3(NT), (size_t)K * V * V * sizeof(float), (hipStream_t)stream, P);
  ISTGCN_CHECK_LAUNCH();
  return ISTGCN_OK;
}

extern "C" int istgcn_fold_bwd(const float* B, int J, const float* imp0, const float* imp1, const float* imp2,
                               const float* bias, const float* dA, const float* S, float* dimp0, float* dimp1,
                               float* dimp2, float* dbias, int K, int V, int C, void* stream) {
  if (!B || !imp0 || !dimp0 || J < 1 || J > 3 || (J > 1 && (!imp1 || !dimp1)) || (J > 2 && (!imp2 || !dimp2)))
    return ISTGCN_EINVAL;
  if (K < 1 || V < 1 || K * V > MAXKV || K * V * V > MAXA || (S && bias && C < 1)) return ISTGCN_EINVAL;
  FoldParams P{};
  P.B = B; P.imp[0] = imp0; P.imp[1] = imp1; P.imp[2] = imp2; P.bias = bias; P.dA = dA; P.S = S;
  P.dimp[0] = dimp0; P.dimp[1] = dimp1; P.dimp[2] = dimp2; P.dbias = dbias;
  P.J = J; P.K = K; P.V = V; P.C = C;
  ISTGCN_LAUNCH(fold_bwd_kernel, dim3(1), dim3(NT), (size_t)K * V * V * sizeof(float), (hipStream_t)stream, P);
  ISTGCN_CHECK_LAUNCH();
  return ISTGCN_OK;
}

extern "C" int istgcn_fold_fwd_batch(int nb, const float* B, int J, const float* const* imps, const float* const* biases,
                                     float* const* A_eff, float* const* bterm, const int* Cs, int K, int V, void* stream) {
  if (nb < 1 || nb > MAXB || !B || !imps || !A_eff || !Cs || J < 1 || J > 3) return ISTGCN_EINVAL;
  if (K < 1 || V < 1 || K * V > MAXKV || K * V * V > MAXA) return ISTGCN_EINVAL;
  FoldBatch PB{};
  for (int i = 0; i < nb; ++i) {
    FoldParams& P = PB.p[i];
    P.B = B;
    for (int j = 0; j < J; ++j) { P.imp[j] = imps[i * 3 + j]; if (!P.imp[j]) return ISTGCN_EINVAL; }
    P.bias = biases ? biases[i] : nullptr;
    P.A_eff = A_eff[i];
    P.bterm = bterm ? bterm[i] : nullptr;
    if (!P.A_eff || (P.bias && (!P.bterm || Cs[i] < 1))) return ISTGCN_EINVAL;
    P.J = J; P.K = K; P.V = V; P.C = Cs[i];
  }
  ISTGCN_LAUNCH(fold_fwd_batch_kernel, dim3(nb), dim3(NT), (size_t)K * V * V * sizeof(float), (hipStream_t)stream, PB);
  ISTGCN_CHECK_LAUNCH();
  return ISTGCN_OK;
}

extern "C" int istgcn_fold_bwd_batch(int nb, const float* B, int J, const float* const* imps, const float* const* biases,
                                     const float* const* dA, const float* const* S, float* const* dimps, float* const* dbias,
                                     const int* Cs, int K, int V, void* stream) {
  if (nb < 1 || nb > MAXB || !B || !imps || !dimps || !Cs || J < 1 || J > 3) return ISTGCN_EINVAL;
  if (K < 1 || V < 1 || K * V > MAXKV || K * V * V > MAXA) return ISTGCN_EINVAL;
  FoldBatch PB{};
  for (int i = 0; i < nb; ++i) {
    FoldParams& P = PB.p[i];
    P.B = B;
    for (int j = 0; j < J; ++j) {
      P.imp[j] = imps[i * 3 + j]; P.dimp[j] = dimps[i * 3 + j];
      if (!P.imp[j] || !P.dimp[j]) return ISTGCN_EINVAL;
    }
    P.bias = biases ? biases[i] : nullptr;
    P.dA = dA ? dA[i] : nullptr;
    P.S = S ? S[i] : nullptr;
    P.dbias = dbias ? dbias[i] : nullptr;
    if (P.S && P.bias && Cs[i] < 1) return ISTGCN_EINVAL;
    P.J = J; P.K = K; P.V = V; P.C = Cs[i];
  }
  ISTGCN_LAUNCH(fold_bwd_batch_kernel, dim3(nb), dim3(NT), (size_t)K * V * V * sizeof(float), (hipStream_t)stream, PB);
  ISTGCN_CHECK_LAUNCH();
  return ISTGCN_OK;
}

extern "C" int istgcn_tcn_fold_fwd(const float* w1, const float* w2, const float* w3, const float* b1, const float* b2,
                                   const float* b3, const float* mst, float scale, float* taps, float* bias, int Co,
                                   int Ci, void* stream) {
  if (!w1 || !w2 || !w3 || !b1 || !b2 || !b3 || !mst || !taps || !bias || Co < 1 || Ci < 1) return ISTGCN_EINVAL;
  TcnFold P{};
  P.w1 = w1; P.w2 = w2; P.w3 = w3; P.b1 = b1; P.b2 = b2; P.b3 = b3; P.mst = mst; P.taps = taps; P.bias = bias;
  P.scale = scale; P.Co = Co; P.Ci = Ci;
  ISTGCN_LAUNCH(tcn_fold_fwd_kernel, dim3(ceil_div(Co * Ci, 256)), dim3(256), 0, (hipStream_t)stream, P);
  ISTGCN_CHECK_LAUNCH();
  return ISTGCN_OK;
}

extern "C" int istgcn_tcn_fold_bwd(const float* dtaps, const float* dbias, const float* w1, const float* w2,
                                   const float* w3, const float* b1, const float* b2, const float* b3, const float* mst,
                                   float scale, float* dw1, float* dw2, float* dw3, float* db1, float* db2, float* db3,
                                   float* dmst, int Co, int Ci, void* stream) {
  if (!dtaps || !dbias || !w1 || !w2 || !w3 || !b1 || !b2 || !b3 || !mst || !dw1 || !dw2 || !dw3 || !db1 || !db2 || !db3 ||
      !dmst || Co < 1 || Ci < 1)
    return ISTGCN_EINVAL;
  TcnFold P{};
  P.w1 = w1; P.w2 = w2; P.w3 = w3; P.b1 = b1; P.b2 = b2; P.b3 = b3; P.mst = mst; P.dtaps = dtaps; P.dbias = dbias;
  P.dw1 = dw1; P.dw2 = dw2; P.dw3 = dw3; P.db1 = db1; P.db2 = db2; P.db3 = db3; P.dmst = dmst;
  P.scale = scale; P.Co = Co; P.Ci = Ci;
  ISTGCN_LAUNCH(tcn_fold_bwd_kernel, dim3(ceil_div(Co * Ci, 256)), dim3(256), 0, (hipStream_t)stream, P);
  ISTGCN_CHECK_LAUNCH();
  return ISTGCN_OK;
}
