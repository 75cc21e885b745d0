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

__global__ __launch_bounds__(NT) void fold_fwd_kernel(const FoldParams P) {
  __shared__ float colsum[MAXKV];
  extern __shared__ float a_l[];
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

__global__ __launch_bounds__(NT) void fold_bwd_kernel(const FoldParams P) {
  __shared__ float colsum[MAXKV];
  __shared__ float dcol[MAXKV];
  extern __shared__ float a_l[];
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
