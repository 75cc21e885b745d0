// Parameter gradients of the graph-convolution unit (autograd of net/utils/tgcn.py:79-86 and its folded variants):
//     dW[k][c][i]  += sum_{n,t,w} dy[n,t,w,c] * xa_k[n,t,w,i]        xa_k[(t,w)][i] = sum_v A[k][v][w] x[(t,v)][i]
//     dA[k][v][w]  += sum_{n,t,i} x[n,t,v,i] * dxa_k[n,t,w,i]        dxa_k[(t,w)][i] = sum_c W[k][c][i] dy[(t,w)][c]
//                     (only where A[k][v][w] != 0: every importance gradient upstream is A (.) dA,
//                      st_gcnold.py:86, tgcn_multi3_fix_3A.py:86-88, st_gcn_msgcn.py:116-117)
//     S[w][c]      += sum_{n,t} dy[n,t,w,c]                           (gradient of the bias term pushed through the einsum)
// Same skeleton as tconv_wgrad: a workgroup owns one 32x32 (c-tile, i-tile) block, keeps its K accumulator tiles
// in registers across a grid-stride walk over position tiles, each wave contracting its own 32 positions, and
// flushes once with fp32 atomics.  The K "taps" are the K aggregated images xa_k that the same sparse LDS pass
// as the forward kernel builds; a second small MFMA product (contraction over the c-tile) gives the dxa_k rows
// whose dot products with x rows are the adjacency gradient.
#include "common.hpp"

namespace {

constexpr int NTHREADS = 256;
constexpr int TR = 128;
constexpr int CB = 32;
constexpr int KMAX = 4;

struct GwgParams {
  const void* dy;       // [NM][T][V][Cout]
  const void* x;        // [NM][T][V][Cin]
  const float* A;       // [K][V][V]
  const void* Wq;       // fragments of W^T for the dxa product (see istgcn.h)
  float* dW;            // [K][Cout][Cin] fp32, accumulated
  float* dA;            // [K][V][V] fp32, accumulated (pattern entries only) or null
  float* S;             // [V][Cout] fp32, accumulated, or null
  int NM, T, V, Cin, Cout, K, nnz_cap;
  int F, tiles_per_seq, total_tiles, n_itile, n_ctile;
  int ds_stride;
  int off_csr_v, off_csr_kw, off_csr_a, off_dacc, off_S, off_rows, off_dys, off_xs, off_xa;
};

template <typename T, int KT>
__global__ __launch_bounds__(NTHREADS) void gcn_wgrad_kernel(const GwgParams P) {
  using E = Elem<T>;
  constexpr int EPL = E::EPL;
  constexpr int KGS = E::KGS;
  constexpr int QV = CB / EPL;
  typedef typename E::frag frag_t;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  int* csr_off = reinterpret_cast<int*>(smem);                                   // [K*V+1]
  unsigned char* csr_v = smem + P.off_csr_v;                                     // [cap]
  unsigned short* csr_kw = reinterpret_cast<unsigned short*>(smem + P.off_csr_kw);  // [cap] column = k*V+w
  float* csr_a = reinterpret_cast<float*>(smem + P.off_csr_a);                   // [cap]
  float* dacc = reinterpret_cast<float*>(smem + P.off_dacc);                     // [cap]
  float* S_l = reinterpret_cast<float*>(smem + P.off_S);                         // [V][CB]
  unsigned char* row_f = smem + P.off_rows;                                      // [TR]
  unsigned char* row_w = row_f + TR;                                             // [TR]
  T* dys = reinterpret_cast<T*>(smem + P.off_dys);                               // [TR][ds_stride]
  T* xs = reinterpret_cast<T*>(smem + P.off_xs);                                 // [TR][CB]
  T* xa = reinterpret_cast<T*>(smem + P.off_xa);                                 // [K][TR][CB]  (later: dxa)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int V = P.V, K = P.K, KV = K * V;
  const int ct = blockIdx.y / P.n_itile, it = blockIdx.y - ct * P.n_itile;
  const int c0 = ct * CB, i0 = it * CB;
  const bool vec = (P.Cin % EPL == 0) && (P.Cout % EPL == 0);
  const int DS = P.ds_stride;

  // ---- adjacency column lists (as in gcn_fwd) + entry -> column map ----
  for (int c = tid; c <= KV; c += NTHREADS) csr_off[c] = 0;
  for (int c = tid; c < P.nnz_cap; c += NTHREADS) dacc[c] = 0.f;
  for (int c = tid; c < V * CB; c += NTHREADS) S_l[c] = 0.f;
  for (int r = tid; r < TR; r += NTHREADS) {
    int f = r / V;
    row_f[r] = (unsigned char)f;
    row_w[r] = (unsigned char)(r - f * V);
  }
  __syncthreads();
  for (int col = tid; col < KV; col += NTHREADS) {
    int k = col / V, w = col - k * V, cnt = 0;
    for (int v = 0; v < V; ++v) cnt += (P.A[(k * V + v) * V + w] != 0.f);
    csr_off[col + 1] = cnt;
  }
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int c = 0; c < KV; ++c) { int n = csr_off[c + 1]; csr_off[c] = run; run += n; }
    csr_off[KV] = run;
  }
  __syncthreads();
  for (int col = tid; col < KV; col += NTHREADS) {
    int k = col / V, w = col - k * V, e = csr_off[col];
    for (int v = 0; v < V; ++v) {
      float a = P.A[(k * V + v) * V + w];
      if (a != 0.f) {
        if (e < P.nnz_cap) { csr_v[e] = (unsigned char)v; csr_a[e] = a; csr_kw[e] = (unsigned short)col; }
        ++e;
      }
    }
  }
  __syncthreads();
  const int nnz = min(csr_off[KV], P.nnz_cap);

  f32x16 acc1[KT];
#pragma unroll
  for (int k = 0; k < KT; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc1[k][r] = 0.f;

  const T* dyg = reinterpret_cast<const T*>(P.dy);
  const T* xg = reinterpret_cast<const T*>(P.x);
  const T* Wq = reinterpret_cast<const T*>(P.Wq);
  constexpr int NKG2 = CB / KGS;

  for (int tile = blockIdx.x; tile < P.total_tiles; tile += gridDim.x) {
    const int n = tile / P.tiles_per_seq;
    const int t0 = (tile - n * P.tiles_per_seq) * P.F;
    const int nf = min(P.F, P.T - t0);
    const int rows = nf * V;

    // ---- stage dy (c-tile) and x (i-tile), zero padded ----
    {
      const size_t pos0 = (size_t)(n * P.T + t0) * V;
      if (vec) {
        stage_block<T, 4, true>(dyg + pos0 * P.Cout + c0, (size_t)P.Cout, P.Cout - c0, dys, DS, TR, 0, rows, QV, nullptr,
                                nullptr, 0, tid, NTHREADS);
        stage_block<T, 4, true>(xg + pos0 * P.Cin + i0, (size_t)P.Cin, P.Cin - i0, xs, CB, TR, 0, rows, QV, nullptr,
                                nullptr, 0, tid, NTHREADS);
      } else {
        stage_block<T, 4, false>(dyg + pos0 * P.Cout + c0, (size_t)P.Cout, P.Cout - c0, dys, DS, TR, 0, rows, QV, nullptr,
                                 nullptr, 0, tid, NTHREADS);
        stage_block<T, 4, false>(xg + pos0 * P.Cin + i0, (size_t)P.Cin, P.Cin - i0, xs, CB, TR, 0, rows, QV, nullptr,
                                 nullptr, 0, tid, NTHREADS);
      }
    }
    __syncthreads();
    // ---- K aggregated images xa_k[p][i] ----
    for (int idx = tid; idx < K * TR * QV; idx += NTHREADS) {
      const int q = idx % QV;
      const int r = (idx / QV) % TR;
      const int k = idx / (QV * TR);
      float sum[EPL];
#pragma unroll
      for (int e = 0; e < EPL; ++e) sum[e] = 0.f;
      if (r < rows) {
        const int f = row_f[r], col = k * V + row_w[r];
        const int e1 = min(csr_off[col + 1], P.nnz_cap);
        for (int en = csr_off[col]; en < e1; ++en) {
          const float a = csr_a[en];
          const frag_t xv = *reinterpret_cast<const frag_t*>(xs + (f * V + csr_v[en]) * CB + q * EPL);
#pragma unroll
          for (int e = 0; e < EPL; ++e) sum[e] += a * E::to_f(xv[e]);
        }
      }
      frag_t o;
#pragma unroll
      for (int e = 0; e < EPL; ++e) o[e] = E::from_f(sum[e]);
      *reinterpret_cast<frag_t*>(xa + (k * TR + r) * CB + q * EPL) = o;
    }
    // S[w][c] partial (unique owner per (w,c): no atomics)
    if (P.S && it == 0) {
      for (int idx = tid; idx < V * CB; idx += NTHREADS) {
        const int w = idx / CB, c = idx - w * CB;
        float s = 0.f;
        for (int f = 0; f < nf; ++f) s += E::to_f(dys[(f * V + w) * DS + c]);
        S_l[idx] += s;
      }
    }
    __syncthreads();

    // ---- product 1: acc1[k][c][i] += dy[p][c] * xa_k[p][i] over this wave's 32 positions ----
    if constexpr (sizeof(T) == 4) {
      const int r = lane & 31, h = lane >> 5;
#pragma unroll 4
      for (int kk = 0; kk < 16; ++kk) {
        const int p = wave * 32 + 2 * kk + h;
        const float a = dys[p * DS + r];
#pragma unroll
        for (int k = 0; k < KT; ++k) {
          if (k < K) {
            const float b = xa[(k * TR + p) * CB + r];
            acc1[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc1[k], 0, 0, 0);
          }
        }
      }
    } else {
      typedef short s16x4 __attribute__((ext_vector_type(4)));
      const int grp = lane >> 4, h = grp >> 1, cblk = (grp & 1) * 16;
      const int q = (lane & 15) >> 2, pp = lane & 3;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int pb = wave * 32 + 16 * kk + 8 * h;
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(dys + (pb + q) * DS + cblk + 4 * pp));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(dys + (pb + 4 + q) * DS + cblk + 4 * pp));
        bf16x4 l4 = __builtin_bit_cast(bf16x4, lo), h4 = __builtin_bit_cast(bf16x4, hi);
        bf16x8 a;
        a[0] = l4[0]; a[1] = l4[1]; a[2] = l4[2]; a[3] = l4[3];
        a[4] = h4[0]; a[5] = h4[1]; a[6] = h4[2]; a[7] = h4[3];
#pragma unroll
        for (int k = 0; k < KT; ++k) {
          if (k < K) {
            s16x4 blo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (__attribute__((address_space(3))) s16x4*)(xa + (k * TR + pb + q) * CB + cblk + 4 * pp));
            s16x4 bhi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (__attribute__((address_space(3))) s16x4*)(xa + (k * TR + pb + 4 + q) * CB + cblk + 4 * pp));
            bf16x4 bl = __builtin_bit_cast(bf16x4, blo), bh = __builtin_bit_cast(bf16x4, bhi);
            bf16x8 b;
            b[0] = bl[0]; b[1] = bl[1]; b[2] = bl[2]; b[3] = bl[3];
            b[4] = bh[0]; b[5] = bh[1]; b[6] = bh[2]; b[7] = bh[3];
            acc1[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1[k], 0, 0, 0);
          }
        }
      }
    }

    if (P.dA) {
      // ---- product 2: dxa_k[p][i] = sum_{c in tile} W[k][c][i] dy[p][c]   (rows i, columns p) ----
      f32x16 acc2[KT];
#pragma unroll
      for (int k = 0; k < KT; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[k][r] = 0.f;
      {
        const T* brow = dys + (wave * 32 + (lane & 31)) * DS + (lane >> 5) * EPL;
#pragma unroll
        for (int kg = 0; kg < NKG2; ++kg) {
          const frag_t b = *reinterpret_cast<const frag_t*>(brow + kg * KGS);
#pragma unroll
          for (int k = 0; k < KT; ++k) {
            if (k < K) {
              const frag_t a = *reinterpret_cast<const frag_t*>(
                  Wq + ((((size_t)(ct * P.n_itile + it) * K + k) * NKG2 + kg) * 64 + lane) * EPL);
              mma_kgroup(acc2[k], a, b);
            }
          }
        }
      }
      __syncthreads();                       // every wave is done reading xa
      {
        const int p = wave * 32 + (lane & 31);
#pragma unroll
        for (int k = 0; k < KT; ++k) {
          if (k < K) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              float v4[4] = {acc2[k][4 * g], acc2[k][4 * g + 1], acc2[k][4 * g + 2], acc2[k][4 * g + 3]};
              store4(xa + (k * TR + p) * CB + 8 * g + 4 * (lane >> 5), v4);
            }
          }
        }
      }
      __syncthreads();
      // ---- adjacency gradient on the pattern: dA_e += sum_f < x[(f,v)][:], dxa_k[(f,w)][:] > ----
      for (int idx = tid; idx < nnz * nf; idx += NTHREADS) {
        const int en = idx / nf, f = idx - en * nf;
        const int col = csr_kw[en];
        const int k = col / V, w = col - k * V;
        const T* xr = xs + (f * V + csr_v[en]) * CB;
        const T* dr = xa + (k * TR + f * V + w) * CB;
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < QV; ++q) {
          const frag_t a = *reinterpret_cast<const frag_t*>(xr + q * EPL);
          const frag_t b = *reinterpret_cast<const frag_t*>(dr + q * EPL);
#pragma unroll
          for (int e = 0; e < EPL; ++e) s += E::to_f(a[e]) * E::to_f(b[e]);
        }
        atomicAdd(&dacc[en], s);
      }
    }
    __syncthreads();
  }

  // ---- flush ----
#pragma unroll
  for (int k = 0; k < KT; ++k) {
    if (k < K) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = c0 + mfma_row(r, lane), i = i0 + (lane & 31);
        if (c < P.Cout && i < P.Cin) atomicAdd(P.dW + ((size_t)k * P.Cout + c) * P.Cin + i, acc1[k][r]);
      }
    }
  }
  if (P.dA) {
    for (int en = tid; en < nnz; en += NTHREADS) {
      const int col = csr_kw[en];
      const int k = col / V, w = col - k * V;
      atomicAdd(P.dA + (k * V + csr_v[en]) * V + w, dacc[en]);
    }
  }
  if (P.S && it == 0) {
    for (int idx = tid; idx < V * CB; idx += NTHREADS) {
      const int w = idx / CB, c = idx - w * CB;
      if (c0 + c < P.Cout) atomicAdd(P.S + w * P.Cout + c0 + c, S_l[idx]);
    }
  }
}

template <typename T>
int launch_T(GwgParams& P, int grid_cap, hipStream_t stream) {
  const int esz = sizeof(T), epl = Elem<T>::EPL;
  P.F = TR / P.V;
  P.tiles_per_seq = ceil_div(P.T, P.F);
  P.total_tiles = P.NM * P.tiles_per_seq;
  P.n_itile = ceil_div(P.Cin, CB);
  P.n_ctile = ceil_div(P.Cout, CB);
  P.ds_stride = CB + epl;
  size_t off = (size_t)(P.K * P.V + 1) * 4;
  off = (off + 15) & ~(size_t)15; P.off_csr_v = (int)off; off += P.nnz_cap;
  off = (off + 15) & ~(size_t)15; P.off_csr_kw = (int)off; off += (size_t)P.nnz_cap * 2;
  off = (off + 15) & ~(size_t)15; P.off_csr_a = (int)off; off += (size_t)P.nnz_cap * 4;
  off = (off + 15) & ~(size_t)15; P.off_dacc = (int)off; off += (size_t)P.nnz_cap * 4;
  off = (off + 15) & ~(size_t)15; P.off_S = (int)off; off += (size_t)P.V * CB * 4;
  off = (off + 15) & ~(size_t)15; P.off_rows = (int)off; off += 2 * TR;
  off = (off + 15) & ~(size_t)15; P.off_dys = (int)off; off += (size_t)TR * P.ds_stride * esz;
  off = (off + 15) & ~(size_t)15; P.off_xs = (int)off; off += (size_t)TR * CB * esz;
  off = (off + 15) & ~(size_t)15; P.off_xa = (int)off; off += (size_t)P.K * TR * CB * esz;
  if (off > 160 * 1024) return ISTGCN_EINVAL;
  const int pairs = P.n_ctile * P.n_itile;
  int gx = grid_cap / pairs;
  if (gx < 1) gx = 1;
  if (gx > P.total_tiles) gx = P.total_tiles;
  dim3 grid(gx, pairs);
#define GO(KTv)                                                                                             \
  do {                                                                                                      \
    auto kfn = gcn_wgrad_kernel<T, KTv>;                                                                    \
    static bool attr_done = false;                                                                          \
    if (!attr_done) {                                                                                       \
      hipError_t ea_ = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
      if (ea_ != hipSuccess) return 2000 + (int)ea_; \
      attr_done = true;                                                                                     \
    }                                                                                                       \
    ISTGCN_LAUNCH(kfn, grid, dim3(NTHREADS), off, stream, P);                                          \
  } while (0)
  if (P.K == 1) GO(1);
  else if (P.K <= 3) GO(3);
  else GO(4);
#undef GO
  ISTGCN_CHECK_LAUNCH();
  return ISTGCN_OK;
}

}  // namespace

extern "C" int istgcn_gcn_wgrad(const void* dy, const void* x, const float* A, const void* Wq, float* dW, float* dA,
                                float* S, int NM, int T, int V, int Cin, int Cout, int K, int nnz_cap, int dtype,
                                int grid_cap, void* stream) {
  if (!dy || !x || !A || !dW) return ISTGCN_EINVAL;
  if (dA && !Wq) return ISTGCN_EINVAL;
  if (V < 1 || V > 128 || Cin < 1 || Cout < 1 || K < 1 || K > KMAX || NM < 0 || T < 0) return ISTGCN_EINVAL;
  if (nnz_cap < 1 || nnz_cap > K * V * V) return ISTGCN_EINVAL;
  if (dtype != 0 && dtype != 1) return ISTGCN_EINVAL;
  if (NM == 0 || T == 0) return ISTGCN_OK;
  GwgParams P{};
  P.dy = dy; P.x = x; P.A = A; P.Wq = Wq; P.dW = dW; P.dA = dA; P.S = S;
  P.NM = NM; P.T = T; P.V = V; P.Cin = Cin; P.Cout = Cout; P.K = K; P.nnz_cap = nnz_cap;
  if (grid_cap < 1) grid_cap = 1024;
  if (dtype == 0) return launch_T<float>(P, grid_cap, (hipStream_t)stream);
  return launch_T<__bf16>(P, grid_cap, (hipStream_t)stream);
}
