// Weight gradient of the temporal convolution:   dWf[j][o][i] += sum_{n,m,v} dz[n, m, v, o] * u[n, in_mul*m + tap_off[j], v, i]
// with u = pre(g) (BatchNorm affine + ReLU recomputed on the fly from the GCN output g, frames outside [0,Tin)
// are zero), and the conv-bias gradient dbias[o] += sum dz.  Autograd of net/st_gcnold.py:167-173 (and of the
// pre-summed 15-tap Inception-TCN, st_gcn_multi3_fix_3A_mstcn.py:160-180,212-215).
//
// This is a GEMM whose contraction runs over every position of the batch (millions) and whose output is tiny,
// so: a workgroup owns ONE 32x32 (o-tile, i-tile) block for ALL taps and keeps the ntaps accumulator tiles in
// registers while it walks position tiles in a grid-stride loop; each of its 4 waves contracts its own quarter
// of the tile's positions (no intra-tile reduction), and the per-wave partial tiles are flushed ONCE at the end
// with fp32 atomics shaped as two 128-byte row segments per instruction.  dz fragments are reused across taps;
// the shifted u fragments come from one staged halo tile.  bf16 operands are row-major [position][channel] in LDS
// and reach the MFMA k axis through ds_read_b64_tr_b16.
#include "common.hpp"

namespace {

constexpr int NTHREADS = 256;
constexpr int TR = 128;
constexpr int MAX_TAPS = 16;

struct TwgParams {
  const void* dz;        // [NM][Tz][V][Cout]
  const void* g;         // [NM][Tin][V][Cin]
  const float* pre;      // [2][Cin] or null
  float* dW;             // [ntaps][Cout][Cin] fp32, caller-zeroed
  float* dbias;          // [Cout] or null
  int NM, Tin, Tz, V, Cin, Cout, ntaps, in_mul, pre_relu;
  int tap_off[MAX_TAPS];
  int F, tiles_per_seq, total_tiles, min_off, Fin, n_itile;
  int off_urow, off_dz, off_u;   // LDS byte offsets
};

template <typename T, int JT>
__global__ __launch_bounds__(NTHREADS) void tconv_wgrad_kernel(const TwgParams P) {
  using E = Elem<T>;
  constexpr int EPL = E::EPL;
  typedef typename E::frag frag_t;
  constexpr int CB = 32;                         // channels per operand tile
  constexpr int QV = CB / EPL;                   // 16-byte vectors per staged row
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned short* row_f = reinterpret_cast<unsigned short*>(smem);      // [TR]
  unsigned short* row_v = row_f + TR;                                    // [TR]
  unsigned short* urow = reinterpret_cast<unsigned short*>(smem + P.off_urow);  // [TR] u-tile row at tap offset 0
  T* dzs = reinterpret_cast<T*>(smem + P.off_dz);                        // [TR][CB]
  T* us = reinterpret_cast<T*>(smem + P.off_u);                          // [Fin*V][CB]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int V = P.V;
  const int ot = blockIdx.y / P.n_itile, it = blockIdx.y - ot * P.n_itile;
  const int o0 = ot * CB, i0 = it * CB;
  const bool vec = (P.Cin % EPL == 0) && (P.Cout % EPL == 0);

  for (int r = tid; r < TR; r += NTHREADS) {
    int f = r / V;
    row_f[r] = (unsigned short)f;
    row_v[r] = (unsigned short)(r - f * V);
  }
  __syncthreads();

  f32x16 acc[JT];
#pragma unroll
  for (int j = 0; j < JT; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  float bsum = 0.f;   // dbias partial: thread (c = tid & 31, rows tid>>5 step 8)

  const T* dzg = reinterpret_cast<const T*>(P.dz);
  const T* gg = reinterpret_cast<const T*>(P.g);

  for (int tile = blockIdx.x; tile < P.total_tiles; tile += gridDim.x) {
    const int n = tile / P.tiles_per_seq;
    const int m0 = (tile - n * P.tiles_per_seq) * P.F;
    const int nf = min(P.F, P.Tz - m0);
    const int rows = nf * V;
    const int fin0 = P.in_mul * m0 + P.min_off;
    const int in_rows = (P.in_mul * (nf - 1) + P.Fin - P.in_mul * (P.F - 1)) * V;

    // ---- stage dz tile (zero pad rows) ----
    for (int idx = tid; idx < TR * QV; idx += NTHREADS) {
      const int r = idx / QV, q = idx - r * QV;
      const int c0 = o0 + q * EPL;
      frag_t val;
      zero_frag<T>(val);
      if (r < rows && c0 < P.Cout) {
        const size_t a = ((size_t)(n * P.Tz + m0 + row_f[r]) * V + row_v[r]) * P.Cout + c0;
        if (vec) val = *reinterpret_cast<const frag_t*>(dzg + a);
        else {
#pragma unroll
          for (int e = 0; e < EPL; ++e) if (c0 + e < P.Cout) val[e] = dzg[a + e];
        }
      }
      *reinterpret_cast<frag_t*>(dzs + r * CB + q * EPL) = val;
    }
    for (int r = tid; r < TR; r += NTHREADS)
      urow[r] = r < rows ? (unsigned short)((P.in_mul * row_f[r]) * V + row_v[r]) : (unsigned short)0;
    // ---- stage u tile with halo: pre(g), zero outside the sequence ----
    for (int idx = tid; idx < in_rows * QV; idx += NTHREADS) {
      const int r = idx / QV, q = idx - r * QV;
      const int fl = r / V, v = r - fl * V;
      const int fr = fin0 + fl;
      const int c0 = i0 + q * EPL;
      frag_t val;
      zero_frag<T>(val);
      if (fr >= 0 && fr < P.Tin && c0 < P.Cin) {
        const size_t a = ((size_t)(n * P.Tin + fr) * V + v) * P.Cin + c0;
        if (vec) val = *reinterpret_cast<const frag_t*>(gg + a);
        else {
#pragma unroll
          for (int e = 0; e < EPL; ++e) if (c0 + e < P.Cin) val[e] = gg[a + e];
        }
        if (P.pre) {
#pragma unroll
          for (int e = 0; e < EPL; ++e) {
            if (c0 + e < P.Cin) {
              float fv = E::to_f(val[e]) * P.pre[c0 + e] + P.pre[P.Cin + c0 + e];
              if (P.pre_relu) fv = fmaxf(fv, 0.f);
              val[e] = E::from_f(fv);
            }
          }
        }
      }
      *reinterpret_cast<frag_t*>(us + r * CB + q * EPL) = val;
    }
    __syncthreads();

    if (P.dbias && it == 0) {
      const int c = tid & 31;
      for (int r = tid >> 5; r < rows; r += 8) bsum += E::to_f(dzs[r * CB + c]);
    }

    // ---- this wave's 32 positions: D_j[o][i] += dz[p][o] * u[row(p) + tap_j][i] ----
    if constexpr (sizeof(T) == 4) {
      const int r = lane & 31, h = lane >> 5;
#pragma unroll 4
      for (int kk = 0; kk < 16; ++kk) {
        const int p = wave * 32 + 2 * kk + h;
        const float a = dzs[p * CB + r];
        const int ub = urow[p];
#pragma unroll
        for (int j = 0; j < JT; ++j) {
          if (j < P.ntaps) {
            const float b = us[(ub + (P.tap_off[j] - P.min_off) * V) * CB + r];
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
          }
        }
      }
    } else {
      const int grp = lane >> 4, h = grp >> 1, cblk = (grp & 1) * 16;
      const int q = (lane & 15) >> 2, pp = lane & 3;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int pb = wave * 32 + 16 * kk + 8 * h;       // this lane-half's 8 positions: pb .. pb+7
        // A = dz^T: rows pb+4s+q of dzs, columns cblk + 4*pp
        bf16x8 a;
        {
          typedef short s16x4 __attribute__((ext_vector_type(4)));
          s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4*)(dzs + (pb + q) * CB + cblk + 4 * pp));
          s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4*)(dzs + (pb + 4 + q) * CB + cblk + 4 * pp));
          bf16x4 l4 = __builtin_bit_cast(bf16x4, lo), h4 = __builtin_bit_cast(bf16x4, hi);
          a[0] = l4[0]; a[1] = l4[1]; a[2] = l4[2]; a[3] = l4[3];
          a[4] = h4[0]; a[5] = h4[1]; a[6] = h4[2]; a[7] = h4[3];
        }
        const int ub0 = urow[pb + q], ub1 = urow[pb + 4 + q];
#pragma unroll
        for (int j = 0; j < JT; ++j) {
          if (j < P.ntaps) {
            const int ro = (P.tap_off[j] - P.min_off) * V;
            typedef short s16x4 __attribute__((ext_vector_type(4)));
            s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (__attribute__((address_space(3))) s16x4*)(us + (ub0 + ro) * CB + cblk + 4 * pp));
            s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (__attribute__((address_space(3))) s16x4*)(us + (ub1 + ro) * CB + cblk + 4 * pp));
            bf16x4 l4 = __builtin_bit_cast(bf16x4, lo), h4 = __builtin_bit_cast(bf16x4, hi);
            bf16x8 b;
            b[0] = l4[0]; b[1] = l4[1]; b[2] = l4[2]; b[3] = l4[3];
            b[4] = h4[0]; b[5] = h4[1]; b[6] = h4[2]; b[7] = h4[3];
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
          }
        }
      }
    }
    __syncthreads();
  }

  // ---- flush: D tile rows = o (registers), cols = i (lanes): two 128-byte segments per atomic instruction ----
#pragma unroll
  for (int j = 0; j < JT; ++j) {
    if (j < P.ntaps) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int o = o0 + mfma_row(r, lane), i = i0 + (lane & 31);
        if (o < P.Cout && i < P.Cin) atomicAdd(P.dW + ((size_t)j * P.Cout + o) * P.Cin + i, acc[j][r]);
      }
    }
  }
  if (P.dbias && it == 0) {
    // reduce the 8 row-groups that share a channel: lanes l and l+32 within a wave, then across waves via atomics
    bsum += __shfl_xor(bsum, 32);
    if (lane < 32 && o0 + lane < P.Cout) atomicAdd(P.dbias + o0 + lane, bsum);
  }
}

template <typename T>
int launch_T(TwgParams& P, int grid_cap, hipStream_t stream) {
  const int esz = sizeof(T);
  int mn = P.tap_off[0], mx = P.tap_off[0];
  for (int j = 1; j < P.ntaps; ++j) { mn = P.tap_off[j] < mn ? P.tap_off[j] : mn; mx = P.tap_off[j] > mx ? P.tap_off[j] : mx; }
  P.min_off = mn;
  P.F = TR / P.V;
  P.Fin = P.in_mul * (P.F - 1) + (mx - mn) + 1;
  P.tiles_per_seq = ceil_div(P.Tz, P.F);
  P.total_tiles = P.NM * P.tiles_per_seq;
  P.n_itile = ceil_div(P.Cin, 32);
  const int n_otile = ceil_div(P.Cout, 32);
  size_t off = (size_t)2 * TR * 2;
  off = (off + 15) & ~(size_t)15; P.off_urow = (int)off; off += (size_t)TR * 2;
  off = (off + 15) & ~(size_t)15; P.off_dz = (int)off; off += (size_t)TR * 32 * esz;
  off = (off + 15) & ~(size_t)15; P.off_u = (int)off; off += (size_t)P.Fin * P.V * 32 * esz;
  if (off > 160 * 1024 || P.Fin * P.V > 65535) return ISTGCN_EINVAL;
  const int pairs = n_otile * P.n_itile;
  int gx = grid_cap / pairs;
  if (gx < 1) gx = 1;
  if (gx > P.total_tiles) gx = P.total_tiles;
  dim3 grid(gx, pairs);
#define GO(JTv)                                                                                             \
  do {                                                                                                      \
    auto kfn = tconv_wgrad_kernel<T, JTv>;                                                                  \
    static bool attr_done = false;                                                                          \
    if (!attr_done) {                                                                                       \
      hipError_t ea_ = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
      if (ea_ != hipSuccess) return 2000 + (int)ea_; \
      attr_done = true;                                                                                     \
    }                                                                                                       \
    ISTGCN_LAUNCH(kfn, grid, dim3(NTHREADS), off, stream, P);                                          \
  } while (0)
  if (P.ntaps <= 1) GO(1);
  else if (P.ntaps <= 3) GO(3);
  else if (P.ntaps <= 9) GO(9);
  else GO(15);
#undef GO
  ISTGCN_CHECK_LAUNCH();
  return ISTGCN_OK;
}

}  // namespace

extern "C" int istgcn_tconv_wgrad(const void* dz, const void* g, const float* pre, int pre_relu, float* dW,
                                  float* dbias, int NM, int Tin, int Tz, int V, int Cin, int Cout, int ntaps,
                                  const int* tap_off, int in_mul, int dtype, int grid_cap, void* stream) {
  if (!dz || !g || !dW || !tap_off) return ISTGCN_EINVAL;
  if (ntaps < 1 || ntaps > 15 || V < 1 || V > 128 || Cin < 1 || Cout < 1 || in_mul < 1 || NM < 0 || Tz < 0)
    return ISTGCN_EINVAL;
  if (dtype != 0 && dtype != 1) return ISTGCN_EINVAL;
  if (NM == 0 || Tz == 0) return ISTGCN_OK;
  TwgParams P{};
  P.dz = dz; P.g = g; P.pre = pre; P.dW = dW; P.dbias = dbias;
  P.NM = NM; P.Tin = Tin; P.Tz = Tz; P.V = V; P.Cin = Cin; P.Cout = Cout; P.ntaps = ntaps; P.in_mul = in_mul;
  P.pre_relu = pre_relu;
  for (int j = 0; j < ntaps; ++j) P.tap_off[j] = tap_off[j];
  if (grid_cap < 1) grid_cap = 1024;
  if (dtype == 0) return launch_T<float>(P, grid_cap, (hipStream_t)stream);
  return launch_T<__bf16>(P, grid_cap, (hipStream_t)stream);
}
