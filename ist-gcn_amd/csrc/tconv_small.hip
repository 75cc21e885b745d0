// Round-1 temporal-convolution kernel (two independent 4-wave workgroups per CU, every wave stages, computes and stores),
// kept for the shapes where it beats the wave-specialised kernel of tconv.hip: convolutions with so few channels that
// there is no matrix work to overlap (the sqrt(C)-wide bottleneck of net/st_gcn_mstcn_1x1*.py: 8-16 channels, one tap) are
// pure streaming, and eight loading waves per CU keep more bytes in flight than four memory waves (config 5: tconv
// 22.2 ms/step here against 26.5 there).  istgcn_tconv / istgcn_tconv_geometry (tconv.hip) dispatch by shape; the packed
// weight layout is the same family, with THIS file's chunk width for the shapes it serves.
// Temporal convolution over the frame axis of an NTVC tensor as an implicit GEMM on the matrix cores:
//
//   out[n, out_mul*m + out_off, v, o] = epi( sum_j sum_i Wf[j][o][i] * pre(in[n, in_mul*m + tap_off[j], v, i]) )
//
// for m in [0, Mlog); frames outside [0, Tin) contribute zero (the Conv2d zero padding, applied AFTER `pre`).
//   pre  = optional per-channel affine + ReLU  -> the BatchNorm2d+ReLU in front of the conv
//          (net/st_gcnold.py:165-166 `tcn.0/tcn.1`, `tcn_start` of net/st_gcn_multi3_fix_3A_mstcn.py:159-162)
//   sum  = the (k,1) Conv2d, stride (s,1): st_gcnold.py:167-173; the three-branch Inception-TCN
//          x1*m0 + x2*m1 + x3*m2 of st_gcn_multi3_fix_3A_mstcn.py:212-215 (and the /3 of st_gcn_mstcn.py:245)
//          is ONE 15-tap convolution whose taps the host pre-sums (linear in the weights).
//   epi  = + bias, and per-channel sum / sum-of-squares for the train-mode BatchNorm2d that follows
//          (st_gcnold.py:174), or -- for the data gradient -- the ReLU mask of the producer's BatchNorm+ReLU
//          recomputed from `aux`, with the two BatchNorm-backward reductions; or -- inference, every BatchNorm
//          folded into the weights by the host -- the block's tail relu(conv + residual) (st_gcnold.py:201-203)
//          so that an eval-mode st_gcn block is two launches.
// Forward: in_mul = stride, tap_off[j] = j - pad, out_mul = 1.  Data gradient: one launch per output phase
// (t mod stride) with the taps that hit that phase, in_mul = 1, out_mul = stride, out_off = phase.
//
// A workgroup (4 waves) owns TR = 128*NT output rows (whole frames of one sequence) x up to 128 output
// channels; the input rows it needs (with the tap halo) are staged per channel chunk into LDS once and then
// re-read at a row offset per tap, so the activation is fetched from HBM/L2 once per chunk, not once per tap.
// Tiles of one sequence are kept on one XCD (grid-stride order below) so halo re-reads hit that XCD's L2.
#include "common.hpp"

#ifdef ISTGCN_STAMP
// diagnostic build only: per-phase cycle sums (lane 0 of every wave), read back with istgcn_debug_stamps
__device__ unsigned long long g_stamp[8];
#define STAMP(i)                                                                                   \
  do {                                                                                             \
    unsigned long long t_ = __builtin_amdgcn_s_memtime();                                          \
    if (lane == 0) st_acc[i] += t_ - st_prev;                                                      \
    st_prev = __builtin_amdgcn_s_memtime();                                                        \
  } while (0)
#else
#define STAMP(i)
#endif

namespace {

constexpr int NTHREADS = 256;
constexpr int MAX_TAPS = 16;

struct TconvParams {
  const void* in;
  const void* Wp;
  const float* bias;     // [Cout] or null
  const float* pre;      // [2][Cin] scale, shift or null
  const void* aux;       // epilogue mode 1: [NM][Tout][V][Cout]
  const float* maux;     // epilogue mode 1: [4][Cout] scale, shift, mean, rstd
  void* out;
  double* stats;         // [stats_rep][2][Cout] or null
  int NM, Tin, Tout, Mlog, V, Cin, Cout, ntaps;
  int in_mul, out_mul, out_off, pre_relu, mode, stats_rep;
  int tap_off[MAX_TAPS];
  // derived on the host
  int F, tiles_per_seq, total_tiles, CC, nch, NKG, MTtot, min_off, Fin;
  int us_stride, out_stride, off_stat, off_work;
};

// WM = waves along the channel axis: the workgroup has 4*WM waves; wave (wr = wave & 3, wm = wave >> 2) owns row slab wr
// and the MT/WM output-channel tiles [wm*MTW, (wm+1)*MTW).  WM = 2 doubles the waves per CU at the same LDS footprint
// (these kernels wait on memory and barriers more than half of their wave-cycles).
template <typename T, int MT, int NT, bool VEC, int WM, int MODE>
__global__ __launch_bounds__(NTHREADS * WM, 2) void tconv_kernel(const TconvParams P) {
  constexpr int NTH = NTHREADS * WM;
  constexpr int MTW = MT / WM;
  static_assert(MT % WM == 0, "channel tiles must split evenly over the channel waves");
  using E = Elem<T>;
  constexpr int EPL = E::EPL;
  constexpr int KGS = E::KGS;
  constexpr int TR = 128 * NT;
  typedef typename E::frag frag_t;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned short* row_f = reinterpret_cast<unsigned short*>(smem);          // [TR]
  unsigned short* row_v = row_f + TR;                                        // [TR]
  float* stat = reinterpret_cast<float*>(smem + P.off_stat);                 // [2][MT*32]
  int* tap_roff = reinterpret_cast<int*>(stat + 2 * MT * 32);                 // [MAX_TAPS] LDS row offset per tap
  T* us = reinterpret_cast<T*>(smem + P.off_work);                           // [Fin*V][us_stride]
  T* outs = us;                                                              // [TR][out_stride]

  const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) & 3, wm = tid >> 8;
  const int V = P.V;
  const int mt0 = blockIdx.y * MT;
  const int cbase_blk = mt0 * 32;
  const int Q = P.CC / EPL;

  for (int r = tid; r < TR; r += NTH) {
    int f = r / V;
    row_f[r] = (unsigned short)f;
    row_v[r] = (unsigned short)(r - f * V);
  }
  for (int c = tid; c < 2 * MT * 32; c += NTH) stat[c] = 0.f;
  if (tid < P.ntaps) tap_roff[tid] = (P.tap_off[tid] - P.min_off) * V;
  __syncthreads();

  const T* ing = reinterpret_cast<const T*>(P.in);
  const T* Wp = reinterpret_cast<const T*>(P.Wp);
  const T* auxg = reinterpret_cast<const T*>(P.aux);
  T* outg = reinterpret_cast<T*>(P.out);

  // XCD-affine persistent order: XCD x (= blockIdx.x % 8 under round-robin dispatch; speed only) walks the
  // contiguous tile range [x*chunk, (x+1)*chunk), its workgroups taking neighbouring tiles at each step.
  const int G8 = gridDim.x >> 3;
  const int chunk = (P.total_tiles + 7) >> 3;
  const int xcd = blockIdx.x & 7;

  // BatchNorm partial sums: a thread always copies out the same channel vector.  Where registers allow (small
  // accumulator footprints) the sums stay in registers for the whole grid-stride walk and are reduced across lanes once
  // per workgroup; the register-heavy instantiations reduce per tile instead.
  constexpr bool REG_STATS = MTW * NT < 8;
  constexpr int NPASS_ = (MT + 1) / 2;
  float st1[REG_STATS ? NPASS_ : 1][EPL], st2[REG_STATS ? NPASS_ : 1][EPL];
#pragma unroll
  for (int ps = 0; ps < (REG_STATS ? NPASS_ : 1); ++ps)
#pragma unroll
    for (int jj = 0; jj < EPL; ++jj) { st1[ps][jj] = 0.f; st2[ps][jj] = 0.f; }

#ifdef ISTGCN_STAMP
  unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_prev = __builtin_amdgcn_s_memtime();
#endif
  for (int slot = blockIdx.x >> 3; slot < chunk; slot += G8) {
    const int tile = xcd * chunk + slot;
    if (tile >= P.total_tiles) break;
    STAMP(0);
    const int n = tile / P.tiles_per_seq;
    const int m0 = (tile - n * P.tiles_per_seq) * P.F;
    const int nf = min(P.F, P.Mlog - m0);
    const int rows = nf * V;
    const int fin0 = P.in_mul * m0 + P.min_off;          // first staged input frame (may be < 0)
    const int in_rows = (P.in_mul * (nf - 1) + P.Fin - P.in_mul * (P.F - 1)) * V;   // frames actually needed

    // accumulators start at the conv bias (rows of the D tile = output channels: 4 consecutive ones per register
    // quad), so the epilogue has no bias pass
    f32x16 acc[MTW][NT];
#pragma unroll
    for (int m = 0; m < MTW; ++m) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (P.bias) {
          const int cg = cbase_blk + (wm * MTW + m) * 32 + 8 * g + 4 * (lane >> 5);
          if (VEC && cg + 3 < P.Cout) {
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(P.bias + cg);
            bv[0] = b4[0]; bv[1] = b4[1]; bv[2] = b4[2]; bv[3] = b4[3];
          } else {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) if (cg + jj < P.Cout) bv[jj] = P.bias[cg + jj];
          }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) acc[m][t][4 * g + jj] = bv[jj];
      }
    }

    // per-lane LDS row of its output rows at tap offset 0 (pad rows clamp to row 0: computed, never stored)
    int brow[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int p = wave * (32 * NT) + t * 32 + (lane & 31);
      brow[t] = p < rows ? (P.in_mul * row_f[p]) * V + row_v[p] : 0;
    }

    for (int ch = 0; ch < P.nch; ++ch) {
      const int cb = ch * P.CC;
      // ---- stage the input rows of this chunk (BatchNorm affine + ReLU applied on the way in) ----
      {
        const long long row0 = (long long)(n * P.Tin + fin0) * V;
        const int r_lo = fin0 < 0 ? -fin0 * V : 0;
        const int r_hi = min(in_rows, (P.Tin - fin0) * V);
        stage_block<T, (MTW * NT >= 8 ? 4 : 8), VEC>(ing + row0 * P.Cin + cb, (size_t)P.Cin, P.Cin - cb, us, P.us_stride, in_rows, r_lo, r_hi,
                               Q, P.pre ? P.pre + cb : nullptr, P.pre ? P.pre + P.Cin + cb : nullptr, P.pre_relu, tid,
                               NTH);
      }
      STAMP(1);
      __syncthreads();
      STAMP(2);
      // ---- taps x k-groups on the matrix cores, software pipelined: the weight fragments (L2) and the shifted
      //      activation fragments (LDS) of step it+1 are in flight while the MFMAs of step it issue ----
      {
        const int nit = P.ntaps * P.NKG;                       // NKG is a power of two
        const int lkg = 31 - __builtin_clz(P.NKG);
        const T* wbase = Wp + (((size_t)ch * nit) * P.MTtot + mt0) * 64 * EPL + lane * EPL;
        const int hoff = (lane >> 5) * EPL;
        const int roff0 = (P.tap_off[0] - P.min_off) * V;
        const int rstep = P.ntaps > 1 ? (P.tap_off[1] - P.tap_off[0]) * V : 0;
        auto load_step = [&](int it, frag_t (&a)[MTW], frag_t (&b)[NT]) {
          const int j = it >> lkg, kg = it & (P.NKG - 1);
          const int roff = roff0 + j * rstep;              // taps are an arithmetic progression (checked on the host)
#pragma unroll
          for (int m = 0; m < MTW; ++m)
            a[m] = *reinterpret_cast<const frag_t*>(wbase + ((size_t)it * P.MTtot + wm * MTW + m) * 64 * EPL);
#pragma unroll
          for (int t = 0; t < NT; ++t)
            b[t] = *reinterpret_cast<const frag_t*>(us + (brow[t] + roff) * P.us_stride + kg * KGS + hoff);
        };
        auto mma_step = [&](const frag_t (&a)[MTW], const frag_t (&b)[NT]) {
#pragma unroll
          for (int m = 0; m < MTW; ++m)
#pragma unroll
            for (int t = 0; t < NT; ++t) mma_kgroup(acc[m][t], a[m], b[t]);
        };
        // ring depth: fp32 steps are 4x longer (4 MFMAs of 64 cycles per k-group), two slots cover L2; bf16 needs more
        constexpr int DEPTH = sizeof(T) == 4 ? 2 : (MTW * NT <= 4 ? 4 : 3);
        mfma_ring<DEPTH, MTW, NT, frag_t>(nit, load_step, mma_step);
      }
      STAMP(3);
      __syncthreads();
      STAMP(4);
    }

    // ---- epilogue: one pass per 64-channel pair over ALL TR rows of the tile (in-kernel stamps showed this stage at
    //      35-50 % of the bf16 kernel when it made one pass per 32-row slab): accumulators -> LDS (row-major, channels
    //      innermost) -> coalesced 16-byte stores with the mask / BatchNorm sums applied on the way out ----
    constexpr int NPASS = (MT + 1) / 2;
    constexpr int VPR = 64 / EPL;
    constexpr int RSTEP = NTH / VPR;
    const bool dense_rows = P.out_mul == 1;        // the tile's output rows are then one contiguous run in HBM
    const size_t out_base = ((size_t)(n * P.Tout + m0 * P.out_mul + P.out_off) * V) * P.Cout;
    // per-channel constants of this thread's channel vector (fixed across passes up to the 64-channel offset)
    const int vq = tid % VPR;
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int sr = wave * (32 * NT) + t * 32 + (lane & 31);       // staging row = tile row
#pragma unroll
        for (int ml = 0; ml < 2; ++ml) {
          const int mg = 2 * ps + ml;                  // channel tile of this pass; held by channel-wave mg / MTW
          const int m = mg % MTW;
          if (mg < MT && mg / MTW == wm) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const int cl = ml * 32 + 8 * g + 4 * (lane >> 5);
              float v4[4] = {acc[m][t][4 * g], acc[m][t][4 * g + 1], acc[m][t][4 * g + 2], acc[m][t][4 * g + 3]};
              store4(outs + sr * P.out_stride + cl, v4);
            }
          }
        }
      }
      __syncthreads();
      {
        const int cg = cbase_blk + ps * 64 + vq * EPL;
        const bool col_live = (ps * 64 + vq * EPL) < MT * 32 && cg < P.Cout;
        float s1[EPL], s2[EPL], msc[EPL], msh[EPL], mmu[EPL], mrs[EPL];
#pragma unroll
        for (int jj = 0; jj < EPL; ++jj) { s1[jj] = 0.f; s2[jj] = 0.f; msc[jj] = 0.f; msh[jj] = 0.f; mmu[jj] = 0.f; mrs[jj] = 0.f; }
        if constexpr (MODE == 2) {
          // residual affine (the folded BatchNorm of the strided 1x1 residual conv) or identity
#pragma unroll
          for (int jj = 0; jj < EPL; ++jj) msc[jj] = 1.f;
          if (P.maux && col_live) {
#pragma unroll
            for (int jj = 0; jj < EPL; ++jj) {
              if (cg + jj < P.Cout) { msc[jj] = P.maux[cg + jj]; msh[jj] = P.maux[P.Cout + cg + jj]; }
            }
          }
        }
        if constexpr (MODE == 1) {
          // producer's BatchNorm constants of this thread's channel vector: whole 16-byte loads, no per-element branches
          if (col_live) {
            if (VEC) {
#pragma unroll
              for (int j4 = 0; j4 < EPL; j4 += 4) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(P.maux + cg + j4);
                const f32x4 b = *reinterpret_cast<const f32x4*>(P.maux + P.Cout + cg + j4);
                const f32x4 c = *reinterpret_cast<const f32x4*>(P.maux + 2 * P.Cout + cg + j4);
                const f32x4 d = *reinterpret_cast<const f32x4*>(P.maux + 3 * P.Cout + cg + j4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { msc[j4 + e] = a[e]; msh[j4 + e] = b[e]; mmu[j4 + e] = c[e]; mrs[j4 + e] = d[e]; }
              }
            } else {
#pragma unroll
              for (int jj = 0; jj < EPL; ++jj) {
                if (cg + jj < P.Cout) {
                  msc[jj] = P.maux[cg + jj]; msh[jj] = P.maux[P.Cout + cg + jj];
                  mmu[jj] = P.maux[2 * P.Cout + cg + jj]; mrs[jj] = P.maux[3 * P.Cout + cg + jj];
                }
              }
            }
          }
        }
        if (col_live) {
          // UB rows per batch: their LDS reads and (data gradient) aux loads are all issued before the first is used
          constexpr int UB = 4;
          for (int p0 = tid / VPR; p0 < rows; p0 += RSTEP * UB) {
            frag_t sv[UB], av[UB];
            size_t g[UB];
            bool ok[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
              const int p = p0 + u * RSTEP;
              ok[u] = p < rows;
              const int pc = ok[u] ? p : p0;
              if (dense_rows) g[u] = out_base + (size_t)pc * P.Cout + cg;
              else g[u] = ((size_t)(n * P.Tout + (m0 + row_f[pc]) * P.out_mul + P.out_off) * V + row_v[pc]) * P.Cout + cg;
              sv[u] = *reinterpret_cast<const frag_t*>(outs + pc * P.out_stride + vq * EPL);
              if constexpr (MODE >= 1) {
                if (MODE == 1 || auxg) {
                  if (VEC) av[u] = *reinterpret_cast<const frag_t*>(auxg + g[u]);
                  else {
#pragma unroll
                    for (int jj = 0; jj < EPL; ++jj) av[u][jj] = (cg + jj < P.Cout) ? auxg[g[u] + jj] : E::from_f(0.f);
                  }
                } else {
                  zero_frag<T>(av[u]);
                }
              }
            }
#pragma unroll
            for (int u = 0; u < UB; ++u) {
              if (!ok[u]) continue;
              if constexpr (MODE == 1) {
#pragma unroll
                for (int jj = 0; jj < EPL; ++jj) {
                  if (VEC || cg + jj < P.Cout) {
                    const float xa = E::to_f(av[u][jj]);
                    const T o = E::from_f(xa * msc[jj] + msh[jj] > 0.f ? E::to_f(sv[u][jj]) : 0.f);
                    sv[u][jj] = o;
                    const float fv = E::to_f(o);
                    s1[jj] += fv;
                    s2[jj] += fv * (xa - mmu[jj]) * mrs[jj];
                  }
                }
              } else if constexpr (MODE == 2) {
#pragma unroll
                for (int jj = 0; jj < EPL; ++jj) {
                  if (VEC || cg + jj < P.Cout) {
                    const float r = auxg ? E::to_f(av[u][jj]) * msc[jj] + msh[jj] : 0.f;
                    sv[u][jj] = E::from_f(fmaxf(E::to_f(sv[u][jj]) + r, 0.f));
                  }
                }
              } else {
#pragma unroll
                for (int jj = 0; jj < EPL; ++jj) {
                  if (VEC || cg + jj < P.Cout) {
                    const float fv = E::to_f(sv[u][jj]);
                    s1[jj] += fv;
                    s2[jj] += fv * fv;
                  }
                }
              }
              if (VEC) *reinterpret_cast<frag_t*>(outg + g[u]) = sv[u];
              else {
#pragma unroll
                for (int jj = 0; jj < EPL; ++jj) if (cg + jj < P.Cout) outg[g[u] + jj] = sv[u][jj];
              }
            }
          }
        }
        if constexpr (REG_STATS) {
#pragma unroll
          for (int jj = 0; jj < EPL; ++jj) { st1[ps][jj] += s1[jj]; st2[ps][jj] += s2[jj]; }
        } else if (P.stats) {
#pragma unroll
          for (int jj = 0; jj < EPL; ++jj) {
#pragma unroll
            for (int msk = VPR; msk < 64; msk <<= 1) {
              s1[jj] += __shfl_xor(s1[jj], msk);
              s2[jj] += __shfl_xor(s2[jj], msk);
            }
          }
          if (lane < VPR && col_live) {
#pragma unroll
            for (int jj = 0; jj < EPL; ++jj) {
              const int cl = ps * 64 + vq * EPL + jj;
              if (cbase_blk + cl < P.Cout) {
                atomicAdd(&stat[cl], s1[jj]);
                atomicAdd(&stat[MT * 32 + cl], s2[jj]);
              }
            }
          }
        }
      }
      __syncthreads();
    }
    STAMP(5);
  }
#ifdef ISTGCN_STAMP
  if (lane == 0)
    for (int i = 0; i < 8; ++i) atomicAdd(&g_stamp[i], st_acc[i]);
#endif

  if (P.stats) {
    if constexpr (REG_STATS) {
      constexpr int VPR = 64 / EPL;
      const int vq = tid % VPR;
#pragma unroll
      for (int ps = 0; ps < NPASS_; ++ps) {
#pragma unroll
        for (int jj = 0; jj < EPL; ++jj) {
          float a = st1[ps][jj], b = st2[ps][jj];
#pragma unroll
          for (int msk = VPR; msk < 64; msk <<= 1) { a += __shfl_xor(a, msk); b += __shfl_xor(b, msk); }
          const int cl = ps * 64 + vq * EPL + jj;
          if (lane < VPR && cl < MT * 32 && cbase_blk + cl < P.Cout) {
            atomicAdd(&stat[cl], a);
            atomicAdd(&stat[MT * 32 + cl], b);
          }
        }
      }
    }
    __syncthreads();
    double* dst = P.stats + (size_t)(blockIdx.x % P.stats_rep) * 2 * P.Cout;
    for (int c = tid; c < MT * 32; c += NTH) {
      if (cbase_blk + c < P.Cout) {
        atomic_add_f64(dst + cbase_blk + c, (double)stat[c]);
        atomic_add_f64(dst + P.Cout + cbase_blk + c, (double)stat[MT * 32 + c]);
      }
    }
  }
}

template <typename T, int MT, int NT>
int launch3(const TconvParams& P, int grid_cap, int gy, size_t lds, hipStream_t stream) {
  constexpr int WM = 1;   // WM = 2 (8 waves) needs <= 128 VGPRs for two workgroups per CU; the staging/epilogue code does not fit yet
  const bool vec = (P.Cin % Elem<T>::EPL) == 0 && (P.Cout % Elem<T>::EPL) == 0;
#define GO(VV, MD)                                                                                           \
  do {                                                                                                      \
    auto kfn = tconv_kernel<T, MT, NT, VV, WM, MD>;                                                               \
    static std::atomic<unsigned long long> optin{0};                                                        \
    if (int ea_ = istgcn_lds_optin((const void*)kfn, optin)) return ea_;                                    \
    int gx = (grid_cap > 0 ? grid_cap : istgcn_resident_blocks((const void*)kfn, NTHREADS * WM, lds)) / gy; \
    gx = round_up(gx < 1 ? 1 : (gx > P.total_tiles ? P.total_tiles : gx), 8);  /* XCD-affine order: multiple of 8 */ \
    ISTGCN_LAUNCH(kfn, dim3(gx, gy), dim3(NTHREADS * WM), lds, stream, P);                                  \
  } while (0)
  if (P.mode == 1) { if (vec) GO(true, 1); else GO(false, 1); }
  else if (P.mode == 2) { if (vec) GO(true, 2); else GO(false, 2); }
  else { if (vec) GO(true, 0); else GO(false, 0); }
#undef GO
  ISTGCN_CHECK_LAUNCH();
  return ISTGCN_OK;
}

// Tiling decision shared by the launcher and the geometry query (the host packs weights to match).
struct TconvGeom { int CC, nch, NKG, MT, MTtot, gy, NT, F, Fin, min_off, lds, off_stat, off_work, us_stride, out_stride; };

inline int tconv_geom(int V, int Cin, int Cout, int ntaps, const int* tap_off, int in_mul, int dtype, TconvGeom* G) {
  const int epl = dtype == 0 ? 4 : 8, kgs = 2 * epl, esz = dtype == 0 ? 4 : 2;
  int mn = tap_off[0], mx = tap_off[0];
  for (int j = 1; j < ntaps; ++j) { mn = tap_off[j] < mn ? tap_off[j] : mn; mx = tap_off[j] > mx ? tap_off[j] : mx; }
  G->min_off = mn;
  G->MT = Cout <= 32 ? 1 : Cout <= 64 ? 2 : 4;
  G->gy = ceil_div(Cout, G->MT * 32);
  G->MTtot = G->gy * G->MT;
  G->out_stride = 64 + epl;
  const int budget = 78 * 1024;                       // two workgroups per CU
  const int cc_max = dtype == 0 ? 32 : 64;
  int best_nt = 0, best_cc = 0;
  for (int nt = 2; nt >= 1 && !best_nt; --nt) {
    if (nt * 128 / V < 1) continue;
    for (int cc = cc_max; cc >= kgs; cc >>= 1) {
      const int F = nt * 128 / V;
      const int Fin = in_mul * (F - 1) + (mx - mn) + 1;
      const long need = (long)Fin * V * (cc + epl) * esz;
      if (need <= budget) { best_nt = nt; best_cc = cc; break; }
    }
  }
  if (!best_nt) { best_nt = 1; best_cc = kgs; }
  int cc = best_cc;
  if (Cin < cc) cc = round_up(Cin, kgs);
  G->NT = best_nt; G->CC = cc; G->nch = ceil_div(Cin, cc); G->NKG = cc / kgs;
  G->F = best_nt * 128 / V;
  G->Fin = in_mul * (G->F - 1) + (mx - mn) + 1;
  G->us_stride = cc + epl;
  size_t off = (size_t)2 * 128 * best_nt * sizeof(unsigned short);
  off = (off + 15) & ~(size_t)15; G->off_stat = (int)off; off += (size_t)2 * G->MT * 32 * 4 + MAX_TAPS * 4;
  off = (off + 15) & ~(size_t)15; G->off_work = (int)off;
  size_t work = (size_t)G->Fin * V * G->us_stride * esz;
  size_t ost = (size_t)128 * best_nt * G->out_stride * esz;
  off += work > ost ? work : ost;
  G->lds = (int)off;
  return off <= 160 * 1024 ? ISTGCN_OK : ISTGCN_EINVAL;
}

template <typename T>
int launch_T(TconvParams& P, const TconvGeom& G, int grid_cap, hipStream_t stream) {
  P.F = G.F; P.CC = G.CC; P.nch = G.nch; P.NKG = G.NKG; P.MTtot = G.MTtot; P.min_off = G.min_off; P.Fin = G.Fin;
  P.us_stride = G.us_stride; P.out_stride = G.out_stride; P.off_stat = G.off_stat; P.off_work = G.off_work;
  P.tiles_per_seq = ceil_div(P.Mlog, P.F);
  P.total_tiles = P.NM * P.tiles_per_seq;
  const size_t lds = G.lds;
#define CASE(MTv, NTv) if (G.MT == MTv && G.NT == NTv) return launch3<T, MTv, NTv>(P, grid_cap, G.gy, lds, stream)
  CASE(1, 1); CASE(2, 1); CASE(4, 1); CASE(1, 2); CASE(2, 2); CASE(4, 2);
#undef CASE
  return ISTGCN_EINVAL;
}

}  // namespace

extern "C" int istgcn_tconv_v1_geometry(int V, int Cin, int Cout, int ntaps, const int* tap_off, int in_mul, int dtype,
                                     int* CC, int* nch, int* MTtot, int* EPL) {
  if (!istgcn_dtype_ok(dtype)) return ISTGCN_EINVAL;
  if (!tap_off || ntaps < 1 || ntaps > MAX_TAPS || V < 1 || V > 128 || Cin < 1 || Cout < 1 || in_mul < 1) return ISTGCN_EINVAL;
  TconvGeom G;
  int rc = tconv_geom(V, Cin, Cout, ntaps, tap_off, in_mul, dtype, &G);
  if (rc) return rc;
  *CC = G.CC; *nch = G.nch; *MTtot = G.MTtot; *EPL = dtype == 0 ? 4 : 8;
  return ISTGCN_OK;
}

extern "C" int istgcn_tconv_v1(const void* in, const void* Wp, const float* bias, const float* pre, int pre_relu,
                            const void* aux, const float* maux, void* out, double* stats, int stats_rep, int mode,
                            int NM, int Tin, int Tout, int Mlog, int V, int Cin, int Cout, int ntaps,
                            const int* tap_off, int in_mul, int out_mul, int out_off, int dtype, int grid_cap,
                            void* stream) {
  if (!in || !Wp || !out || !tap_off) return ISTGCN_EINVAL;
  if (ntaps < 1 || ntaps > MAX_TAPS || V < 1 || V > 128 || Cin < 1 || Cout < 1 || in_mul < 1 || out_mul < 1) return ISTGCN_EINVAL;
  if (NM < 0 || Mlog < 0 || out_off < 0 || mode < 0 || mode > 2) return ISTGCN_EINVAL;
  if (mode == 2 && stats) return ISTGCN_EINVAL;
  if (mode == 1 && (!aux || !maux)) return ISTGCN_EINVAL;
  if (Mlog > 0 && (Mlog - 1) * out_mul + out_off >= Tout) return ISTGCN_EINVAL;
  if (stats && stats_rep < 1) return ISTGCN_EINVAL;
  if (!istgcn_dtype_ok(dtype)) return ISTGCN_EINVAL;
  for (int j = 2; j < ntaps; ++j)      // taps must be equally spaced (every forward / data-gradient phase of a conv is)
    if (tap_off[j] - tap_off[j - 1] != tap_off[1] - tap_off[0]) return ISTGCN_EINVAL;
  if (NM == 0 || Mlog == 0) return ISTGCN_OK;
  TconvParams P{};
  P.in = in; P.Wp = Wp; P.bias = bias; P.pre = pre; P.aux = aux; P.maux = maux; P.out = out; P.stats = stats;
  P.NM = NM; P.Tin = Tin; P.Tout = Tout; P.Mlog = Mlog; P.V = V; P.Cin = Cin; P.Cout = Cout; P.ntaps = ntaps;
  P.in_mul = in_mul; P.out_mul = out_mul; P.out_off = out_off; P.pre_relu = pre_relu; P.mode = mode;
  P.stats_rep = stats_rep < 1 ? 1 : stats_rep;
  for (int j = 0; j < ntaps; ++j) P.tap_off[j] = tap_off[j];
  TconvGeom G;
  int rc = tconv_geom(V, Cin, Cout, ntaps, tap_off, in_mul, dtype, &G);
  if (rc) return rc;
  if (dtype == 0) return launch_T<float>(P, G, grid_cap, (hipStream_t)stream);
  if (dtype == 2) return launch_T<_Float16>(P, G, grid_cap, (hipStream_t)stream);
  return launch_T<__bf16>(P, G, grid_cap, (hipStream_t)stream);
}

