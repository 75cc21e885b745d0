// Data gradient and adjacency gradient of the graph-convolution unit in ONE pass (autograd of net/utils/tgcn.py:79-86
// and the folded variants):
//     dxa_k[(t,w)][i] = sum_c W[k][c][i] * dy[(t,w)][c]                       MFMA 32x32, contraction over Cout
//     dx[(t,v)][i]    = sum_k sum_w A[k][v][w] * dxa_k[(t,w)][i]  (+ addend)  sparse LDS pass (rows of A)
//     dA[k][v][w]    += sum_{t,i} x[(t,v)][i] * dxa_k[(t,w)][i]               only where A[k][v][w] != 0
// This is the forward kernel run "GEMM-first": the K*Cin-wide intermediate dxa never leaves LDS, and because it is the
// very tile the adjacency gradient needs, dA costs a handful of LDS dot products instead of a second GEMM (every
// importance gradient upstream is A (.) dA: st_gcnold.py:86, tgcn_multi3_fix_3A.py:86-88, st_gcn_msgcn.py:116-117).
// `addend` carries the identity-residual gradient of the st_gcn block (st_gcnold.py:181-182,201), so the block's input
// gradient is written once.
//
// One workgroup (4 waves) owns 128 rows (whole frames) of one sequence; wave w owns rows [32w, 32w+32) as MFMA columns,
// the (k,i) outputs of the current input-channel chunk are the MFMA rows.  Weight fragments stream from L2 one k-group
// ahead of the MFMAs that consume them.
#include "common.hpp"
namespace {

constexpr int NTHREADS = 256;
constexpr int TR = 128;

struct GbdParams {
  const void* dy;        // [NM][T][V][Cout]
  const void* x;         // [NM][T][V][Cin] or null
  const float* A;        // [K][V][V]
  const float* pat;      // [K][V][V] or null: entries != 0 are the sparsity pattern dA is computed on (null: A itself)
  const void* Wb;        // fragments, see istgcn.h
  const void* addend;    // [NM][T][V][Cin] or null (may alias dx)
  void* dx;              // [NM][T][V][Cin]
  float* dA;             // [K][V][V] accumulated, or null
  int NM, T, V, Cin, Cout, K, nnz_cap;
  int F, tiles_per_seq, total_tiles, CCi, nchi, CCc, nchc, NKGc, ds_stride;
  int off_rv, off_rkw, off_ra, off_dacc, off_rows, off_afrag, off_dys, off_dxa;
};

template <typename T, int MTK, bool VEC, bool AGGM>
__global__ __launch_bounds__(NTHREADS, 2) void gcn_bwd_kernel(const GbdParams P) {
  using E = Elem<T>;
  constexpr int EPL = E::EPL;
  constexpr int KGS = E::KGS;
  typedef typename E::frag frag_t;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  int* r_off = reinterpret_cast<int*>(smem);                                         // [V+1] row lists of A
  unsigned char* r_v = smem + P.off_rv;                                              // [cap] v of entry
  unsigned short* r_kw = reinterpret_cast<unsigned short*>(smem + P.off_rkw);        // [cap] k*V + w
  float* r_a = reinterpret_cast<float*>(smem + P.off_ra);                            // [cap]
  int* r_ofs = reinterpret_cast<int*>(smem + P.off_dacc);                            // [cap] (k*NCH*TR + w)*EPL: dxa offset of entry (chunk 0, frame 0)
  unsigned char* row_f = smem + P.off_rows;                                          // [TR]
  unsigned char* row_v = row_f + TR;                                                 // [TR]
  T* dys = reinterpret_cast<T*>(smem + P.off_dys);                                   // [TR][ds_stride]  (later: x chunk)
  T* dxa = reinterpret_cast<T*>(smem + P.off_dxa);                                   // [K][NCH][TR][EPL] (+32 zero slots)
  T* afrag = reinterpret_cast<T*>(smem + P.off_afrag);                               // bf16: [K][2][64][8] fragments of A_k^T

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int V = P.V, K = P.K, CCi = P.CCi, DS = P.ds_stride;
  // dxa lives CHUNK-MAJOR in LDS: [k][16-byte channel chunk][row][EPL].  Row-major 128-byte rows put the 32 lanes of an
  // accumulator store (consecutive rows, same channels) on 4 banks -- a 16-way conflict that was 45 % of the bf16 kernel;
  // here consecutive rows are consecutive 16-byte slots.  element (k, row, col) -> ((k*NCH + col/EPL)*TR + row)*EPL + col%EPL
  const int NCH = CCi / EPL;
  const int Qi = CCi / EPL;
  const bool cci_pow2 = (CCi & (CCi - 1)) == 0;
  const int cci_lg = 31 - __builtin_clz(CCi);

  // ---- adjacency -> LDS (coalesced), then per-ROW compressed lists: row v -> entries (k, w, a) ----
  {
    // lists are built on the PATTERN (entries whose gradient is wanted: the constant adjacency B of A_eff = B (.) imp,
    // which stays fixed when an importance value passes through zero); the values come from A
    float* A_l = reinterpret_cast<float*>(dxa);
    const float* patg = P.pat ? P.pat : P.A;
    for (int i = tid; i < K * V * V; i += NTHREADS) A_l[i] = patg[i];
    // the VALUES too, into the (not yet used) dy staging region when they fit: the list fill below reads one value per
    // pattern entry inside a data-dependent loop, and from global memory that was a chain of dependent round trips per
    // row (the same setup cost 30 us per launch in gcn_fwd before it was moved to an LDS copy)
    float* Av_l = reinterpret_cast<float*>(dys);
    const bool av = (size_t)K * V * V * sizeof(float) <= (size_t)TR * DS * sizeof(T);
    if (av) for (int i = tid; i < K * V * V; i += NTHREADS) Av_l[i] = P.A[i];
    for (int r = tid; r < TR; r += NTHREADS) {
      int f = r / V;
      row_f[r] = (unsigned char)f;
      row_v[r] = (unsigned char)(r - f * V);
    }
    __syncthreads();
    if (tid < V) {
      int cnt = 0;
      for (int k = 0; k < K; ++k)
        for (int w = 0; w < V; ++w) cnt += (A_l[(k * V + tid) * V + w] != 0.f);
      r_off[tid + 1] = cnt;
    }
    if (tid == 0) r_off[0] = 0;
    __syncthreads();
    if (tid == 0) {
      int run = 0;
      for (int v = 0; v < V; ++v) { int nn = r_off[v + 1]; r_off[v] = run; run += nn; }
      r_off[V] = run;
    }
    __syncthreads();
    if (tid < V) {
      int e = r_off[tid];
      for (int k = 0; k < K; ++k)
        for (int w = 0; w < V; ++w) {
          if (A_l[(k * V + tid) * V + w] != 0.f) {
            if (e < P.nnz_cap) {
              r_v[e] = (unsigned char)tid; r_kw[e] = (unsigned short)(k * V + w);
              // (two loads behind a uniform branch, not one load through a selected pointer: that would be a flat load)
              float aval;
              if (av) aval = Av_l[(k * V + tid) * V + w]; else aval = P.A[(k * V + tid) * V + w];
              r_a[e] = aval;
              r_ofs[e] = (k * NCH * TR + w) * EPL;
            }
            ++e;
          }
        }
    }
    __syncthreads();
  }
  // AGGM (bf16 and V <= 32, chosen by the launcher): aggregation and adjacency gradient on the matrix cores
  constexpr bool mfma_agg = AGGM;
  // adjacency gradient as an MFMA product (wave k owns dA_k): measured SLOWER than the per-entry dot products (a chain
  // of 20 dependent MFMAs on three waves vs 135 independent threads); kept for reference, off
  constexpr bool DAM = false;
  static_assert(!AGGM || sizeof(T) == 2, "MFMA aggregation is the bf16 path");
  if constexpr (mfma_agg) {
    // B-operand fragments of A_k^T for the transposed aggregation on the matrix cores: lane (v = lane&31, h), k-step s,
    // element j holds A[k][v][w = 16s + 8h + j]; and 32 zero rows behind the last dxa image (the last frame's k-range)
    for (int idx = tid; idx < K * 2 * 64; idx += NTHREADS) {
      const int ln = idx & 63, sstep = (idx >> 6) & 1, k = idx >> 7;
      const int v = ln & 31, h = ln >> 5;
      frag_t fr;
#pragma unroll
      for (int j = 0; j < EPL; ++j) {
        const int w = 16 * sstep + 8 * h + j;
        fr[j] = E::from_f((v < V && w < V) ? P.A[(k * V + v) * V + w] : 0.f);
      }
      *reinterpret_cast<frag_t*>(afrag + idx * EPL) = fr;
    }
    for (int idx = tid; idx < 32; idx += NTHREADS) {     // the last frame's k-range runs 32 - V rows past the last chunk
      frag_t z;
      zero_frag<T>(z);
      *reinterpret_cast<frag_t*>(dxa + ((size_t)K * NCH * TR + idx) * EPL) = z;
    }
  }
  const int nnz = min(r_off[V], P.nnz_cap);
  constexpr int NPE = 16;                    // entries per thread: 16 * 256 >= K*V*V (checked on the host)
  float dsum[DAM ? 1 : NPE];
#pragma unroll
  for (int pe = 0; pe < (DAM ? 1 : NPE); ++pe) dsum[pe] = 0.f;
  // bf16: the adjacency gradient is a [V x C] x [C x V] product per (frame, k) -- wave k keeps the dense 32x32 tile
  // dA_k[v][w] in 16 accumulator registers across the whole walk (instead of 16 per-entry sums) and the operands are
  // plain 16-byte row reads of the x tile and of the chunk-major dxa image
  f32x16 dAacc;
#pragma unroll
  for (int r = 0; r < 16; ++r) dAacc[r] = 0.f;

  const T* dyg = reinterpret_cast<const T*>(P.dy);
  const T* xg = reinterpret_cast<const T*>(P.x);
  const T* Wb = reinterpret_cast<const T*>(P.Wb);
  const T* addg = reinterpret_cast<const T*>(P.addend);
  T* dxg = reinterpret_cast<T*>(P.dx);

  // The GEMM's weight fragments of one (input chunk, output chunk) step live in REGISTERS: with an even tile count the
  // four waves split the (k,i) rows in two halves and the positions in two 64-row halves, so a wave needs only
  // NKGc * MTK/2 <= 12 fragments per step and every fragment feeds two MFMAs.  They are fetched at the top of each step,
  // in front of the dy chunk (one memory round trip for both).  (Streaming them per k-group from L2, one fragment per
  // MFMA, left the matrix cores waiting on L2 latency: ~30 % of the kernel.)
  constexpr bool MSPLIT = (MTK % 2) == 0;
  constexpr int MH = MSPLIT ? MTK / 2 : MTK;         // (k,i) tiles per wave
  constexpr int NTW = MSPLIT ? 2 : 1;                // 32-row position tiles per wave
  const int ph = MSPLIT ? (wave & 1) : wave, mh = MSPLIT ? (wave >> 1) : 0;
  frag_t wr[4][MH];
  auto load_weights = [&](int step) {
    const T* wfrag = Wb + ((size_t)step * P.NKGc * MTK * 64 + lane) * EPL;
#pragma unroll
    for (int kg = 0; kg < 4; ++kg)
#pragma unroll
      for (int m = 0; m < MH; ++m)
        if (kg < P.NKGc) wr[kg][m] = *reinterpret_cast<const frag_t*>(wfrag + ((size_t)kg * MTK + mh * MH + m) * 64 * EPL);
  };

  for (int tile = blockIdx.x; tile < P.total_tiles; tile += gridDim.x) {
    const int n = tile / P.tiles_per_seq;
    const int t0 = (tile - n * P.tiles_per_seq) * P.F;
    const int nf = min(P.F, P.T - t0);
    const int rows = nf * V;
    const size_t pos0 = (size_t)(n * P.T + t0) * V;

    for (int ich = 0; ich < P.nchi; ++ich) {
      const int ib = ich * CCi;
      f32x16 acc[MH][NTW];
#pragma unroll
      for (int m = 0; m < MH; ++m)
#pragma unroll
        for (int t = 0; t < NTW; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[m][t][r] = 0.f;

      for (int cch = 0; cch < P.nchc; ++cch) {
        const int cb = cch * P.CCc;
        // this step's weight fragments first (L2), the dy chunk right behind them: one memory round trip for both
        load_weights(ich * P.nchc + cch);
        // (dy rows are whole 16-byte vectors whenever Cout is a multiple of the vector width, whatever Cin is: the 3-channel
        //  first layer staged its 64-channel dy element by element under the kernel-wide VEC flag)
        if (VEC || (P.Cout % EPL) == 0)
          stage_block<T, 4, true>(dyg + pos0 * P.Cout + cb, (size_t)P.Cout, P.Cout - cb, dys, DS, TR, 0, rows, P.CCc / EPL,
                                  nullptr, nullptr, 0, tid, NTHREADS);
        else
          stage_block<T, 4, false>(dyg + pos0 * P.Cout + cb, (size_t)P.Cout, P.Cout - cb, dys, DS, TR, 0, rows, P.CCc / EPL,
                                   nullptr, nullptr, 0, tid, NTHREADS);
        __syncthreads();
        {
          const T* brow = dys + (ph * 32 * NTW + (lane & 31)) * DS + (lane >> 5) * EPL;
#pragma unroll
          for (int kg = 0; kg < 4; ++kg) {
            if (kg < P.NKGc) {
              frag_t bf[NTW];
#pragma unroll
              for (int t = 0; t < NTW; ++t) bf[t] = *reinterpret_cast<const frag_t*>(brow + t * 32 * DS + kg * KGS);
#pragma unroll
              for (int m = 0; m < MH; ++m)
#pragma unroll
                for (int t = 0; t < NTW; ++t) mma_kgroup(acc[m][t], wr[kg][m], bf[t]);
            }
          }
        }
        __syncthreads();
      }

      // ---- dxa chunk -> LDS (chunk-major); x chunk -> the (now free) dy buffer for the adjacency gradient ----
      {
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
          const int p = ph * 32 * NTW + t * 32 + (lane & 31);
#pragma unroll
          for (int m = 0; m < MH; ++m) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const int kk = (mh * MH + m) * 32 + 8 * g + 4 * (lane >> 5);
              if (kk < K * CCi) {
                // (runtime integer division costs ~40 instructions; CCi is a power of two except for odd tiny Cin)
                const int k = cci_pow2 ? (kk >> cci_lg) : kk / CCi, il = kk - k * CCi;
                float v4[4] = {acc[m][t][4 * g], acc[m][t][4 * g + 1], acc[m][t][4 * g + 2], acc[m][t][4 * g + 3]};
                store4(dxa + ((k * NCH + il / EPL) * TR + p) * EPL + il % EPL, v4);
              }
            }
          }
        }
      }
      if (P.dA)
        stage_block<T, 4, VEC>(xg + pos0 * P.Cin + ib, (size_t)P.Cin, P.Cin - ib, dys, DS, TR, 0, rows, Qi, nullptr,
                               nullptr, 0, tid, NTHREADS);
      __syncthreads();

      auto dA_dots = [&]() {
      // ---- adjacency gradient on the pattern: thread t owns entries t, t+256, ... and keeps their sums in registers
      //      across tiles and chunks (flushed once per workgroup) ----
      bool dots_done = false;
      if constexpr (DAM) if (P.dA) {
        dots_done = true;
        if (wave < K) {
          const int k = wave, rl = lane & 31, hq = lane >> 5;
          const int nks = (Qi + 1) >> 1;
          for (int f = 0; f < nf; ++f) {
            const T* xrow = dys + (f * V + rl) * DS;
            const T* drow = dxa + ((size_t)k * NCH * TR + f * V + rl) * EPL;
            for (int ks = 0; ks < nks; ++ks) {
              const int q = 2 * ks + hq;
              frag_t a, b;
              zero_frag<T>(a);
              zero_frag<T>(b);
              if (q < Qi) {
                a = *reinterpret_cast<const frag_t*>(xrow + q * EPL);
                b = *reinterpret_cast<const frag_t*>(drow + (size_t)q * TR * EPL);
              }
              if constexpr (sizeof(T) == 2) mma_kgroup(dAacc, a, b);
            }
          }
        }
      }
      if constexpr (!DAM) if (P.dA && !dots_done) {
#pragma unroll
        for (int pe = 0; pe < NPE; ++pe) {
          const int en = tid + pe * NTHREADS;
          if (en < nnz) {
            const T* xr = dys + r_v[en] * DS;
            const T* dr = dxa + r_ofs[en];
            float s = 0.f;
            for (int f = 0; f < nf; ++f) {
#pragma unroll 4
              for (int q = 0; q < Qi; ++q) {
                const frag_t a = *reinterpret_cast<const frag_t*>(xr + (f * V) * DS + q * EPL);
                const frag_t b = *reinterpret_cast<const frag_t*>(dr + (q * TR + f * V) * EPL);
                if constexpr (sizeof(T) == 2) {
                  s = dot8(a, b, s);           // v_dot2: two 16-bit products per lane-op, no conversions
                } else {
#pragma unroll
                  for (int e = 0; e < EPL; ++e) s += E::to_f(a[e]) * E::to_f(b[e]);
                }
              }
            }
            dsum[pe] += s;
          }
        }
      }
      };
      bool agg_done = false;
      if constexpr (AGGM) {
        // ---- bf16: transposed aggregation on the matrix cores.  Per (frame f, 32-channel tile ct):
        //      D[i][v] = sum_k sum_w dxa_k[(f,w)][i] * A_k[v][w]; dxa^T comes straight from the row-major images with
        //      ds_read_b64_tr_b16, the adjacency fragments from LDS; lane = joint v, 4 consecutive channels per quad. ----
        agg_done = true;
        // the adjacency gradient first (it needs x in `dys`); then `dys` becomes the staging buffer of dx, so the
        // global stores are whole 16-byte vectors of contiguous rows instead of 8-byte pieces of 25 different lines
        dA_dots();
        __syncthreads();
        const int CT = (CCi + 31) >> 5;
        const int grp = lane >> 4, h = grp >> 1, cblk = (grp & 1) * 16;
        const int q4 = (lane & 15) >> 2, pp = lane & 3;
        const int v = lane & 31;
        for (int pr = wave; pr < nf * CT; pr += 4) {
          const int f = pr / CT, ct = pr - f * CT;
          f32x16 d;
#pragma unroll
          for (int r = 0; r < 16; ++r) d[r] = 0.f;
          for (int k = 0; k < K; ++k) {
#pragma unroll
            for (int sstep = 0; sstep < 2; ++sstep) {
              const int c0 = ct * 32 + cblk + 4 * pp;
              const int cc = c0 < CCi ? c0 : 0;         // channels beyond the chunk: any finite data (results discarded)
              const T* r0 = dxa + ((k * NCH + cc / EPL) * TR + f * V + 16 * sstep + 8 * h + q4) * EPL + cc % EPL;
              const frag_t a = tr_pair<T>(r0, r0 + 4 * EPL);
              const frag_t bfr = *reinterpret_cast<const frag_t*>(afrag + ((k * 2 + sstep) * 64 + lane) * EPL);
              mma_kgroup(d, a, bfr);
            }
          }
          if (v < V) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const int il = ct * 32 + 8 * g + 4 * (lane >> 5);
              if (il < CCi) {
                float v4[4] = {d[4 * g], d[4 * g + 1], d[4 * g + 2], d[4 * g + 3]};
                store4(dys + (f * V + v) * DS + il, v4);
              }
            }
          }
        }
        __syncthreads();
        {
          // rows of the tile are contiguous in HBM: item -> (row, vector), UB rows per batch in flight
          constexpr int UB = 4;
          const int q = tid % Qi, r0 = tid / Qi, RS = NTHREADS / Qi;      // Qi divides 256 (CCi is a power of two here)
          const int i0 = ib + q * EPL;
          if ((NTHREADS % Qi) == 0 && i0 < P.Cin) {
            for (int rb = r0; rb < rows; rb += RS * UB) {
              frag_t sv[UB], av[UB];
              bool ok[UB];
#pragma unroll
              for (int u = 0; u < UB; ++u) {
                const int r = rb + u * RS;
                ok[u] = r < rows;
                const int rc = ok[u] ? r : rb;
                sv[u] = *reinterpret_cast<const frag_t*>(dys + rc * DS + q * EPL);
                if (addg) {
                  const size_t g = (pos0 + rc) * P.Cin + i0;
                  if (VEC) av[u] = *reinterpret_cast<const frag_t*>(addg + g);
                  else {
#pragma unroll
                    for (int e = 0; e < EPL; ++e) av[u][e] = (i0 + e < P.Cin) ? addg[g + e] : E::from_f(0.f);
                  }
                }
              }
#pragma unroll
              for (int u = 0; u < UB; ++u) {
                if (!ok[u]) continue;
                const size_t g = (pos0 + rb + u * RS) * P.Cin + i0;
                frag_t o = sv[u];
                if (addg) {
#pragma unroll
                  for (int e = 0; e < EPL; ++e) o[e] = E::from_f(E::to_f(sv[u][e]) + E::to_f(av[u][e]));
                }
                if (VEC) *reinterpret_cast<frag_t*>(dxg + g) = o;
                else {
#pragma unroll
                  for (int e = 0; e < EPL; ++e) if (i0 + e < P.Cin) dxg[g + e] = o[e];
                }
              }
            }
          } else if ((NTHREADS % Qi) != 0) {
            for (int it = tid; it < rows * Qi; it += NTHREADS) {        // odd chunk widths: plain item loop
              const int r = it / Qi, qq = it - r * Qi;
              const size_t g = (pos0 + r) * P.Cin + ib + qq * EPL;
#pragma unroll
              for (int e = 0; e < EPL; ++e) {
                if (ib + qq * EPL + e < P.Cin) {
                  float fv = E::to_f(dys[r * DS + qq * EPL + e]);
                  if (addg) fv += E::to_f(addg[g + e]);
                  dxg[g + e] = E::from_f(fv);
                }
              }
            }
          }
        }
      }
      if constexpr (!AGGM) if (!agg_done) {
      // ---- (VALU path) dx[(f,v)][i] = sum over row v of A: a * dxa_k[(f,w)][i]  (+ addend), straight to HBM.
      //      Wave w owns joints v = w, w+4, ... (wave-uniform entry lists); lanes span (frame, channel vector). ----
      {
        const int npair = nf * Qi;
        for (int v = wave; v < V; v += 4) {
          const int e0 = r_off[v], e1 = min(r_off[v + 1], P.nnz_cap);
          for (int pr = lane; pr < npair; pr += 64) {
            const int f = pr / Qi, q = pr - f * Qi;
            const int i0 = ib + q * EPL;
            if (i0 >= P.Cin) continue;
            float sum[EPL];
#pragma unroll
            for (int e = 0; e < EPL; ++e) sum[e] = 0.f;
            const T* dbase = dxa + (q * TR + f * V) * EPL;
            for (int en = e0; en < e1; ++en) {
              const float a = r_a[en];
              const frag_t dv = *reinterpret_cast<const frag_t*>(dbase + r_ofs[en]);
#pragma unroll
              for (int e = 0; e < EPL; ++e) sum[e] += a * E::to_f(dv[e]);
            }
            const size_t g = (pos0 + f * V + v) * P.Cin + i0;
            if (VEC) {
              frag_t o;
              if (addg) {
                const frag_t av = *reinterpret_cast<const frag_t*>(addg + g);
#pragma unroll
                for (int e = 0; e < EPL; ++e) sum[e] += E::to_f(av[e]);
              }
#pragma unroll
              for (int e = 0; e < EPL; ++e) o[e] = E::from_f(sum[e]);
              *reinterpret_cast<frag_t*>(dxg + g) = o;
            } else {
#pragma unroll
              for (int e = 0; e < EPL; ++e) {
                if (i0 + e < P.Cin) {
                  float fv = sum[e];
                  if (addg) fv += E::to_f(addg[g + e]);
                  dxg[g + e] = E::from_f(fv);
                }
              }
            }
          }
        }
      }
      }
      if (!agg_done) dA_dots();
      __syncthreads();
    }
  }

  if (P.dA) {
    bool flushed = false;
    if constexpr (DAM) {
      flushed = true;
      if (wave < K) {
        const int w = lane & 31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int v = mfma_row(r, lane);
          if (v < V && w < V && (P.pat ? P.pat : P.A)[(wave * V + v) * V + w] != 0.f)     // gradient on the sparsity pattern only
            atomicAdd(P.dA + (wave * V + v) * V + w, dAacc[r]);
        }
      }
    }
    if constexpr (!DAM) if (!flushed) {
#pragma unroll
      for (int pe = 0; pe < NPE; ++pe) {
        const int en = tid + pe * NTHREADS;
        if (en < nnz) {
          const int kw = r_kw[en];
          const int k = kw / V, w = kw - k * V;
          atomicAdd(P.dA + (k * V + r_v[en]) * V + w, dsum[pe]);
        }
      }
    }
  }
}


struct GbdGeom { int CCi, nchi, CCc, nchc, NKGc, KKp, MTK; };

inline void gbd_geom(int Cin, int Cout, int K, int dtype, GbdGeom* G) {
  const int epl = dtype == 0 ? 4 : 8, cc = dtype == 0 ? 32 : 64, kgs = 2 * epl;
  G->CCi = Cin >= cc ? cc : round_up(Cin, epl);
  G->nchi = ceil_div(Cin, G->CCi);
  G->CCc = Cout >= cc ? cc : round_up(Cout, kgs);
  G->nchc = ceil_div(Cout, G->CCc);
  G->NKGc = G->CCc / kgs;
  G->KKp = round_up(K * G->CCi, 32);
  G->MTK = G->KKp / 32;
}

template <typename T, int MTK>
int launch_mtk(GbdParams& P, int grid_cap, size_t lds, hipStream_t stream) {
  const bool vec = (P.Cin % Elem<T>::EPL) == 0 && (P.Cout % Elem<T>::EPL) == 0;
#define GO(VV, AG)                                                                                          \
  do {                                                                                                      \
    auto kfn = gcn_bwd_kernel<T, MTK, VV, AG>;                                                                \
    static std::atomic<unsigned long long> optin{0};                                                        \
    if (int ea_ = istgcn_lds_optin((const void*)kfn, optin)) return ea_;                                    \
    int gx = grid_cap > 0 ? grid_cap : istgcn_resident_blocks((const void*)kfn, NTHREADS, lds);             \
    gx = gx < 1 ? 1 : (gx > P.total_tiles ? P.total_tiles : gx);                                            \
    ISTGCN_LAUNCH(kfn, dim3(gx), dim3(NTHREADS), lds, stream, P);                                           \
  } while (0)
  if constexpr (sizeof(T) == 2) {
    if (P.V <= 32) { if (vec) GO(true, true); else GO(false, true); }
    else { if (vec) GO(true, false); else GO(false, false); }
  } else {
    if (vec) GO(true, false); else GO(false, false);
  }
#undef GO
  ISTGCN_CHECK_LAUNCH();
  return ISTGCN_OK;
}

template <typename T>
int launch_T(GbdParams& P, const GbdGeom& G, int grid_cap, hipStream_t stream) {
  const int esz = sizeof(T), epl = Elem<T>::EPL;
  P.CCi = G.CCi; P.nchi = G.nchi; P.CCc = G.CCc; P.nchc = G.nchc; P.NKGc = G.NKGc;
  P.F = TR / P.V;
  P.tiles_per_seq = ceil_div(P.T, P.F);
  P.total_tiles = P.NM * P.tiles_per_seq;
  const int wide = P.CCc > P.CCi ? P.CCc : P.CCi;
  P.ds_stride = wide + epl;
  size_t off = (size_t)(P.V + 1) * 4;
  off = (off + 15) & ~(size_t)15; P.off_rv = (int)off; off += P.nnz_cap;
  off = (off + 15) & ~(size_t)15; P.off_rkw = (int)off; off += (size_t)P.nnz_cap * 2;
  off = (off + 15) & ~(size_t)15; P.off_ra = (int)off; off += (size_t)P.nnz_cap * 4;
  off = (off + 15) & ~(size_t)15; P.off_dacc = (int)off; off += (size_t)P.nnz_cap * 4;
  off = (off + 15) & ~(size_t)15; P.off_rows = (int)off; off += 2 * TR;
  off = (off + 15) & ~(size_t)15; P.off_afrag = (int)off; off += esz == 2 ? (size_t)P.K * 2 * 64 * 16 : 0;
  off = (off + 15) & ~(size_t)15; P.off_dys = (int)off; off += (size_t)TR * P.ds_stride * esz;
  off = (off + 15) & ~(size_t)15; P.off_dxa = (int)off;
  size_t dxa = ((size_t)P.K * TR + 32) * P.CCi * esz, al = (size_t)P.K * P.V * P.V * 4;
  off += dxa > al ? dxa : al;
  if (off > 160 * 1024) return ISTGCN_EINVAL;
  switch (G.MTK) {
    case 1: return launch_mtk<T, 1>(P, grid_cap, off, stream);
    case 2: return launch_mtk<T, 2>(P, grid_cap, off, stream);
    case 3: return launch_mtk<T, 3>(P, grid_cap, off, stream);
    case 4: return launch_mtk<T, 4>(P, grid_cap, off, stream);
    case 5: return launch_mtk<T, 5>(P, grid_cap, off, stream);
    case 6: return launch_mtk<T, 6>(P, grid_cap, off, stream);
    case 7: return launch_mtk<T, 7>(P, grid_cap, off, stream);
    case 8: return launch_mtk<T, 8>(P, grid_cap, off, stream);
    default: return ISTGCN_EINVAL;
  }
}

}  // namespace

extern "C" int istgcn_gcn_bwd_geometry(int Cin, int Cout, int K, int dtype, int* CCi, int* nchi, int* CCc, int* nchc,
                                       int* KKp, int* EPL) {
  if (!istgcn_dtype_ok(dtype) || Cin < 1 || Cout < 1 || K < 1 || K > 4) return ISTGCN_EINVAL;
  GbdGeom G;
  gbd_geom(Cin, Cout, K, dtype, &G);
  *CCi = G.CCi; *nchi = G.nchi; *CCc = G.CCc; *nchc = G.nchc; *KKp = G.KKp; *EPL = dtype == 0 ? 4 : 8;
  return ISTGCN_OK;
}

// Register-chained kernel (gcn_rc_bwd.hip): 16-bit storage, 64/128/256 output channels, input channels a multiple of
// 64, V <= 32; its weights are a second section of Wb.  ISTGCN_GCN_RC=0 disables it.
extern "C" int istgcn_gcn_bwd_rc_layout(int Cin, int Cout, int K, int dtype);
extern "C" long long istgcn_gcn_bwd_rc_offset(int Cin, int Cout, int K, int dtype);
extern "C" int istgcn_gcn_bwd_data_rc(const void* dy, const void* x, const float* A, const float* pattern, const void* Wq,
                                      const void* addend, const unsigned char* addend_mask, void* dx, float* dA, int NM, int T,
                                      int V, int Cin, int Cout, int K, int dtype, int grid_cap, void* stream);

// Can istgcn_gcn_bwd_data take its addend as (tensor, ReLU byte mask)?  Only the register-chained kernel does (istgcn.h).
extern "C" int istgcn_gcn_bwd_addend_mask_ok(int V, int Cin, int Cout, int K, int dtype) {
  static const bool rc_on = [] { const char* e = getenv("ISTGCN_GCN_RC"); return !e || atoi(e) != 0; }();
  return (rc_on && V <= 32 && Cin != 3 && Cin % 8 == 0 && istgcn_gcn_bwd_rc_layout(Cin, Cout, K, dtype) &&
          istgcn_gcn_bwd_rc_offset(Cin, Cout, K, dtype) >= 0) ? 1 : 0;
}

extern "C" int istgcn_gcn_bwd_data(const void* dy, const void* x, const float* A, const float* pattern, const void* Wb,
                                   const void* addend, const unsigned char* addend_mask, void* dx, float* dA, int NM, int T,
                                   int V, int Cin, int Cout, int K, int nnz_cap, int dtype, int grid_cap, void* stream) {
  if (!dy || !A || !Wb || (!dx && !dA)) return ISTGCN_EINVAL;
  if (dA && !x) return ISTGCN_EINVAL;
  if (V < 1 || V > 128 || Cin < 1 || Cout < 1 || K < 1 || K > 4 || NM < 0 || T < 0) return ISTGCN_EINVAL;
  if (nnz_cap < 1 || nnz_cap > K * V * V || (dA && nnz_cap > 16 * NTHREADS)) return ISTGCN_EINVAL;
  if (!istgcn_dtype_ok(dtype)) return ISTGCN_EINVAL;
  if (NM == 0 || T == 0) return ISTGCN_OK;
  {
    // dispatch override ISTGCN_GCN_RC=0 (the round-2 kernels: A/B timing, one process per setting), read once
    static const bool rc_on = [] { const char* e = getenv("ISTGCN_GCN_RC"); return !e || atoi(e) != 0; }();
    // (round 5: 256 output channels WITH the adjacency gradient too -- H' transposed from H, gcn_rc_bwd.hip; the
    //  wave-specialised round-2 kernel that served them is gone)
    if (rc_on && V <= 32 && istgcn_gcn_bwd_rc_layout(Cin, Cout, K, dtype) &&
        (Cin != 3 ? dx != nullptr : (dA && !addend))) {      // (first layer: dA always, dx optional)
      const long long off = istgcn_gcn_bwd_rc_offset(Cin, Cout, K, dtype);
      if (off >= 0)
        return istgcn_gcn_bwd_data_rc(dy, x, A, pattern, reinterpret_cast<const char*>(Wb) + (size_t)off * 2, addend, addend_mask, dx, dA,
                                      NM, T, V, Cin, Cout, K, dtype, grid_cap, stream);
    }
  }
  if (!dx) return ISTGCN_EINVAL;          // dx == NULL (adjacency gradient only) exists for the 16-bit first layer only; callers allocate dx otherwise
  if (addend_mask) return ISTGCN_EINVAL;  // the masked addend exists in the register-chained kernel only (istgcn_gcn_bwd_addend_mask_ok)
  GbdParams P{};
  P.dy = dy; P.x = x; P.A = A; P.pat = pattern; P.Wb = Wb; P.addend = addend; P.dx = dx; P.dA = dA;
  P.NM = NM; P.T = T; P.V = V; P.Cin = Cin; P.Cout = Cout; P.K = K; P.nnz_cap = nnz_cap;
  GbdGeom G;
  gbd_geom(Cin, Cout, K, dtype, &G);
  if (dtype == 0) return launch_T<float>(P, G, grid_cap, (hipStream_t)stream);
  if (dtype == 2) return launch_T<_Float16>(P, G, grid_cap, (hipStream_t)stream);
  return launch_T<__bf16>(P, G, grid_cap, (hipStream_t)stream);
}
