// Round-1 graph-convolution forward kernel (two independent 4-wave workgroups per CU, every wave stages, aggregates,
// contracts and stores), kept for fp32 storage: there the aggregation is a VALU pass over the compressed adjacency
// columns either way, and the wave-specialised kernel of gcn_fwd.hip has to halve its fp32 chunk width at 128 output
// channels to fit its double-buffered chunk + fp32 output image in LDS (fp32, NM=128: 64->64 532 vs 545 us, 128->128 686
// vs 1010, 256->256 1389 vs 1954).  istgcn_gcn_fwd / istgcn_gcn_geometry (gcn_fwd.hip) dispatch by storage type; the packed
// weight layout is the same family with THIS file's chunk width and unpadded contraction length for the fp32 packs.
// Graph-convolution unit, forward: y = einsum('nkctv,kvw->nctw', conv1x1(x), A)  (+ per-joint bias term)
// computed aggregate-first:
//     xa[p=(t,w)][(k,i)] = sum_v A[k][v][w] * x[(t,v)][i]          sparse VALU pass, LDS -> LDS
//     y[p][c]            = sum_(k,i) Wr[c][(k,i)] * xa[p][(k,i)]   MFMA 32x32, fp32 accumulate
// Replaces (reference file:line): net/utils/tgcn.py:76-89, net/utils/tgcn_multi3_fix_3A.py:76-92,
// net/utils/inceptionv2_gcn.py:64-89 (all variants fold into one effective adjacency A, host side),
// and -- with K=1, A=I and a frame stride -- the residual 1x1 strided Conv2d of st_gcnold.py:186-193.
// The same kernel run on dy with A^T and the transposed weights is the unit's data gradient.
//
// One workgroup (4 waves) owns a tile of F = floor(128/V) whole frames of one sequence (<=128 rows
// of the NTVC tensor, contiguous in HBM) x all output channels of its grid.y block; workgroups walk
// tiles in a grid-stride loop so that the adjacency lists, BatchNorm partial sums and the
// weight-fragment working set are amortised.  Wave w owns rows [32w, 32w+32) (the MFMA "column" axis);
// output channels are the MFMA "row" axis, so each lane ends up with 4 consecutive channels of one
// row per register quad, which is what the LDS-staged, fully coalesced epilogue wants.
#include "common.hpp"

namespace {

struct GcnFwdParams {
  const void* x;
  const float* A;        // [K][V][V]  A[k][v][w]
  const void* Wp;        // fragment-ordered weights, see istgcn.h
  const float* bterm;    // [V][Cout] or null
  const void* addend;    // same layout as y or null (may alias y)
  void* y;
  double* stats;         // [stats_rep][2][Cout] or null
  int* status;           // overflow flag or null
  int NM, Tin, Tout, Tlog, V, Cin, Cout, K;
  int in_t_stride, out_t_stride;
  int nnz_cap, stats_rep;
  int F, tiles_per_seq, total_tiles;
  int CCeff, nch, KKp, NKG, MTtot;
  int xs_stride, xa_stride, out_stride, xs_rows;   // in elements / rows
  int off_csr_v, off_csr_a, off_stat, off_rows, off_afrag, off_work;  // LDS byte offsets
};

constexpr int TILE_ROWS = 128;
constexpr int NTHREADS = 256;

template <typename T, int MT, bool VEC_IN, bool VEC_OUT>
__global__ __launch_bounds__(NTHREADS, MT <= 4 ? 2 : 1) void gcn_fwd_kernel(const GcnFwdParams P) {
  using E = Elem<T>;
  constexpr int EPL = E::EPL;
  constexpr int KGS = E::KGS;
  typedef typename E::frag frag_t;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  int* csr_off = reinterpret_cast<int*>(smem);                       // [K*V+1]
  unsigned char* csr_v = smem + P.off_csr_v;                         // [nnz_cap]
  float* csr_a = reinterpret_cast<float*>(smem + P.off_csr_a);       // [nnz_cap]
  float* stat = reinterpret_cast<float*>(smem + P.off_stat);         // [2][MT*32]
  unsigned char* row_f = smem + P.off_rows;                          // [128]
  unsigned char* row_w = row_f + TILE_ROWS;                          // [128]
  unsigned char* col_k = row_w + TILE_ROWS;                          // [K*V]
  unsigned char* col_w = col_k + P.K * P.V;                          // [K*V]
  T* afrag = reinterpret_cast<T*>(smem + P.off_afrag);               // bf16 only: [K][2][64][8] MFMA fragments of A_k
  T* xs = reinterpret_cast<T*>(smem + P.off_work);                   // [xs_rows][xs_stride]
  T* xa = xs + P.xs_rows * P.xs_stride;                              // [128][xa_stride]
  T* outs = xs;                                                      // [128][out_stride] (aliases xs/xa)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int V = P.V, K = P.K;
  const int KV = K * V;
  constexpr bool MSPLIT = (MT % 2) == 0;
  constexpr int MH = MSPLIT ? MT / 2 : MT;           // channel tiles per wave
  constexpr int NTW = MSPLIT ? 2 : 1;                // 32-row tiles per wave
  const int ph = MSPLIT ? (wave & 1) : wave, mh = MSPLIT ? (wave >> 1) : 0;
  const int mt0 = blockIdx.y * MT;
  const int cbase_blk = mt0 * 32;

  // ---- one-time setup: CSR lists of the adjacency columns, row tables, stat accumulators ----
  for (int c = tid; c <= KV; c += NTHREADS) csr_off[c] = 0;
  for (int c = tid; c < 2 * MT * 32; c += NTHREADS) stat[c] = 0.f;
  for (int r = tid; r < TILE_ROWS; r += NTHREADS) {
    int f = r / V;
    row_f[r] = (unsigned char)f;
    row_w[r] = (unsigned char)(r - f * V);
  }
  __syncthreads();
  for (int col = tid; col < KV; col += NTHREADS) {
    int k = col / V, w = col - k * V, cnt = 0;
    col_k[col] = (unsigned char)k;
    col_w[col] = (unsigned char)w;
    for (int v = 0; v < V; ++v) cnt += (P.A[(k * V + v) * V + w] != 0.f);
    csr_off[col + 1] = cnt;
  }
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int c = 0; c < KV; ++c) { int n = csr_off[c + 1]; csr_off[c] = run; run += n; }
    csr_off[KV] = run;
    if (run > P.nnz_cap && P.status) *P.status = 1;
  }
  __syncthreads();
  for (int col = tid; col < KV; col += NTHREADS) {
    int k = col / V, w = col - k * V, e = csr_off[col];
    for (int v = 0; v < V; ++v) {
      float a = P.A[(k * V + v) * V + w];
      if (a != 0.f) {
        if (e < P.nnz_cap) { csr_v[e] = (unsigned char)v; csr_a[e] = a; }
        ++e;
      }
    }
  }
  __syncthreads();

  if constexpr (sizeof(T) == 2) {
    // B-operand fragments of the adjacency for the MFMA aggregation: lane (w = lane&31, h = lane>>5), k-step s,
    // element j holds A[k][v = 16s + 8h + j][w] (zero outside the V x V block)
    for (int idx = tid; idx < K * 2 * 64; idx += NTHREADS) {
      const int ln = idx & 63, sstep = (idx >> 6) & 1, k = idx >> 7;
      const int w = ln & 31, h = ln >> 5;
      frag_t fr;
#pragma unroll
      for (int j = 0; j < EPL; ++j) {
        const int v = 16 * sstep + 8 * h + j;
        fr[j] = E::from_f((v < V && w < V) ? P.A[(k * V + v) * V + w] : 0.f);
      }
      *reinterpret_cast<frag_t*>(afrag + idx * EPL) = fr;
    }
    __syncthreads();
  }
  const int Q = P.CCeff / EPL;            // channel vectors per partition
  const int NV = P.KKp / EPL;             // vectors per xa row (incl. zero padding)
  const T* xg = reinterpret_cast<const T*>(P.x);
  const T* Wp = reinterpret_cast<const T*>(P.Wp);
  T* yg = reinterpret_cast<T*>(P.y);
  const T* addg = reinterpret_cast<const T*>(P.addend);

  // BatchNorm partial sums: a thread always copies out the same channel vector, so it keeps its sums in registers for
  // the whole grid-stride walk and the cross-lane reduction happens once per workgroup, not once per tile
  constexpr int NPASS_ = (MT + 1) / 2;
  float st1[NPASS_][EPL], st2[NPASS_][EPL];
#pragma unroll
  for (int ps = 0; ps < NPASS_; ++ps)
#pragma unroll
    for (int j = 0; j < EPL; ++j) { st1[ps][j] = 0.f; st2[ps][j] = 0.f; }

  for (int tile = blockIdx.x; tile < P.total_tiles; tile += gridDim.x) {
    const int n = tile / P.tiles_per_seq;
    const int t0 = (tile - n * P.tiles_per_seq) * P.F;
    const int nf = min(P.F, P.Tlog - t0);
    const int rows = nf * V;

    // Wave decomposition of the contraction: with an even number of channel tiles the four waves split 2 (64-row
    // halves) x 2 (channel-tile halves), so a weight fragment feeds TWO MFMAs -- one fragment per MFMA asks the vector
    // L1 for 128 B/clk per CU, twice what it delivers.  (Odd MT: one 32-row slab and all tiles per wave, as before.)
    // accumulators start at the bias term bterm[w][c] of their (row, channel): loads issued here, behind the staging
    f32x16 acc[MH][NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
      const int p = ph * 32 * NTW + t * 32 + (lane & 31);
      const bool rowb = P.bterm && p < rows;
      const float* brow = P.bterm + (rowb ? (int)row_w[p] : 0) * P.Cout + cbase_blk + 4 * (lane >> 5);
#pragma unroll
      for (int m = 0; m < MH; ++m) {
        const int mg = mh * MH + m;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float bv[4] = {0.f, 0.f, 0.f, 0.f};
          const int cg = cbase_blk + mg * 32 + 8 * g + 4 * (lane >> 5);
          if (rowb) {
            if (VEC_OUT && cg + 3 < P.Cout) {
              const f32x4 b4 = *reinterpret_cast<const f32x4*>(brow + mg * 32 + 8 * g);
              bv[0] = b4[0]; bv[1] = b4[1]; bv[2] = b4[2]; bv[3] = b4[3];
            } else {
#pragma unroll
              for (int j = 0; j < 4; ++j) if (cg + j < P.Cout) bv[j] = brow[mg * 32 + 8 * g + j];
            }
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[m][t][4 * g + j] = bv[j];
        }
      }
    }

    for (int ch = 0; ch < P.nch; ++ch) {
      const int cb = ch * P.CCeff;
      // ---- stage x chunk: rows x CCeff channels -> xs (zero beyond Cin) ----
      if (P.in_t_stride == 1) {
        stage_block<T, 8, VEC_IN>(xg + ((size_t)(n * P.Tin + t0) * V) * P.Cin + cb, (size_t)P.Cin, P.Cin - cb, xs,
                                  P.xs_stride, rows, 0, rows, Q, nullptr, nullptr, 0, tid, NTHREADS);
      } else {
        const int tot = rows * Q;
        for (int it = tid; it < tot; it += NTHREADS) {
          int r = it / Q, q = it - r * Q;
          int f = row_f[r], v = row_w[r];
          size_t g = ((size_t)(n * P.Tin + (t0 + f) * P.in_t_stride) * V + v) * P.Cin + cb + q * EPL;
          frag_t val;
          if (VEC_IN) {
            if (cb + q * EPL < P.Cin) val = *reinterpret_cast<const frag_t*>(xg + g);
            else zero_frag<T>(val);
          } else {
#pragma unroll
            for (int j = 0; j < EPL; ++j) val[j] = (cb + q * EPL + j < P.Cin) ? xg[g + j] : E::from_f(0.f);
          }
          *reinterpret_cast<frag_t*>(xs + r * P.xs_stride + q * EPL) = val;
        }
      }
      __syncthreads();
      // weight fragments of the first ring slots: fetched now, in flight during the aggregation (which touches only LDS).
      // The contraction loop is 12 k-groups long; an L2 round trip in front of its first MFMA was most of its time.
      constexpr int DEPTH = sizeof(T) == 4 ? 2 : (MT <= 2 ? 4 : (MT <= 4 ? 3 : 2));
      const T* wfrag = Wp + ((size_t)(ch * P.MTtot + mt0) * P.NKG * 64 + lane) * EPL;
      auto load_a = [&](int kg, frag_t (&a)[MH]) {
#pragma unroll
        for (int m = 0; m < MH; ++m)
          a[m] = *reinterpret_cast<const frag_t*>(wfrag + ((size_t)(mh * MH + m) * P.NKG + kg) * 64 * EPL);
      };
      MfmaRing<DEPTH, MH, NTW, frag_t> ring;
      ring_prime_a(ring, P.NKG, load_a);
      bool agg_done = false;
      if constexpr (sizeof(T) == 2) if (V <= 32) {
        agg_done = true;
        // ---- bf16: aggregation on the matrix cores.  Per (frame f, 32-channel tile ct): D[i][w] = sum_v x[(f,v)][i] *
        //      A_k[v][w], x^T read straight from the row-major tile with ds_read_b64_tr_b16 (rows v beyond the frame
        //      multiply zero adjacency rows), A_k fragments from LDS; the VALU version of this pass cost ~2500
        //      instructions per wave and tile (conversions + addressing) against 24 MFMAs of real work. ----
        const int CT = (P.CCeff + 31) >> 5;
        const int npair = nf * CT;
        const int grp = lane >> 4, h = grp >> 1, cblk = (grp & 1) * 16;
        const int q4 = (lane & 15) >> 2, pp = lane & 3;
        // the last frame's 32-row k-range reaches rows [rows, (nf-1)*V + 32): keep them finite (they meet zero adjacency)
        const int zrows = (nf - 1) * V + 32 - rows, zq = P.xs_stride / EPL;
        for (int idx = tid; idx < zrows * zq; idx += NTHREADS) {
          frag_t z;
          zero_frag<T>(z);
          *reinterpret_cast<frag_t*>(xs + (rows + idx / zq) * P.xs_stride + (idx % zq) * EPL) = z;
        }
        __syncthreads();
        for (int pr = wave; pr < npair; pr += 4) {
          const int f = pr / CT, ct = pr - f * CT;
          frag_t a[2];
#pragma unroll
          for (int sstep = 0; sstep < 2; ++sstep) {
            const T* r0 = xs + (f * V + 16 * sstep + 8 * h + q4) * P.xs_stride + ct * 32 + cblk + 4 * pp;
            a[sstep] = tr_pair<T>(r0, r0 + 4 * P.xs_stride);
          }
          const int w = lane & 31;
          for (int k = 0; k < K; ++k) {
            f32x16 d;
#pragma unroll
            for (int r = 0; r < 16; ++r) d[r] = 0.f;
            const frag_t b0 = *reinterpret_cast<const frag_t*>(afrag + ((k * 2 + 0) * 64 + lane) * EPL);
            const frag_t b1 = *reinterpret_cast<const frag_t*>(afrag + ((k * 2 + 1) * 64 + lane) * EPL);
            mma_kgroup(d, a[0], b0);
            mma_kgroup(d, a[1], b1);
            if (w < V) {
#pragma unroll
              for (int g = 0; g < 4; ++g) {
                const int i0 = ct * 32 + 8 * g + 4 * (lane >> 5);
                if (i0 < P.CCeff) {
                  float v4[4] = {d[4 * g], d[4 * g + 1], d[4 * g + 2], d[4 * g + 3]};
                  store4(xa + (f * V + w) * P.xa_stride + k * P.CCeff + i0, v4);
                }
              }
            }
          }
        }
      }
      if (!agg_done) {
      // ---- sparse aggregation xs -> xa.  Wave w owns adjacency columns col = w, w+4, ... (their compressed lists are
        //      wave-uniform: no divergence, LDS broadcast reads); lanes span (frame, channel vector).  Rows >= rows are
        //      never written: they only feed output rows that are never stored. ----
        {
          const int npair = nf * Q;
          for (int col = wave; col < KV; col += 4) {
            const int k = col_k[col], w = col_w[col];
            const int e0 = csr_off[col], e1 = min(csr_off[col + 1], P.nnz_cap);
            for (int pr = lane; pr < npair; pr += 64) {
              const int f = pr / Q, q = pr - f * Q;
              const T* xrow = xs + (f * V) * P.xs_stride + q * EPL;
              float sum[EPL];
#pragma unroll
              for (int j = 0; j < EPL; ++j) sum[j] = 0.f;
              for (int e = e0; e < e1; ++e) {
                const float a = csr_a[e];
                const frag_t xv = *reinterpret_cast<const frag_t*>(xrow + csr_v[e] * P.xs_stride);
#pragma unroll
                for (int j = 0; j < EPL; ++j) sum[j] += a * E::to_f(xv[j]);
              }
              frag_t o;
#pragma unroll
              for (int j = 0; j < EPL; ++j) o[j] = E::from_f(sum[j]);
              *reinterpret_cast<frag_t*>(xa + (f * V + w) * P.xa_stride + k * P.CCeff + q * EPL) = o;
            }
          }
        }
      }
      {
        if (NV > K * Q) {          // contraction padding columns (tiny Cin only) must be finite: zero them
          const int padv = NV - K * Q;
          for (int idx = tid; idx < TILE_ROWS * padv; idx += NTHREADS) {
            const int r = idx / padv, c = idx - r * padv;
            frag_t o;
            zero_frag<T>(o);
            *reinterpret_cast<frag_t*>(xa + r * P.xa_stride + (K * Q + c) * EPL) = o;
          }
        }
      }
      __syncthreads();
      // ---- channel contraction on the matrix cores (ring of weight fragments from L2 and xa fragments from LDS) ----
      {
        const T* brow = xa + (ph * 32 * NTW + (lane & 31)) * P.xa_stride + (lane >> 5) * EPL;
        auto load_b = [&](int kg, frag_t (&b)[NTW]) {
#pragma unroll
          for (int t = 0; t < NTW; ++t) b[t] = *reinterpret_cast<const frag_t*>(brow + t * 32 * P.xa_stride + kg * KGS);
        };
        auto mma_step = [&](const frag_t (&a)[MH], const frag_t (&b)[NTW]) {
#pragma unroll
          for (int m = 0; m < MH; ++m)
#pragma unroll
            for (int t = 0; t < NTW; ++t) mma_kgroup(acc[m][t], a[m], b[t]);
        };
        ring_run(ring, P.NKG, load_a, load_b, mma_step);
      }
      __syncthreads();   // xa / xs free again (next chunk or the epilogue's staging buffer)
    }

    // ---- epilogue: accumulators -> LDS (row-major, channels innermost) -> coalesced HBM store ----
    constexpr int NPASS = (MT + 1) / 2;
    constexpr int VPR = 64 / EPL;                // vectors per staged row
    constexpr int RSTEP = NTHREADS / VPR;
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      {
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
          const int p = ph * 32 * NTW + t * 32 + (lane & 31);
#pragma unroll
          for (int ml = 0; ml < 2; ++ml) {
            const int mg = 2 * ps + ml;                    // channel tile of this pass; held by the waves with mh == mg / MH
            const int m = mg % MH;
            if (mg < MT && mg / MH == mh) {
#pragma unroll
              for (int g = 0; g < 4; ++g) {
                const int cl = ml * 32 + 8 * g + 4 * (lane >> 5);
                float v4[4] = {acc[m][t][4 * g], acc[m][t][4 * g + 1], acc[m][t][4 * g + 2], acc[m][t][4 * g + 3]};
                store4(outs + p * P.out_stride + cl, v4);
              }
            }
          }
        }
      }
      __syncthreads();
      {
        const int vq = tid % VPR;
        const int cg = cbase_blk + ps * 64 + vq * EPL;
        float s1[EPL], s2[EPL];
#pragma unroll
        for (int j = 0; j < EPL; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
        const bool col_live = (2 * ps * 32 + vq * EPL) < MT * 32 && cg < P.Cout;
        if (col_live) {
          for (int r = tid / VPR; r < rows; r += RSTEP) {
            const int f = row_f[r], w = row_w[r];
            const size_t g = ((size_t)(n * P.Tout + (t0 + f) * P.out_t_stride) * V + w) * P.Cout + cg;
            const frag_t sv = *reinterpret_cast<const frag_t*>(outs + r * P.out_stride + vq * EPL);
            if (VEC_OUT) {
              frag_t o = sv;
              if (addg) {
                const frag_t av = *reinterpret_cast<const frag_t*>(addg + g);
#pragma unroll
                for (int j = 0; j < EPL; ++j) o[j] = E::from_f(E::to_f(sv[j]) + E::to_f(av[j]));
              }
              *reinterpret_cast<frag_t*>(yg + g) = o;
#pragma unroll
              for (int j = 0; j < EPL; ++j) { float fv = E::to_f(o[j]); s1[j] += fv; s2[j] += fv * fv; }
            } else {
#pragma unroll
              for (int j = 0; j < EPL; ++j) {
                if (cg + j < P.Cout) {
                  float fv = E::to_f(sv[j]);
                  if (addg) fv += E::to_f(addg[g + j]);
                  const T o = E::from_f(fv);
                  yg[g + j] = o;
                  fv = E::to_f(o);
                  s1[j] += fv; s2[j] += fv * fv;
                }
              }
            }
          }
        }
#pragma unroll
        for (int j = 0; j < EPL; ++j) { st1[ps][j] += s1[j]; st2[ps][j] += s2[j]; }
      }
      __syncthreads();
    }
  }

  if (P.stats) {
    constexpr int VPR = 64 / EPL;
    const int vq = tid % VPR;
#pragma unroll
    for (int ps = 0; ps < NPASS_; ++ps) {
#pragma unroll
      for (int j = 0; j < EPL; ++j) {
        float a = st1[ps][j], b = st2[ps][j];
#pragma unroll
        for (int msk = VPR; msk < 64; msk <<= 1) { a += __shfl_xor(a, msk); b += __shfl_xor(b, msk); }
        const int cl = ps * 64 + vq * EPL + j;
        if (lane < VPR && cl < MT * 32 && cbase_blk + cl < P.Cout) {
          atomicAdd(&stat[cl], a);
          atomicAdd(&stat[MT * 32 + cl], b);
        }
      }
    }
    __syncthreads();
    double* dst = P.stats + (size_t)(blockIdx.x % P.stats_rep) * 2 * P.Cout;
    for (int c = tid; c < MT * 32; c += NTHREADS) {
      if (cbase_blk + c < P.Cout) {
        atomic_add_f64(dst + cbase_blk + c, (double)stat[c]);
        atomic_add_f64(dst + P.Cout + cbase_blk + c, (double)stat[MT * 32 + c]);
      }
    }
  }
}

template <typename T, int MT>
int launch_mt(const GcnFwdParams& P, int grid_cap, int gy, size_t lds, hipStream_t stream) {
  constexpr int EPL = Elem<T>::EPL;
  const bool vin = (P.Cin % EPL) == 0, vout = (P.Cout % EPL) == 0;
#define GO(VI, VO)                                                                                          \
  do {                                                                                                      \
    auto kfn = gcn_fwd_kernel<T, MT, VI, VO>;                                                               \
    static std::atomic<unsigned long long> optin{0};                                                        \
    if (int ea_ = istgcn_lds_optin((const void*)kfn, optin)) return ea_;                                    \
    int gx = (grid_cap > 0 ? grid_cap : istgcn_resident_blocks((const void*)kfn, NTHREADS, lds)) / gy;      \
    gx = gx < 1 ? 1 : (gx > P.total_tiles ? P.total_tiles : gx);                                            \
    ISTGCN_LAUNCH(kfn, dim3(gx, gy), dim3(NTHREADS), lds, stream, P);                                       \
  } while (0)
  if (vin && vout) GO(true, true);
  else if (vin) GO(true, false);
  else if (vout) GO(false, true);
  else GO(false, false);
#undef GO
  ISTGCN_CHECK_LAUNCH();
  return ISTGCN_OK;
}

template <typename T>
int launch_T(GcnFwdParams& P, int grid_x_cap, hipStream_t stream) {
  using E = Elem<T>;
  constexpr int EPL = E::EPL;
  P.CCeff = P.Cin >= E::CC ? E::CC : round_up(P.Cin, EPL);
  P.nch = ceil_div(P.Cin, P.CCeff);
  P.KKp = round_up(P.K * P.CCeff, E::KGS);
  P.NKG = P.KKp / E::KGS;
  if (P.KKp / EPL > 64) return ISTGCN_EINVAL;
  // at most 4 channel tiles per workgroup: the 8-tile kernel needs > 256 VGPRs, i.e. ONE 4-wave workgroup per CU, and
  // lost more to exposed latency than the second channel block costs in repeated aggregation
  int MT = P.Cout <= 32 ? 1 : P.Cout <= 64 ? 2 : 4;
  int gy = ceil_div(P.Cout, MT * 32);
  P.MTtot = gy * MT;
  P.F = TILE_ROWS / P.V;
  P.tiles_per_seq = ceil_div(P.Tlog, P.F);
  P.total_tiles = P.NM * P.tiles_per_seq;
  const bool mfma_agg = sizeof(T) == 2 && P.V <= 32;
  P.xs_stride = mfma_agg ? round_up(P.CCeff, 32) : P.CCeff + EPL;
  P.xs_rows = mfma_agg ? (P.F - 1) * P.V + 32 : TILE_ROWS;
  if (P.xs_rows < TILE_ROWS) P.xs_rows = TILE_ROWS;
  P.xa_stride = P.KKp + EPL;
  P.out_stride = 64 + EPL;
  size_t off = (size_t)(P.K * P.V + 1) * sizeof(int);
  off = (off + 15) & ~(size_t)15; P.off_csr_v = (int)off; off += P.nnz_cap;
  off = (off + 15) & ~(size_t)15; P.off_csr_a = (int)off; off += (size_t)P.nnz_cap * 4;
  off = (off + 15) & ~(size_t)15; P.off_stat = (int)off; off += (size_t)2 * MT * 32 * 4;
  off = (off + 15) & ~(size_t)15; P.off_rows = (int)off; off += 2 * TILE_ROWS + 2 * P.K * P.V;
  off = (off + 15) & ~(size_t)15; P.off_afrag = (int)off; off += sizeof(T) == 2 ? (size_t)P.K * 2 * 64 * 16 : 0;
  off = (off + 15) & ~(size_t)15; P.off_work = (int)off;
  size_t work = ((size_t)P.xs_rows * P.xs_stride + (size_t)TILE_ROWS * P.xa_stride) * sizeof(T);
  size_t ost = (size_t)TILE_ROWS * P.out_stride * sizeof(T);
  off += work > ost ? work : ost;
  if (off > 160 * 1024) return ISTGCN_EINVAL;
  if (P.total_tiles < 1) return ISTGCN_OK;
  switch (MT) {
    case 1: return launch_mt<T, 1>(P, grid_x_cap, gy, off, stream);
    case 2: return launch_mt<T, 2>(P, grid_x_cap, gy, off, stream);
    case 4: return launch_mt<T, 4>(P, grid_x_cap, gy, off, stream);
    default: return launch_mt<T, 8>(P, grid_x_cap, gy, off, stream);
  }
}

}  // namespace

extern "C" int istgcn_gcn_fwd_v1(const void* x, const float* A, const void* Wp, const float* bterm,
                              const void* addend, void* y, double* stats, int stats_rep, int* status,
                              int NM, int Tin, int Tout, int Tlog, int V, int Cin, int Cout, int K,
                              int in_t_stride, int out_t_stride, int nnz_cap, int dtype, int grid_cap,
                              void* stream) {
  if (!x || !A || !Wp || !y) return ISTGCN_EINVAL;
  if (NM < 0 || Tlog < 0 || V < 1 || V > 128 || Cin < 1 || Cout < 1 || K < 1 || K > 8) return ISTGCN_EINVAL;
  if (in_t_stride < 1 || out_t_stride < 1 || nnz_cap < 1 || nnz_cap > K * V * V) return ISTGCN_EINVAL;
  if (Tlog > 0 && ((Tlog - 1) * in_t_stride >= Tin || (Tlog - 1) * out_t_stride >= Tout)) return ISTGCN_EINVAL;
  if (stats && stats_rep < 1) return ISTGCN_EINVAL;
  if (NM == 0 || Tlog == 0) return ISTGCN_OK;
  GcnFwdParams P{};
  P.x = x; P.A = A; P.Wp = Wp; P.bterm = bterm; P.addend = addend; P.y = y; P.stats = stats; P.status = status;
  P.NM = NM; P.Tin = Tin; P.Tout = Tout; P.Tlog = Tlog; P.V = V; P.Cin = Cin; P.Cout = Cout; P.K = K;
  P.in_t_stride = in_t_stride; P.out_t_stride = out_t_stride; P.nnz_cap = nnz_cap;
  P.stats_rep = stats_rep < 1 ? 1 : stats_rep;
  if (dtype == 0) return launch_T<float>(P, grid_cap, (hipStream_t)stream);
  if (dtype == 1) return launch_T<__bf16>(P, grid_cap, (hipStream_t)stream);
  if (dtype == 2) return launch_T<_Float16>(P, grid_cap, (hipStream_t)stream);
  return ISTGCN_EINVAL;
}

// Geometry query so the host can size / order the fragment-packed weights exactly as the kernel reads them.
extern "C" int istgcn_gcn_v1_geometry(int Cin, int Cout, int K, int dtype, int* CCeff, int* nch, int* KKp,
                                   int* MTtot, int* EPL) {
  if (!istgcn_dtype_ok(dtype)) return ISTGCN_EINVAL;
  const int epl = dtype == 0 ? 4 : 8, cc = dtype == 0 ? 32 : 64, kgs = 2 * epl;
  int cce = Cin >= cc ? cc : round_up(Cin, epl);
  int MT = Cout <= 32 ? 1 : Cout <= 64 ? 2 : 4;
  *CCeff = cce; *nch = ceil_div(Cin, cce); *KKp = round_up(K * cce, kgs);
  *MTtot = ceil_div(Cout, MT * 32) * MT; *EPL = epl;
  return ISTGCN_OK;
}
