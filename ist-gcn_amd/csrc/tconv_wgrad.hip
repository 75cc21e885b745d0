// Weight gradient of the temporal convolution:   dWf[j][o][i] += sum_{n,m,v} dz[n, m, v, o] * u[n, in_mul*m + tap_off[j], v, i]
// with u = pre(g) (BatchNorm affine + ReLU recomputed on the fly from the GCN output g, frames outside [0,Tin)
// are zero), and the conv-bias gradient dbias[o] += sum dz.  Autograd of net/st_gcnold.py:167-173 (and of the
// pre-summed 15-tap Inception-TCN, st_gcn_multi3_fix_3A_mstcn.py:160-180,212-215).
//
// This is a GEMM whose contraction runs over every position of the batch (millions) and whose output is tiny, so
// the output stays in registers: a workgroup owns an (OT*32) x (IT*32) channel block for ALL taps, each of its 4
// waves keeps the ntaps 32x32 accumulator tiles of one (o-tile, i-tile) pair -- or of one PS-th of the tile's
// positions -- while the workgroup walks position tiles in a grid-stride loop, and everything is flushed ONCE at
// the end with fp32 atomics shaped as two 128-byte row segments per instruction.  A dz fragment is reused across
// all taps; the tap-shifted u fragments come from one staged halo tile.  Operands live in LDS as 32-channel
// sub-tiles [row][32] (64-byte rows for bf16: conflict-free for ds_read_b64_tr_b16, which feeds the MFMA k axis =
// position axis straight from the row-major image); fp32 operand registers are double buffered so LDS latency
// hides behind the MFMAs of the previous k-step.
#include "common.hpp"
#include "tconv_wgrad_lean.hpp"
// Diagnostic hooks (ablation masks whose results are WRONG, in-kernel cycle stamps with their debug buffer) exist only in
// experiment builds (-DISTGCN_EXPERIMENT through tools/build_variant.sh); the shipped library reads no such switch.
#ifdef ISTGCN_EXPERIMENT
#define X_ABL(P) ((P).abl)
#define X_DBG(P) ((P).dbg)
#else
#define X_ABL(P) 0
#define X_DBG(P) ((unsigned long long*)nullptr)
#endif
#include <cstdio>
#include <cstdlib>

#ifdef ISTGCN_STAMP
__device__ unsigned long long g_stamp_wg[16];
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
constexpr int TR = 128;
constexpr int MAX_TAPS = 16;
constexpr int CB = 32;

struct TwgParams {
  const void* dz;        // [NM][Tz][V][Cout]
  const void* g;         // [NM][Tin][V][Cin]
  const float* pre;      // [2][Cin] or null
  float* dW;             // [ntaps][Cout][Cin] fp32, caller-zeroed
  float* dbias;          // [Cout] or null
  // graph-conv mode (AGG): the "taps" are the K adjacency partitions, u_k = sum_v A[k][v][w] x[(t,v)][:]
  const float* A;        // [K][V][V]
  float* S;              // [V][Cout] or null: sum_{n,t} dz[n,t,w,c]
  int nnz_cap, off_csr_v, off_csr_a, off_S, off_afrag, dz_rows, mfma_agg;
  int NM, Tin, Tz, V, Cin, Cout, ntaps, in_mul, pre_relu;
  int tap_off[MAX_TAPS];
  int F, tiles_per_seq, total_tiles, min_off, Fin, n_iblk, urows;
  int capf;              // frames the u region holds (>= Fin): > Fin lets the halo window slide instead of being re-staged
  int off_urow, off_dz, off_u;   // LDS byte offsets
  // optional partial-sum workspace: slice s = [ntaps*Cout*Cin | aux] fp32, one slice per (workgroup.x, position slice);
  // a second kernel sums the slices (no same-address atomic chains).  null: flush with atomics.
  float* ws;
  long long ws_slice;
  int off_dz1, off_x1;   // wave-specialised variants: second halves of the double-buffered dz / x regions
  unsigned long long* dbg;  // diagnostic (ISTGCN_WGRAD_DBG): per-phase cycle sums of workgroup (0,0)
  int abl;               // diagnostic (ISTGCN_WGRAD_ABL): 1 = no tiles (fixed cost of a launch: setup + flush)
};

// sub-tiled staging: vector q of a row goes to sub-tile q / QV.  Split in two so the global loads of the NEXT tile can
// be in flight (in registers) while the matrix cores work on the current one: item it = it0 + tid + u*nthreads.
template <typename T, int U, bool VEC>
__device__ static inline void load_subtiles(typename Elem<T>::frag (&v)[U], const T* __restrict__ g, size_t gstride,
                                            int c_lim, int R, int r_lo, int r_hi, int nsub, int it0, int tid,
                                            int nthreads) {
  using E = Elem<T>;
  constexpr int EPL = E::EPL;
  constexpr int QV = CB / EPL;
  const int Q = nsub * QV;                 // power of two
  const int lq = 31 - __builtin_clz(Q);
  const int tot = R * Q;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int it = it0 + tid + u * nthreads;
    const int r = it >> lq, q = it & (Q - 1);
    const bool live = it < tot && r >= r_lo && r < r_hi && q * EPL < c_lim;
    zero_frag<T>(v[u]);
    if (live) {
      const T* src = g + (size_t)r * gstride + q * EPL;
      if (VEC) v[u] = *reinterpret_cast<const typename E::frag*>(src);
      else {
#pragma unroll
        for (int e = 0; e < EPL; ++e) if (q * EPL + e < c_lim) v[u][e] = src[e];
      }
    }
  }
}

template <typename T, int U, bool VEC>
__device__ static inline void commit_subtiles(typename Elem<T>::frag (&v)[U], int c_lim, T* lds, int sub_elems, int R,
                                              int r_lo, int r_hi, int nsub, const float* __restrict__ sc,
                                              const float* __restrict__ sh, int relu, int it0, int tid, int nthreads) {
  using E = Elem<T>;
  constexpr int EPL = E::EPL;
  constexpr int QV = CB / EPL;
  const int Q = nsub * QV;
  const int lq = 31 - __builtin_clz(Q);
  const int tot = R * Q;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int it = it0 + tid + u * nthreads;
    if (it < tot) {
      const int r = it >> lq, q = it & (Q - 1);
      const bool live = r >= r_lo && r < r_hi && q * EPL < c_lim;
      if (sc && live) {
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
          if (VEC || q * EPL + e < c_lim) {
            float fv = E::to_f(v[u][e]) * sc[q * EPL + e] + sh[q * EPL + e];
            if (relu) fv = fmaxf(fv, 0.f);
            v[u][e] = E::from_f(fv);
          }
        }
      }
      const int sub = q / QV, ql = q - sub * QV;
      *reinterpret_cast<typename E::frag*>(lds + sub * sub_elems + r * CB + ql * EPL) = v[u];
    }
  }
}

// synchronous staging of items [it0, R*Q): the rare tiles larger than the prefetch registers cover
template <typename T, int U, bool VEC>
__device__ static inline void stage_subtiles(const T* __restrict__ g, size_t gstride, int c_lim, T* lds, int sub_elems,
                                             int R, int r_lo, int r_hi, int nsub, const float* __restrict__ sc,
                                             const float* __restrict__ sh, int relu, int it0, int tid, int nthreads) {
  constexpr int QV = CB / Elem<T>::EPL;
  const int tot = R * nsub * QV;
  for (int base = it0; base < tot; base += nthreads * U) {
    typename Elem<T>::frag v[U];
    load_subtiles<T, U, VEC>(v, g, gstride, c_lim, R, r_lo, r_hi, nsub, base, tid, nthreads);
    commit_subtiles<T, U, VEC>(v, c_lim, lds, sub_elems, R, r_lo, r_hi, nsub, sc, sh, relu, base, tid, nthreads);
  }
}

// TS = tap split: the taps of one (o-tile, i-tile, position-slice) are divided over TS waves (workgroup = 4*TS waves), so
// a wave keeps ceil(JT/TS) accumulator tiles: twice the waves per CU at the same LDS footprint for the 9/15-tap layers.
template <typename T, int JT, int OT, int IT, int PS, bool AGG, int TS, bool VEC>
__global__ __launch_bounds__(64 * OT * IT * PS * TS, (AGG && sizeof(T) == 2) ? 4 : 1) void tconv_wgrad_kernel(const TwgParams P) {
  static_assert(OT * IT * PS == 4 || OT * IT * PS == 8, "one (o-tile, i-tile, position-slice) per wave group");
  constexpr int NWG = OT * IT * PS;                // waves per tap group
  constexpr int NTH = 64 * NWG * TS;
  constexpr int JTW = (JT + TS - 1) / TS;
  using E = Elem<T>;
  constexpr int EPL = E::EPL;
  constexpr int NPOS = TR / PS;                  // positions contracted by one wave per tile
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned short* row_f = reinterpret_cast<unsigned short*>(smem);               // [TR]
  unsigned short* row_v = row_f + TR;                                             // [TR]
  unsigned short* urow = reinterpret_cast<unsigned short*>(smem + P.off_urow);    // [TR] u-tile row at tap offset 0
  T* dzs = reinterpret_cast<T*>(smem + P.off_dz);                                 // [OT][TR][CB]
  T* us = reinterpret_cast<T*>(smem + P.off_u);                                   // [IT][urows][CB]
  // AGG only: adjacency column lists (column = k*V + w -> entries (v, a)), S accumulators; x is staged in the dz region
  int* csr_off = reinterpret_cast<int*>(smem + P.off_urow);                       // [K*V+1] (aliases urow: unused in AGG)
  unsigned char* csr_v = smem + P.off_csr_v;
  float* csr_a = reinterpret_cast<float*>(smem + P.off_csr_a);
  float* S_l = reinterpret_cast<float*>(smem + P.off_S);                          // [V][OT*CB + 1]
  constexpr int SLS = OT * CB + 1;                                                 // odd stride: joints land in different banks
  T* afrag = reinterpret_cast<T*>(smem + P.off_afrag);                            // AGG bf16: [K][2][64][8] fragments of A_k

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int V = P.V;
  const int oblk = blockIdx.y / P.n_iblk, iblk = blockIdx.y - oblk * P.n_iblk;
  const int w4 = wave % NWG, ts = wave / NWG;
  const int ot = w4 % OT, it = (w4 / OT) % IT, ps = w4 / (OT * IT);
  const int o0 = oblk * (OT * CB), i0 = iblk * (IT * CB);
  const int u_sub = P.urows * CB;

  for (int r = tid; r < TR; r += NTH) {
    int f = r / V;
    row_f[r] = (unsigned short)f;
    row_v[r] = (unsigned short)(r - f * V);
  }
  if constexpr (AGG) {
    // adjacency -> LDS (coalesced), then per-column compressed lists
    const int K = P.ntaps, KV = K * V;
    float* A_l = reinterpret_cast<float*>(us);
    for (int i = tid; i < K * V * V; i += NTH) A_l[i] = P.A[i];
    for (int c = tid; c <= KV; c += NTH) csr_off[c] = 0;
    for (int c = tid; c < V * SLS; c += NTH) S_l[c] = 0.f;
    __syncthreads();
    for (int col = tid; col < KV; col += NTH) {
      int k = col / V, w = col - k * V, cnt = 0;
      for (int v = 0; v < V; ++v) cnt += (A_l[(k * V + v) * V + w] != 0.f);
      csr_off[col + 1] = cnt;
    }
    __syncthreads();
    if (tid == 0) {
      int run = 0;
      for (int c = 0; c < KV; ++c) { int nn = csr_off[c + 1]; csr_off[c] = run; run += nn; }
      csr_off[KV] = run;
    }
    __syncthreads();
    if (!P.mfma_agg) {                         // VALU aggregation (the MFMA one reads A fragments instead)
      for (int col = tid; col < KV; col += NTH) {
        int k = col / V, w = col - k * V, e = csr_off[col];
        for (int v = 0; v < V; ++v) {
          float a = A_l[(k * V + v) * V + w];
          if (a != 0.f) {
            if (e < P.nnz_cap) { csr_v[e] = (unsigned char)v; csr_a[e] = a; }
            ++e;
          }
        }
      }
    }
    if constexpr (sizeof(T) == 2) {
      if (P.mfma_agg) {
        // fragments of A_k for the MFMA aggregation (lane (w = lane&31, h), k-step s, element j = A[k][16s+8h+j][w])
        // and the zero rows behind every x sub-tile
        for (int idx = tid; idx < K * 2 * 64; idx += NTH) {
          const int ln = idx & 63, sstep = (idx >> 6) & 1, k = idx >> 7;
          const int w = ln & 31, h = ln >> 5;
          typename E::frag fr;
#pragma unroll
          for (int j = 0; j < EPL; ++j) {
            const int v = 16 * sstep + 8 * h + j;
            fr[j] = E::from_f((v < V && w < V) ? A_l[(k * V + v) * V + w] : 0.f);
          }
          *reinterpret_cast<typename E::frag*>(afrag + idx * EPL) = fr;
        }
        constexpr int NSUB = IT > OT ? IT : OT;
        const int ztail = (P.dz_rows - TR) * (CB / EPL);
        for (int idx = tid; idx < NSUB * ztail; idx += NTH) {
          const int sub = idx / ztail, rem = idx - sub * ztail;
          typename E::frag z;
          zero_frag<T>(z);
          *reinterpret_cast<typename E::frag*>(dzs + (sub * P.dz_rows + TR) * CB + rem * EPL) = z;
        }
      }
    }
  }
  __syncthreads();

  f32x16 acc[JTW];
#pragma unroll
  for (int j = 0; j < JTW; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  // column sums of dz (conv bias gradient / S of the graph conv) are taken from the prefetch registers at commit time:
  // a thread's channel vector q is the same for all its rows.  With several i-blocks the vectors are dealt out over them.
  float bs[EPL];
#pragma unroll
  for (int e = 0; e < EPL; ++e) bs[e] = 0.f;
  constexpr int QZ = OT * (CB / EPL);
  const bool aux_any = P.ws ? iblk == 0 : iblk < (TR * QZ + NTH - 1) / NTH;     // i-blocks that own some vector
  auto aux_own = [&](int u) { return P.ws ? iblk == 0 : (u % P.n_iblk == iblk); };

  // per-tap element offset into a u sub-tile (taps beyond ntaps alias tap 0: computed, never flushed)
  int toff[JTW];
#pragma unroll
  for (int jj = 0; jj < JTW; ++jj) {
    const int j = ts * JTW + jj;                     // this wave's taps; padding taps alias tap 0 (computed, never flushed)
    const int jv = j < P.ntaps ? j : 0;
    toff[jj] = AGG ? jv * TR * CB : (P.tap_off[jv] - P.min_off) * V * CB;
  }

  const T* dzg = reinterpret_cast<const T*>(P.dz);
  const T* gg = reinterpret_cast<const T*>(P.g);
  const int dz_sub = P.dz_rows * CB;               // sub-tile stride of the dz / x staging region
  const T* dz_w = dzs + ot * dz_sub;
  typedef typename E::frag frag_t;
  constexpr int QV = CB / EPL;

#ifdef ISTGCN_STAMP
  unsigned long long st_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_prev = __builtin_amdgcn_s_memtime();
#endif
  // ---- register prefetch of the next tile's global data (issued before the MFMA loop, consumed after it) ----
  constexpr int UZ = (TR * OT * QV + NTH - 1) / NTH;      // dz tile vectors per thread (exact cover)
  constexpr int UX = (TR * IT * QV + NTH - 1) / NTH;      // AGG: x tile vectors per thread
  constexpr int UH = (264 * IT * QV + NTH - 1) / NTH;     // u rows fetched per tile (new frames of the sliding window) ...
  constexpr int UU = AGG ? UX : (UH < 8 ? UH : 8);        // ... prefetched; anything beyond finishes synchronously
  frag_t pz[UZ], pu[UU];
  // A workgroup walks CONSECUTIVE tiles (frames m0, m0+F, ... of one sequence, then the next sequence): the halo'd u
  // window of a tile overlaps its predecessor's in Fin - in_mul*F frames, which stay in LDS; only the new frames are
  // fetched and transformed.  The window slides through a region of capf frames and is moved back to the front when it
  // reaches the end.  A tile is "fresh" (whole window staged) at a sequence start or when the region has no slack.
  const int chunk = (P.total_tiles + gridDim.x - 1) / gridDim.x;
  const int t_begin = blockIdx.x * chunk, t_end = min(P.total_tiles, t_begin + chunk);
  const int adv = P.in_mul * P.F, keep = P.Fin - adv;
  const bool slide = !AGG && keep > 0 && P.capf >= P.Fin + adv;
  const int keep_items = keep * V * IT * QV;
  auto is_fresh = [&](int tile) { return !slide || tile == t_begin || tile % P.tiles_per_seq == 0; };
  auto prefetch = [&](int tile) {
    const int n = tile / P.tiles_per_seq;
    const int m0 = (tile - n * P.tiles_per_seq) * P.F;
    const int nf = min(P.F, P.Tz - m0);
    const int rows = nf * V;
    const size_t pos0 = (size_t)(n * P.Tz + m0) * V;
    {
      const T* src = dzg + pos0 * P.Cout + o0;
      // (dz rows are whole 16-byte vectors whenever Cout is a multiple of the vector width, whatever Cin is: the 3-channel
      //  first layer loaded its 64-channel dz element by element under the kernel-wide VEC flag)
      if (VEC || (P.Cout % Elem<T>::EPL) == 0) load_subtiles<T, UZ, true>(pz, src, (size_t)P.Cout, P.Cout - o0, TR, 0, rows, OT, 0, tid, NTH);
      else load_subtiles<T, UZ, false>(pz, src, (size_t)P.Cout, P.Cout - o0, TR, 0, rows, OT, 0, tid, NTH);
    }
    if constexpr (AGG) {
      const T* src = gg + pos0 * P.Cin + i0;
      load_subtiles<T, UU, VEC>(pu, src, (size_t)P.Cin, P.Cin - i0, TR, 0, rows, IT, 0, tid, NTH);
    } else {
      const int fin0 = P.in_mul * m0 + P.min_off;
      const int in_rows = (P.in_mul * (nf - 1) + P.Fin - P.in_mul * (P.F - 1)) * V;
      const long long row0 = (long long)(n * P.Tin + fin0) * V;
      const int r_lo = fin0 < 0 ? -fin0 * V : 0;
      const int r_hi = min(in_rows, (P.Tin - fin0) * V);
      const T* src = gg + row0 * P.Cin + i0;
      load_subtiles<T, UU, VEC>(pu, src, (size_t)P.Cin, P.Cin - i0, in_rows, r_lo, r_hi, IT, is_fresh(tile) ? 0 : keep_items,
                                tid, NTH);
    }
  };
  if (t_begin < t_end) prefetch(t_begin);
  int wslot = 0;                                          // frame slot of the current window in the u region
  const T* us_w = us + it * u_sub;

  for (int tile = t_begin; tile < t_end; ++tile) {
    STAMP(0);
    const int n = tile / P.tiles_per_seq;
    const int m0 = (tile - n * P.tiles_per_seq) * P.F;
    const int nf = min(P.F, P.Tz - m0);
    const int rows = nf * V;
    const int fin0 = P.in_mul * m0 + P.min_off;
    const int in_rows = (P.in_mul * (nf - 1) + P.Fin - P.in_mul * (P.F - 1)) * V;
    const int next_tile = tile + 1;
#ifdef ISTGCN_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    STAMP(7);
#endif

    if constexpr (AGG) {
      // ---- x tile -> dz region, K aggregated images -> us, then the dz tile over the x tile ----
      commit_subtiles<T, UU, VEC>(pu, P.Cin - i0, dzs, dz_sub, TR, 0, rows, IT, nullptr, nullptr, 0, 0, tid, NTH);
      STAMP(8);
      __syncthreads();
      STAMP(9);
      const int K = P.ntaps, KV = K * V;
      {
        bool agg_done = false;
        if constexpr (sizeof(T) == 2) if (P.mfma_agg) {
          // bf16: D[i][w] = sum_v x[(f,v)][i] * A_k[v][w] on the matrix cores, x^T via ds_read_b64_tr_b16 (as gcn_fwd)
          agg_done = true;
          const int grp = lane >> 4, hh = grp >> 1, cblk = (grp & 1) * 16;
          const int q4 = (lane & 15) >> 2, pp = lane & 3;
          const int w = lane & 31;
          for (int unit = wave; unit < IT * nf * K; unit += NWG * TS) {     // (sub-tile, frame, partition) units
            const int k = unit % K, pr = unit / K;
            const int sub = pr / nf, f = pr - sub * nf;
            frag_t a[2];
#pragma unroll
            for (int sstep = 0; sstep < 2; ++sstep) {
              const T* r0 = dzs + (sub * P.dz_rows + f * V + 16 * sstep + 8 * hh + q4) * CB + cblk + 4 * pp;
              a[sstep] = tr_pair<T>(r0, r0 + 4 * CB);
            }
            const frag_t b0 = *reinterpret_cast<const frag_t*>(afrag + ((k * 2 + 0) * 64 + lane) * EPL);
            const frag_t b1 = *reinterpret_cast<const frag_t*>(afrag + ((k * 2 + 1) * 64 + lane) * EPL);
            f32x16 d;
#pragma unroll
            for (int r = 0; r < 16; ++r) d[r] = 0.f;
            mma_kgroup(d, a[0], b0);
            mma_kgroup(d, a[1], b1);
            if (w < V) {
              // a lane owns one image row; the rows are 64 bytes apart (what the transposed reads of the contraction
              // want), so sixteen lanes storing the same 16-byte block of their rows would share two banks.  The block
              // index is XOR-swizzled with bits 1-2 of the row (the contraction's reads apply the same map): the SQ
              // counters of the unswizzled kernel had 60 % of its LDS cycles in bank conflicts.
              const int row = f * V + w, sw = (row >> 1) & 3;
#pragma unroll
              for (int g = 0; g < 4; ++g) {
                float v4[4] = {d[4 * g], d[4 * g + 1], d[4 * g + 2], d[4 * g + 3]};
                store4(us + ((sub * K + k) * TR + row) * CB + 8 * (g ^ sw) + 4 * (lane >> 5), v4);
              }
            }
          }
        }
        if (!agg_done) {
        // wave w owns adjacency columns col = w, w+4, ...; lanes span (sub-tile, frame, channel vector)
        const int npair = IT * nf * QV;
        for (int col = wave; col < KV; col += NWG * TS) {
          const int k = col / V, w = col - k * V;
          const int e0 = csr_off[col], e1 = min(csr_off[col + 1], P.nnz_cap);
          for (int pr = lane; pr < npair; pr += 64) {
            const int q = pr % QV;
            const int f = (pr / QV) % nf;
            const int sub = pr / (QV * nf);
            const T* xrow = dzs + sub * dz_sub + (f * V) * CB + q * EPL;
            float sum[EPL];
#pragma unroll
            for (int e = 0; e < EPL; ++e) sum[e] = 0.f;
            for (int en = e0; en < e1; ++en) {
              const float a = csr_a[en];
              const frag_t xv = *reinterpret_cast<const frag_t*>(xrow + csr_v[en] * CB);
#pragma unroll
              for (int e = 0; e < EPL; ++e) sum[e] += a * E::to_f(xv[e]);
            }
            frag_t o;
#pragma unroll
            for (int e = 0; e < EPL; ++e) o[e] = E::from_f(sum[e]);
            const int qs = sizeof(T) == 2 ? (q ^ (((f * V + w) >> 1) & 3)) : q;      // 16-bit: swizzled block (see above)
            *reinterpret_cast<frag_t*>(us + ((sub * K + k) * TR + f * V + w) * CB + qs * EPL) = o;
          }
        }
        }
        // positions >= rows are contracted too (against zero dz rows): keep their image rows finite (zero)
        const int padr = TR - rows;
        for (int idx = tid; idx < IT * K * padr * QV; idx += NTH) {
          const int q = idx % QV;
          const int r = rows + (idx / QV) % padr;
          const int sk = idx / (QV * padr);
          frag_t o;
          zero_frag<T>(o);
          *reinterpret_cast<frag_t*>(us + (sk * TR + r) * CB + q * EPL) = o;
        }
      }
      STAMP(10);
      __syncthreads();
      STAMP(11);
      commit_subtiles<T, UZ, VEC>(pz, P.Cout - o0, dzs, dz_sub, TR, 0, rows, OT, nullptr, nullptr, 0, 0, tid, NTH);
      __syncthreads();
      STAMP(12);
      if (next_tile < t_end) prefetch(next_tile);
      STAMP(13);
      if (P.S) {
        // S[w][c] += sum_f dz[(f,w)][c]: column sums of the staged tile, (w,c) cells dealt out over the i-blocks;
        // the frame reads are issued together (a serial chain of LDS latencies costs more than the MFMA loop)
        constexpr int NC = OT * CB;
        const int n_own = P.ws ? 1 : P.n_iblk, me = P.ws ? 0 : iblk;
        if (!P.ws || iblk == 0) {
          for (int idx = tid + me * NTH; idx < V * NC; idx += NTH * n_own) {
            const int w = idx / NC, c = idx - w * NC;
            const T* col = dzs + (c / CB) * dz_sub + (c % CB) + w * CB;
            float sacc = 0.f;
            for (int f0 = 0; f0 < nf; f0 += 8) {
              float v[8];
#pragma unroll
              for (int j = 0; j < 8; ++j) v[j] = f0 + j < nf ? E::to_f(col[(f0 + j) * V * CB]) : 0.f;
#pragma unroll
              for (int j = 0; j < 8; ++j) sacc += v[j];
            }
            S_l[w * SLS + c] += sacc;
          }
        }
      }
    } else {
    // ---- dz tile (zero pad rows) and the u tile with halo: pre(g), zero outside the sequence ----
      commit_subtiles<T, UZ, VEC>(pz, P.Cout - o0, dzs, dz_sub, TR, 0, rows, OT, nullptr, nullptr, 0, 0, tid, NTH);
      if (P.dbias) {
#pragma unroll
        for (int u = 0; u < UZ; ++u) {
          const int item = tid + u * NTH;
          if (aux_own(u) && item / QZ < rows) {
#pragma unroll
            for (int e = 0; e < EPL; ++e) bs[e] += E::to_f(pz[u][e]);
          }
        }
      }
      STAMP(8);
      for (int r = tid; r < TR; r += NTH)
        urow[r] = r < rows ? (unsigned short)((P.in_mul * row_f[r]) * V + row_v[r]) : (unsigned short)0;
      STAMP(9);
      {
        const bool fresh = is_fresh(tile);
        if (fresh) wslot = 0;
        else {
          wslot += adv;
          if (wslot + P.Fin > P.capf) {                   // window at the end of the region: overlap back to the front
            // The kept frames move DOWN by wslot frames; source and destination ranges overlap whenever more frames are kept
            // than the window advanced by, and the rows behind the kept ones are where this tile's new frames go.  All reads
            // first, a barrier, then the writes: without it a wave that arrived early overwrote rows a late wave had not read
            // yet (round 4: tap 0 of the 15-tap stride-2 layers wrong by < 1 % of max |dW| in a third of the runs at 256
            // channels, tools/twg_flaky.py; the waves do not enter this loop together).
            constexpr int KC = 8;                           // vectors per thread held across the barrier, per pass
            for (int base = 0; base < keep_items; base += KC * NTH) {
              frag_t tmp[KC];
#pragma unroll
              for (int u = 0; u < KC; ++u) {
                const int idx = base + tid + u * NTH;
                if (idx < keep_items) {
                  const int q = idx % (IT * QV), r = idx / (IT * QV);
                  const int sub = q / QV, ql = q - sub * QV;
                  tmp[u] = *reinterpret_cast<const frag_t*>(us + sub * u_sub + r * CB + ql * EPL + wslot * V * CB);
                }
              }
              __syncthreads();
#pragma unroll
              for (int u = 0; u < KC; ++u) {
                const int idx = base + tid + u * NTH;
                if (idx < keep_items) {
                  const int q = idx % (IT * QV), r = idx / (IT * QV);
                  const int sub = q / QV, ql = q - sub * QV;
                  *reinterpret_cast<frag_t*>(us + sub * u_sub + r * CB + ql * EPL) = tmp[u];
                }
              }
              if (base + KC * NTH < keep_items) __syncthreads();     // (another pass: its reads come after these writes)
            }
            wslot = 0;
          }
        }
        T* uwin = us + wslot * V * CB;
        us_w = uwin + it * u_sub;
        const int it0 = fresh ? 0 : keep_items;
        const long long row0 = (long long)(n * P.Tin + fin0) * V;
        const int r_lo = fin0 < 0 ? -fin0 * V : 0;
        const int r_hi = min(in_rows, (P.Tin - fin0) * V);
        const float* sc = P.pre ? P.pre + i0 : nullptr;
        const float* sh = P.pre ? P.pre + P.Cin + i0 : nullptr;
        const T* src = gg + row0 * P.Cin + i0;
        commit_subtiles<T, UU, VEC>(pu, P.Cin - i0, uwin, u_sub, in_rows, r_lo, r_hi, IT, sc, sh, P.pre_relu, it0, tid, NTH);
        if (in_rows * IT * QV > it0 + UU * NTH) {          // more than the prefetch registers cover (fresh windows)
          stage_subtiles<T, 4, VEC>(src, (size_t)P.Cin, P.Cin - i0, uwin, u_sub, in_rows, r_lo, r_hi, IT, sc, sh, P.pre_relu, it0 + UU * NTH, tid, NTH);
        }
      }
      STAMP(10);
      __syncthreads();
      STAMP(11);
      if (next_tile < t_end) prefetch(next_tile);
    }
    STAMP(1);

    STAMP(2);
    // ---- D_j[o][i] += dz[p][o] * u[row(p) + tap_j][i] over this wave's positions ----
    const int pbase = ps * NPOS;
    if constexpr (sizeof(T) == 4) {
      constexpr int NK = NPOS / 2;
      const int r = lane & 31, h = lane >> 5;
      float a0, a1, b0[JTW], b1[JTW];
      auto load_k = [&](int kk, float& a, float (&b)[JTW]) {
        const int p = pbase + 2 * kk + h;
        a = dz_w[p * CB + r];
        const T* ub = us_w + (AGG ? p : (int)urow[p]) * CB + r;
#pragma unroll
        for (int j = 0; j < JTW; ++j) b[j] = ub[toff[j]];
      };
      auto mma_k = [&](float a, const float (&b)[JTW]) {
#pragma unroll
        for (int j = 0; j < JTW; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[j], acc[j], 0, 0, 0);
      };
      if constexpr (JTW > 4) {
        for (int kk = 0; kk < NK; ++kk) {
          load_k(kk, a0, b0);
          mma_k(a0, b0);
        }
      } else {
        load_k(0, a0, b0);
        for (int kk = 0; kk < NK; kk += 2) {
          load_k(kk + 1, a1, b1);
          __builtin_amdgcn_sched_barrier(0);
          mma_k(a0, b0);
          __builtin_amdgcn_sched_barrier(0);
          load_k(min(kk + 2, NK - 1), a0, b0);
          __builtin_amdgcn_sched_barrier(0);
          mma_k(a1, b1);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    } else {
      constexpr int NK = NPOS / 16;
      const int grp = lane >> 4, h = grp >> 1, cblk = (grp & 1) * 16;
      const int q = (lane & 15) >> 2, pp = lane & 3;
      const int coff = cblk + 4 * pp;
      // AGG: the aggregated images are block-swizzled by row bits 1-2 (see the aggregation); a lane's rows are
      // 8h + q and 8h + q + 4 (mod 16) in every k-step, so its two swizzled column offsets are constants
      const int coff_u0 = (((coff >> 3) ^ (((8 * h + q) >> 1) & 3)) << 3) + (coff & 7);
      const int coff_u1 = (((coff >> 3) ^ (((8 * h + q + 4) >> 1) & 3)) << 3) + (coff & 7);
      frag_t a0, a1, b0[JTW], b1[JTW];
      auto load_k = [&](int kk, frag_t& a, frag_t (&b)[JTW]) {
        const int pb = pbase + 16 * kk + 8 * h + q;          // this lane addresses rows pb and pb+4 of its 8 positions
        a = tr_pair<T>(dz_w + pb * CB + coff, dz_w + (pb + 4) * CB + coff);
        const T* u0 = us_w + (AGG ? pb * CB + coff_u0 : (int)urow[pb] * CB + coff);
        const T* u1 = us_w + (AGG ? (pb + 4) * CB + coff_u1 : (int)urow[pb + 4] * CB + coff);
#pragma unroll
        for (int j = 0; j < JTW; ++j) b[j] = tr_pair<T>(u0 + toff[j], u1 + toff[j]);
      };
      auto mma_k = [&](const frag_t& a, const frag_t (&b)[JTW]) {
#pragma unroll
        for (int j = 0; j < JTW; ++j) mma_kgroup(acc[j], a, b[j]);
      };
      if constexpr (NK == 1 || JTW > 4 || AGG) {
#pragma unroll
        for (int kk = 0; kk < NK; ++kk) {
          load_k(kk, a0, b0);
          mma_k(a0, b0);
        }
      } else {
        load_k(0, a0, b0);
#pragma unroll
        for (int kk = 0; kk < NK; kk += 2) {
          load_k(kk + 1, a1, b1);
          __builtin_amdgcn_sched_barrier(0);
          mma_k(a0, b0);
          __builtin_amdgcn_sched_barrier(0);
          load_k(kk + 2 < NK ? kk + 2 : NK - 1, a0, b0);
          __builtin_amdgcn_sched_barrier(0);
          mma_k(a1, b1);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    STAMP(3);
    __syncthreads();
    STAMP(4);
  }
  STAMP(5);
#ifdef ISTGCN_STAMP
  const unsigned long long t_flush0 = __builtin_amdgcn_s_memtime();
#endif

  float* red = reinterpret_cast<float*>(dzs);             // tile loop is over: reuse the dz region
  if (!AGG && P.dbias && aux_any) {
    constexpr int NC = OT * CB;
    if (tid < NC) red[tid] = 0.f;
    __syncthreads();
#pragma unroll
    for (int e = 0; e < EPL; ++e) atomicAdd(red + (tid % QZ) * EPL + e, bs[e]);
    __syncthreads();
  }

  // ---- flush: D tile rows = o (registers), cols = i (lanes): two 128-byte segments per instruction ----
  if (P.ws) {
    float* sl = P.ws + (size_t)(blockIdx.x * PS + ps) * P.ws_slice;
#pragma unroll
    for (int jj = 0; jj < JTW; ++jj) {
      const int j = ts * JTW + jj;
      if (j < P.ntaps) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int o = o0 + ot * CB + mfma_row(r, lane), i = i0 + it * CB + (lane & 31);
          if (o < P.Cout && i < P.Cin) sl[((size_t)j * P.Cout + o) * P.Cin + i] = acc[jj][r];
        }
      }
    }
    if (iblk == 0) {
      constexpr int NC = OT * CB;
      float* aux = P.ws + (size_t)(blockIdx.x * PS) * P.ws_slice + (size_t)P.ntaps * P.Cout * P.Cin;
      if (AGG && P.S) {
        for (int idx = tid; idx < V * NC; idx += NTH) {
          const int w = idx / NC, c = idx - w * NC;
          if (o0 + c < P.Cout) {
            aux[w * P.Cout + o0 + c] = S_l[w * SLS + c];
#pragma unroll
            for (int p = 1; p < PS; ++p) aux[p * P.ws_slice + w * P.Cout + o0 + c] = 0.f;
          }
        }
      }
      if (!AGG && P.dbias) {
        if (tid < NC && o0 + tid < P.Cout) {
          aux[o0 + tid] = red[tid];
#pragma unroll
          for (int p = 1; p < PS; ++p) aux[p * P.ws_slice + o0 + tid] = 0.f;
        }
      }
    }
  } else {
#pragma unroll
    for (int jj = 0; jj < JTW; ++jj) {
      const int j = ts * JTW + jj;
      if (j < P.ntaps) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int o = o0 + ot * CB + mfma_row(r, lane), i = i0 + it * CB + (lane & 31);
          if (o < P.Cout && i < P.Cin) atomicAdd(P.dW + ((size_t)j * P.Cout + o) * P.Cin + i, acc[jj][r]);
        }
      }
    }
    if (AGG && P.S) {                                    // every i-block owns some (w,c) cells (zeros otherwise)
      constexpr int NC = OT * CB;
      for (int idx = tid; idx < V * NC; idx += NTH) {
        const int w = idx / NC, c = idx - w * NC;
        if (o0 + c < P.Cout) atomicAdd(P.S + w * P.Cout + o0 + c, S_l[w * SLS + c]);
      }
    }
    if (!AGG && P.dbias && aux_any) {
      constexpr int NC = OT * CB;
      if (tid < NC && o0 + tid < P.Cout) atomicAdd(P.dbias + o0 + tid, red[tid]);
    }
  }
#ifdef ISTGCN_STAMP
  if (lane == 0) {
    st_acc[6] += __builtin_amdgcn_s_memtime() - t_flush0;
    for (int i = 0; i < 16; ++i) atomicAdd(&g_stamp_wg[i], st_acc[i]);
  }
#endif
}

// dst[e] += sum over slices [s0, s1) of ws[s*slice + e]; float4 per thread, slices split over blockIdx.y
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ ws, long long slice, int nsl,
                                                           float* __restrict__ d0, int n0, float* __restrict__ d1,
                                                           int n1) {
  const int e = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (e >= n0 + n1) return;
  const int per = (nsl + gridDim.y - 1) / gridDim.y;
  const int s0 = blockIdx.y * per, s1 = min(nsl, s0 + per);
  float a[4] = {0.f, 0.f, 0.f, 0.f};
  const bool v4 = (slice % 4 == 0) && (n0 % 4 == 0) && (e + 3 < n0 + n1);
  if (v4) {
    const float* src = ws + e;
#pragma unroll 4
    for (int s = s0; s < s1; ++s) {
      const float4 v = *reinterpret_cast<const float4*>(src + (size_t)s * slice);
      a[0] += v.x; a[1] += v.y; a[2] += v.z; a[3] += v.w;
    }
  } else {
    for (int s = s0; s < s1; ++s)
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (e + k < n0 + n1) a[k] += ws[(size_t)s * slice + e + k];
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int ek = e + k;
    if (ek < n0) atomicAdd(d0 + ek, a[k]);
    else if (ek < n0 + n1 && d1) atomicAdd(d1 + (ek - n0), a[k]);
  }
}

// ======================================================================================================================
// Wave-specialised variant (round 2) for the shapes that carry the bench: 16-bit storage, stride 1, whole 16-byte channel
// vectors, a 64 x 64 channel block per workgroup, up to 9 taps.  ONE 8-wave workgroup per CU:
//   waves 0-3 "compute": one (o-tile, i-tile) pair each, all JT accumulator tiles in registers.  Stride 1 puts position
//                        p at row p of the halo window, so every operand address of a tile is one per-lane base per tap
//                        plus a compile-time offset: the loop is 72 MFMAs + 160 transposed LDS reads + ~35 others.
//   waves 4-7 "memory":  dz tile (double-buffered) and the NEW frames of the halo window: BatchNorm affine + ReLU on the
//                        way into LDS, conv-bias column sums on the way.
// Budget rule that shaped it: a wave retires about one instruction per 4-5 cycles whatever its SIMD partner does, so a
// tile takes max over waves of (instructions x ~4.5).  Re-staging the whole 13-frame halo per 5-frame tile put ~1000
// instructions per tile on each memory wave (measured: 8000 cycles per tile against 2304 of MFMA work, no faster than the
// round-1 kernel).  Hence the SLIDING window: the halo of consecutive tiles overlaps in Fin - F frames that stay in LDS,
// only the F new frames are fetched and transformed, into a region of capf = Fin + n*F frames that the window slides
// through; when it reaches the end (or a new sequence starts) the tile is FRESH: staged whole at the front.  While the
// compute waves work on tile k the memory waves stage tile k+1 -- behind window k for a slide; at the front for a fresh
// tile, which is disjoint from window k when that sits at frame >= Fin (always the case at an overflow), else the fresh
// tile is staged after the barrier (one extra barrier; once per sequence).
// One raw s_barrier per tile.  Loads of tile k+2 are issued after the commit of tile k+1 (so the commit's wait is simply
// "everything outstanding" and the number of loads per tile may vary).
// ======================================================================================================================
// cache policy of the wave-specialised kernel's global loads (experiment builds: -DTWG_NT_U / -DTWG_NT_DZ = streaming loads)
#ifdef TWG_NT_U
#define TWG_LD_U(p) __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p))
#else
#define TWG_LD_U(p) (*reinterpret_cast<const u32x4*>(p))
#endif
#ifdef TWG_NT_DZ
#define TWG_LD_DZ(p) __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p))
#else
#define TWG_LD_DZ(p) (*reinterpret_cast<const u32x4*>(p))
#endif
// packed-pair helpers of the memory role's BatchNorm + ReLU transform (see tconv_lean.hip)
template <typename T> __device__ static inline void twg_unpack2(uint32_t p, float& lo, float& hi);
template <> __device__ inline void twg_unpack2<__bf16>(uint32_t p, float& lo, float& hi) {
  lo = __builtin_bit_cast(float, p << 16);
  hi = __builtin_bit_cast(float, p & 0xffff0000u);
}
template <> __device__ inline void twg_unpack2<_Float16>(uint32_t p, float& lo, float& hi) {
  const f16x2 v = __builtin_bit_cast(f16x2, p);
  lo = (float)v[0];
  hi = (float)v[1];
}
template <typename T> __device__ static inline uint32_t twg_pk2(float a, float b);
template <> __device__ inline uint32_t twg_pk2<__bf16>(float a, float b) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const f32x2 v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}
template <> __device__ inline uint32_t twg_pk2<_Float16>(float a, float b) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const f32x2 v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2));
}
__device__ static inline uint32_t twg_relu_pk(uint32_t w) {
  typedef short s16x2 __attribute__((ext_vector_type(2)));
  const s16x2 z = {0, 0};
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, w), z));
}
constexpr int WS_NROLE = 256;
constexpr int WS_NTH = 2 * WS_NROLE;
constexpr int WS_UZ = 4;              // dz vectors per memory thread and tile: 128 rows x 8 vectors / 256
constexpr int WS_UF = 11;             // u vectors per memory thread of a FRESH tile: Fin*V rows x 8 vectors <= 11 * 256
constexpr int WS_US = 4;              // ... of a SLIDE tile: F*V rows x 8 vectors <= 4 * 256

__device__ static inline void ws_barrier() {
  // LDS traffic of this wave retired, then the workgroup barrier; NOT __syncthreads() (its fence drains the prefetches)
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <typename T, int JT>
__global__ __launch_bounds__(WS_NTH, 2) void twg_ws_kernel(const TwgParams P) {
  using E = Elem<T>;
  constexpr int EPL = E::EPL;
  static_assert(EPL == 8, "16-bit storage only");
  typedef typename E::frag frag_t;
  constexpr int RB = CB * (int)sizeof(T);                  // bytes per sub-tile row (64)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* red = reinterpret_cast<float*>(smem + P.off_S);                          // [64] conv-bias column sums
  const int tid = (int)(threadIdx.x ^ ISTGCN_ROLE_FLIP), lane = tid & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool is_compute = wave8 < 4;
  const int ltid = tid & (WS_NROLE - 1);
  const int V = P.V;
  const int oblk = blockIdx.y / P.n_iblk, iblk = blockIdx.y - oblk * P.n_iblk;
  const int o0 = oblk * 64, i0 = iblk * 64;
  const int dz_sub = TR * CB, u_sub = P.urows * CB;        // elements per 32-channel sub-tile

  if (tid < 64) red[tid] = 0.f;
  {
    // both dz halves and the whole u region start as zeros: everything the contraction can reach is finite from the
    // first tile on (pad positions behind a tile's frames meet zero dz rows and whatever finite u rows lie there)
    frag_t z;
    zero_frag<T>(z);
    const int n0 = 2 * 2 * dz_sub / EPL, n1 = 2 * u_sub / EPL;
    T* d0 = reinterpret_cast<T*>(smem + P.off_dz);
    T* u0 = reinterpret_cast<T*>(smem + P.off_u);
    for (int i = tid; i < n0; i += WS_NTH) *reinterpret_cast<frag_t*>(d0 + i * EPL) = z;
    for (int i = tid; i < n1; i += WS_NTH) *reinterpret_cast<frag_t*>(u0 + i * EPL) = z;
  }
  __syncthreads();

  const T* dzg = reinterpret_cast<const T*>(P.dz);
  const T* gg = reinterpret_cast<const T*>(P.g);
  const int chunk = (P.total_tiles + gridDim.x - 1) / gridDim.x;
  const int t_begin = blockIdx.x * chunk, t_end = min(P.total_tiles, t_begin + chunk);
  const int ntile = (t_end > t_begin && !(X_ABL(P) & 1)) ? t_end - t_begin : 0;
  const int adv = P.F, keep = P.Fin - adv;                 // frames a window advances by / shares with its predecessor
  // window schedule, computed identically by both roles: tile k is FRESH (window at frame 0, staged whole) at the start of
  // the walk, at a sequence start, or when sliding on would leave the region; otherwise its window is adv frames further on
  // (tiles of a workgroup are consecutive: (sequence n, tile-in-sequence mq) advance by increments, no divisions per tile)
  struct TPos { int n, mq; };
  auto tpos_first = [&]() __attribute__((always_inline)) { TPos c; c.n = t_begin / P.tiles_per_seq; c.mq = t_begin - c.n * P.tiles_per_seq; return c; };
  auto tpos_next = [&](TPos c) __attribute__((always_inline)) { if (++c.mq == P.tiles_per_seq) { c.mq = 0; ++c.n; } return c; };
  auto next_window = [&](int k, const TPos& c, int w_prev, bool& fresh) __attribute__((always_inline)) {
    fresh = k == 0 || c.mq == 0 || w_prev + adv + P.Fin > P.capf;
    return fresh ? 0 : w_prev + adv;
  };

  f32x16 acc[JT];
  unsigned long long tacc[4] = {0, 0, 0, 0}, tlast = 0;
#define WSTAMP(i) if (X_DBG(P)) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); tacc[i] += now_ - tlast; tlast = now_; }
#ifdef ISTGCN_X_PRIO        /* experiment build: issue priority per role (1: compute waves high, 2: memory waves high) */
  if ((ISTGCN_X_PRIO == 1) == is_compute) __builtin_amdgcn_s_setprio(3);
#endif
  if (is_compute) {
    // =========================================== compute waves ===========================================
    const int ot = wave8 & 1, it = wave8 >> 1;
#pragma unroll
    for (int j = 0; j < JT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    int toff[JT];                                           // byte offset of a tap's rows in a u sub-tile
#pragma unroll
    for (int j = 0; j < JT; ++j) {
      const int jv = j < P.ntaps ? j : 0;                   // padding taps alias tap 0 (computed, never flushed)
      toff[j] = (P.tap_off[jv] - P.min_off) * V * RB;
    }
    const int grp = lane >> 4, h = grp >> 1, cblk = (grp & 1) * 16;
    const int q = (lane & 15) >> 2, pp = lane & 3;
    const int coff = cblk + 4 * pp;
    const int lrow = 8 * h + q;                             // this lane addresses rows 16*ks + lrow and + 4 of every k-step
    ws_barrier();                                           // tile 0 staged (the memory waves' prologue)
    tlast = __builtin_amdgcn_s_memtime();
    int w = 0;
    bool fresh = true;
    TPos cpos = tpos_first();
    for (int k = 0; k < ntile; ++k) {
      w = next_window(k, cpos, w, fresh);
      cpos = tpos_next(cpos);
      // byte offsets in LDS, never pointers selected at run time (those decay to flat loads)
      const int dzb = ((k & 1) ? P.off_dz1 : P.off_dz) + (ot * dz_sub + coff) * (int)sizeof(T);
      const int ub = P.off_u + (it * u_sub + coff) * (int)sizeof(T) + w * V * RB;
      constexpr int NK = TR / 16;
      const unsigned char* ap = smem + dzb + lrow * RB;
      const unsigned char* up[JT];
#pragma unroll
      for (int j = 0; j < JT; ++j) up[j] = smem + ub + lrow * RB + toff[j];
      frag_t a0, a1, b0[JT], b1[JT];
      auto load_k = [&](int ks, frag_t& a, frag_t (&b)[JT]) __attribute__((always_inline)) {
        a = tr_pair<T>(reinterpret_cast<const T*>(ap + ks * 16 * RB), reinterpret_cast<const T*>(ap + ks * 16 * RB + 4 * RB));
#pragma unroll
        for (int j = 0; j < JT; ++j)
          b[j] = tr_pair<T>(reinterpret_cast<const T*>(up[j] + ks * 16 * RB), reinterpret_cast<const T*>(up[j] + ks * 16 * RB + 4 * RB));
      };
      auto mma_k = [&](const frag_t& a, const frag_t (&b)[JT]) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < JT; ++j) mma_kgroup(acc[j], a, b[j]);
      };
      load_k(0, a0, b0);
#pragma unroll
      for (int ks = 0; ks < NK; ks += 2) {
        load_k(ks + 1, a1, b1);
        __builtin_amdgcn_sched_barrier(0);
        mma_k(a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        if (ks + 2 < NK) load_k(ks + 2, a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        mma_k(a1, b1);
        __builtin_amdgcn_sched_barrier(0);
      }
      WSTAMP(0)
      ws_barrier();                                         // tile k contracted, tile k+1 staged (unless it is a late fresh one)
      if (k + 1 < ntile) {
        bool f1;
        next_window(k + 1, cpos, w, f1);                    // (cpos is tile k+1 by now)
        if (f1 && w < P.Fin) ws_barrier();                  // fresh tile whose front window overlaps window k: staged now
      }
      WSTAMP(1)
    }
    if (X_DBG(P) && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) { X_DBG(P)[0] = tacc[0]; X_DBG(P)[1] = tacc[1]; X_DBG(P)[7] = (unsigned long long)ntile; }
  } else {
    // =========================================== memory waves ============================================
    const int q = ltid & 7;                                 // this thread's channel vector of a 64-channel row (both tensors)
    const int sub = q >> 2, ql = q & 3;
    const bool zlive_q = o0 + q * EPL < P.Cout, ulive_q = i0 + q * EPL < P.Cin;
    float scv[EPL], shv[EPL], bs[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
      scv[e] = (P.pre && ulive_q) ? P.pre[i0 + q * EPL + e] : 1.f;
      shv[e] = (P.pre && ulive_q) ? P.pre[P.Cin + i0 + q * EPL + e] : 0.f;
      bs[e] = 0.f;
    }
    // per-thread constants: row r of slot u never changes; element offsets of that row in the two tensors
    unsigned zoff[WS_UZ], uoff[WS_UF];
#pragma unroll
    for (int u = 0; u < WS_UZ; ++u) zoff[u] = (unsigned)(((ltid >> 3) + u * (WS_NROLE / 8)) * P.Cout + q * EPL);
#pragma unroll
    for (int u = 0; u < WS_UF; ++u) uoff[u] = (unsigned)(((ltid >> 3) + u * (WS_NROLE / 8)) * P.Cin + q * EPL);
    // geometry of a tile's staging.  Every tile stages the LAST adv frames of its window through the two-deep register
    // pipeline (a constant 8 loads per tile: the compiler's "all but the youngest 8" wait stays exact); a FRESH tile
    // stages the first keep frames as well, synchronously at commit time (once per n+1 tiles: the pipeline drains there).
    struct Geo { int rows; size_t pos0; long long row0; int nrow, lo, hi, wrow; bool valid; };
    auto geo = [&](int k, const TPos& c, int w, int f_lo, int f_n) __attribute__((always_inline)) {   // window frames [f_lo, f_lo + f_n)
      Geo t;
      t.valid = k < ntile;
      const int n = t.valid ? c.n : 0;
      const int m0 = (t.valid ? c.mq : 0) * P.F;
      const int nf = min(P.F, P.Tz - m0);
      t.rows = t.valid ? nf * V : 0;
      const int fin0 = m0 + P.min_off;                      // first input frame of the window (stride 1)
      const int win_frames = nf - 1 + P.Fin - (P.F - 1);    // frames the tile needs
      const int fr0 = fin0 + f_lo;                          // first staged frame as an input frame
      t.nrow = t.valid ? max(0, min(f_n, win_frames - f_lo)) * V : 0;
      t.lo = fr0 < 0 ? min(t.nrow, -fr0 * V) : 0;           // staged rows [lo, hi) exist in the sequence, the others are zeros
      t.hi = max(t.lo, min(t.nrow, (P.Tin - fr0) * V));
      t.wrow = (w + f_lo) * V;                              // first destination row in the region
      t.pos0 = (size_t)(n * P.Tz + m0) * V;
      t.row0 = (long long)(n * P.Tin + fr0) * V;
      return t;
    };
    // INTERIOR tiles (all F frames present, the staged frames entirely inside the sequence, full 64-channel blocks: all but
    // the first and last tiles of a sequence) skip every per-slot test: which slots exist is a per-thread constant there
    const bool allq = P.Cout - o0 >= 64 && P.Cin - i0 >= 64;
    bool zex[WS_UZ], sex[WS_US];
    unsigned zoffm[WS_UZ], uoffm[WS_US];
#pragma unroll
    for (int u = 0; u < WS_UZ; ++u) { zex[u] = (ltid >> 3) + u * (WS_NROLE / 8) < P.F * V; zoffm[u] = zex[u] ? zoff[u] : 0u; }
#pragma unroll
    for (int u = 0; u < WS_US; ++u) { sex[u] = (ltid >> 3) + u * (WS_NROLE / 8) < adv * V; uoffm[u] = sex[u] ? uoff[u] : 0u; }
    auto interior = [&](const Geo& t) __attribute__((always_inline)) {
      return allq && t.valid && t.rows == P.F * V && t.lo == 0 && t.hi == adv * V && t.nrow == adv * V;
    };
    // conv-bias column sums of a dz vector without unpacking it: v_dot2 against (1,0) and (0,1)
    // (the (1,0) / (0,1) selectors are kept in registers behind an asm: as literal bf16 vectors the compiler encoded them
    //  as packed fp16 inline constants and the sums came out wrong)
    unsigned sel_lo, sel_hi;
    {
      const unsigned one = std::is_same<T, __bf16>::value ? 0x3f80u : 0x3c00u;
      asm volatile("v_mov_b32 %0, %1" : "=v"(sel_lo) : "s"(one));
      asm volatile("v_mov_b32 %0, %1" : "=v"(sel_hi) : "s"(one << 16));
    }
    auto bias_add = [&](const frag_t& v) __attribute__((always_inline)) {
#ifdef TWG_X_NOBIAS         /* experiment build: no conv-bias column sums (dbias wrong) */
      return;
#endif
#pragma unroll
      for (int e = 0; e < EPL; e += 2) {
        if constexpr (std::is_same<T, __bf16>::value) {
          const bf16x2 p = {v[e], v[e + 1]};
          bs[e] = __builtin_amdgcn_fdot2_f32_bf16(p, __builtin_bit_cast(bf16x2, sel_lo), bs[e], false);
          bs[e + 1] = __builtin_amdgcn_fdot2_f32_bf16(p, __builtin_bit_cast(bf16x2, sel_hi), bs[e + 1], false);
        } else {
          const f16x2 p = {v[e], v[e + 1]};
          bs[e] = __builtin_amdgcn_fdot2(p, __builtin_bit_cast(f16x2, sel_lo), bs[e], false);
          bs[e + 1] = __builtin_amdgcn_fdot2(p, __builtin_bit_cast(f16x2, sel_hi), bs[e + 1], false);
        }
      }
    };
    auto transform = [&](frag_t& v) __attribute__((always_inline)) {
      // per dword: unpack two elements (shift / and), two plain fma, ONE conversion of the pair, ReLU as v_pk_max_i16 on the
      // packed result (a negative float is a negative int16) -- the form of tconv_lean.hip; packed fp32 math (v_pk_fma_f32)
      // costs 22 cycles per instruction next to the MFMA waves (tools/valu_beside_mfma.hip)
      u32x4 w = __builtin_bit_cast(u32x4, v);
      uint32_t d4[4] = {w[0], w[1], w[2], w[3]};
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        float lo, hi;
        twg_unpack2<T>(d4[d], lo, hi);
        lo = __builtin_fmaf(lo, scv[2 * d], shv[2 * d]);
        hi = __builtin_fmaf(hi, scv[2 * d + 1], shv[2 * d + 1]);
        uint32_t p = twg_pk2<T>(lo, hi);
        if (P.pre_relu) p = twg_relu_pk(p);
        d4[d] = p;
      }
      const u32x4 o = {d4[0], d4[1], d4[2], d4[3]};
      v = __builtin_bit_cast(frag_t, o);
    };
    // pipelined part: dz tile + the window's last adv frames.  UNCONDITIONAL loads with clamped addresses.
    auto issue = [&](int k, const TPos& c, int w, u32x4 (&RZ)[WS_UZ], u32x4 (&RS)[WS_US]) __attribute__((always_inline)) {
      const Geo t = geo(k, c, w, keep, adv);
      const T* zb = dzg + (t.valid ? t.pos0 * P.Cout + o0 : 0);                       // wave-uniform
      const T* ub = gg + (t.valid ? (t.row0 + t.lo) * P.Cin + i0 : 0);              // first row that exists
      if (interior(t)) {
#pragma unroll
        for (int u = 0; u < WS_UZ; ++u) RZ[u] = TWG_LD_DZ(zb + zoffm[u]);
#pragma unroll
        for (int u = 0; u < WS_US; ++u) RS[u] = TWG_LD_U(ub + uoffm[u]);
        return;
      }
      const unsigned ushift = (unsigned)(t.lo * P.Cin);
#pragma unroll
      for (int u = 0; u < WS_UZ; ++u) {
        const int r = (ltid >> 3) + u * (WS_NROLE / 8);
        const bool live = zlive_q && r < t.rows;
        RZ[u] = TWG_LD_DZ(zb + (live ? zoff[u] : 0u));
      }
#pragma unroll
      for (int u = 0; u < WS_US; ++u) {
        const int r = (ltid >> 3) + u * (WS_NROLE / 8);
        const bool live = ulive_q && r >= t.lo && r < t.hi;
        RS[u] = TWG_LD_U(ub + (live ? uoff[u] - ushift : 0u));
      }
    };
    auto commit = [&](int k, const TPos& c, int w, bool fresh, u32x4 (&RZ)[WS_UZ], u32x4 (&RS)[WS_US]) __attribute__((always_inline)) {
      const Geo t = geo(k, c, w, keep, adv);
      if (!t.valid) return;
      T* dzs = reinterpret_cast<T*>(smem + ((k & 1) ? P.off_dz1 : P.off_dz)) + sub * dz_sub + ql * EPL;
      T* ureg = reinterpret_cast<T*>(smem + P.off_u) + sub * u_sub + ql * EPL;
      if (interior(t)) {
#pragma unroll
        for (int u = 0; u < WS_UZ; ++u) {
          const int r = (ltid >> 3) + u * (WS_NROLE / 8);
          frag_t v = __builtin_bit_cast(frag_t, RZ[u]);
          if (!zex[u]) zero_frag<T>(v);
          else if (P.dbias) bias_add(v);
          *reinterpret_cast<frag_t*>(dzs + r * CB) = v;
        }
        T* us = ureg + t.wrow * CB;
#pragma unroll
        for (int u = 0; u < WS_US; ++u) {
          const int r = (ltid >> 3) + u * (WS_NROLE / 8);
          if (sex[u]) {
            frag_t v = __builtin_bit_cast(frag_t, RS[u]);
            if (P.pre) transform(v);
            *reinterpret_cast<frag_t*>(us + r * CB) = v;
          }
        }
      } else {
#pragma unroll
        for (int u = 0; u < WS_UZ; ++u) {
          const int r = (ltid >> 3) + u * (WS_NROLE / 8);
          frag_t v = __builtin_bit_cast(frag_t, RZ[u]);
          if (!(zlive_q && r < t.rows)) zero_frag<T>(v);
          else if (P.dbias) bias_add(v);
          *reinterpret_cast<frag_t*>(dzs + r * CB) = v;
        }
        T* us = ureg + t.wrow * CB;
#pragma unroll
        for (int u = 0; u < WS_US; ++u) {
          const int r = (ltid >> 3) + u * (WS_NROLE / 8);
          if (r < t.nrow) {
            frag_t v = __builtin_bit_cast(frag_t, RS[u]);
            if (!(ulive_q && r >= t.lo && r < t.hi)) zero_frag<T>(v);
            else if (P.pre) transform(v);
            *reinterpret_cast<frag_t*>(us + r * CB) = v;
          }
        }
      }
      if (fresh) {
        // the first keep frames of a fresh window: loaded and staged here (7 more vectors, rare)
        const Geo f = geo(k, c, w, 0, keep);
        const T* ub = gg + (f.row0 + f.lo) * P.Cin + i0;
        const unsigned ushift = (unsigned)(f.lo * P.Cin);
        T* us = ureg + f.wrow * CB;
        u32x4 RX[WS_UF - WS_US];
#pragma unroll
        for (int u = 0; u < WS_UF - WS_US; ++u) {
          const int r = (ltid >> 3) + u * (WS_NROLE / 8);
          const bool live = ulive_q && r >= f.lo && r < f.hi;
          RX[u] = TWG_LD_U(ub + (live ? uoff[u] - ushift : 0u));
        }
#pragma unroll
        for (int u = 0; u < WS_UF - WS_US; ++u) {
          const int r = (ltid >> 3) + u * (WS_NROLE / 8);
          if (r < f.nrow) {
            frag_t v = __builtin_bit_cast(frag_t, RX[u]);
            if (!(ulive_q && r >= f.lo && r < f.hi)) zero_frag<T>(v);
            else if (P.pre) transform(v);
            *reinterpret_cast<frag_t*>(us + r * CB) = v;
          }
        }
      }
    };
    // window schedule of the tiles in flight: tile k (being contracted), k+1 (being committed), k+2 (being issued)
    u32x4 ZA[WS_UZ], SA[WS_US], ZB[WS_UZ], SB[WS_US];
    int w1 = 0, w2 = 0;
    bool f1 = true, f2 = true;
    const TPos c0 = tpos_first();
    TPos c1 = tpos_next(c0), c2 = c1;                       // positions of tiles k+1 and k+2
    issue(0, c0, 0, ZA, SA);
    w1 = next_window(1, c1, 0, f1);
    issue(1, c1, w1, ZB, SB);
    __builtin_amdgcn_sched_barrier(0);
    commit(0, c0, 0, true, ZA, SA);
    ws_barrier();                                           // tile 0 staged
    tlast = __builtin_amdgcn_s_memtime();
    int w_k = 0;                                            // window of the tile the compute waves are on
    auto iteration = [&](int k, u32x4 (&Zn)[WS_UZ], u32x4 (&Sn)[WS_US], u32x4 (&Zf)[WS_UZ], u32x4 (&Sf)[WS_US]) __attribute__((always_inline)) {
      // (Zn, Sn): tile k+1, loaded;  (Zf, Sf): free -> tile k+2
      c2 = tpos_next(c1);
      w2 = next_window(k + 2, c2, w1, f2);
      issue(k + 2, c2, w2, Zf, Sf);                         // (past the last tile: dead slots, same number of loads)
      __builtin_amdgcn_sched_barrier(0);
      WSTAMP(1)
      const bool have = k + 1 < ntile;
      const bool late = have && f1 && w_k < P.Fin;         // fresh window at the front would overlap window k: after the barrier
      if (have && !late) commit(k + 1, c1, w1, f1, Zn, Sn);
      WSTAMP(0)
      ws_barrier();                                         // tile k contracted
      if (late) {
        commit(k + 1, c1, w1, f1, Zn, Sn);
        ws_barrier();
      }
      w_k = w1; w1 = w2; f1 = f2; c1 = c2;
      WSTAMP(2)
    };
    for (int k = 0; k < ntile; k += 2) {
      iteration(k, ZB, SB, ZA, SA);
      if (k + 1 < ntile) iteration(k + 1, ZA, SA, ZB, SB);
    }
    if (X_DBG(P) && blockIdx.x == 0 && blockIdx.y == 0 && ltid == 0) { X_DBG(P)[8] = tacc[0]; X_DBG(P)[9] = tacc[1]; X_DBG(P)[10] = tacc[2]; }
    if (P.dbias) {
      // threads sharing a channel vector: lanes q, q+8, ... of every memory wave -> LDS (once per kernel)
#pragma unroll
      for (int e = 0; e < EPL; ++e) {
        float a = bs[e];
#pragma unroll
        for (int msk = 8; msk < 64; msk <<= 1) a += __shfl_xor(a, msk);
        if (lane < 8) atomicAdd(&red[q * EPL + e], a);
      }
    }
  }
#undef WSTAMP
  __syncthreads();

  // ---- flush: D tile rows = o (registers), cols = i (lanes): two 128-byte segments per instruction ----
  // (every element of a workspace slice is written by exactly one workgroup of that blockIdx.x: its (o, i) block of every
  //  tap, and the bias slots of its o-block by the workgroup with i-block 0 -- workgroups without tiles write zeros)
  const bool aux_wg = iblk == 0;
  if (P.ws) {
    float* sl = P.ws + (size_t)blockIdx.x * P.ws_slice;
    if (is_compute) {
      const int ot = wave8 & 1, it = wave8 >> 1;
#pragma unroll
      for (int j = 0; j < JT; ++j) {
        if (j < P.ntaps) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int o = o0 + ot * CB + mfma_row(r, lane), i = i0 + it * CB + (lane & 31);
            if (o < P.Cout && i < P.Cin) sl[((size_t)j * P.Cout + o) * P.Cin + i] = acc[j][r];
          }
        }
      }
    } else if (aux_wg && P.dbias) {
      float* aux = sl + (size_t)P.ntaps * P.Cout * P.Cin;
      if (ltid < 64 && o0 + ltid < P.Cout) aux[o0 + ltid] = red[ltid];
    }
  } else {
    if (is_compute) {
      const int ot = wave8 & 1, it = wave8 >> 1;
#pragma unroll
      for (int j = 0; j < JT; ++j) {
        if (j < P.ntaps) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int o = o0 + ot * CB + mfma_row(r, lane), i = i0 + it * CB + (lane & 31);
            if (o < P.Cout && i < P.Cin) atomicAdd(P.dW + ((size_t)j * P.Cout + o) * P.Cin + i, acc[j][r]);
          }
        }
      }
    } else if (aux_wg && P.dbias) {
      if (ltid < 64 && o0 + ltid < P.Cout) atomicAdd(P.dbias + o0 + ltid, red[ltid]);
    }
  }
}

// host side of the wave-specialised variant; returns -1 when the shape is not one it serves (caller falls back)
template <typename T, int JT>
int launch_ws(TwgParams& P, int grid_cap, hipStream_t stream) {
  if constexpr (sizeof(T) != 2) return -1;
  else {
    static const int forced = [] { const char* e = getenv("ISTGCN_WGRAD_WS"); return e ? atoi(e) : -1; }();
    if (forced == 0) return -1;
    if (P.in_mul != 1 || P.Cin % 8 || P.Cout % 8 || P.Cin < 64 || P.Cout < 64) return -1;
    if (P.Fin * P.V * 8 > WS_UF * WS_NROLE || P.F * P.V * 8 > WS_US * WS_NROLE || P.Fin <= P.F) return -1;
    P.n_iblk = ceil_div(P.Cin, 64);
    const int n_oblk = ceil_div(P.Cout, 64);
    const int esz = 2;
    size_t base = 0;
    base = (base + 15) & ~(size_t)15; P.off_S = (int)base; base += 64 * 4;
    const size_t dzb = (size_t)2 * TR * CB * esz;
    base = (base + 15) & ~(size_t)15; P.off_dz = (int)base; base += dzb;
    P.off_dz1 = (int)base; base += dzb;
    base = (base + 15) & ~(size_t)15; P.off_u = (int)base;
    // region: as many slides as fit, at least until an overflow re-stage can run beside the window it replaces (n*F >= Fin)
    const int span_rows = (P.Fin - P.F) * P.V;              // tap span in rows; pad positions of a tile read TR + span rows
    int n_adv = 0;
    for (int n = 8; n >= 1; --n) {
      const size_t rows = (size_t)(n * P.F) * P.V + TR + span_rows;
      if (base + 2 * rows * CB * esz <= 158 * 1024) { n_adv = n; break; }
    }
    if (n_adv * P.F < P.Fin) return -1;
    P.capf = P.Fin + n_adv * P.F;
    P.urows = n_adv * P.F * P.V + TR + span_rows;
    const size_t off = base + (size_t)2 * P.urows * CB * esz;
    const int blocks = n_oblk * P.n_iblk;
    auto kfn = twg_ws_kernel<T, JT>;
    static std::atomic<unsigned long long> optin{0};
    if (int ea_ = istgcn_lds_optin((const void*)kfn, optin)) return ea_;
    if (grid_cap < 1) grid_cap = istgcn_resident_blocks((const void*)kfn, WS_NTH, off);
    int gx = grid_cap / blocks;
    if (gx < 1) gx = 1;
    if (gx > P.total_tiles) gx = P.total_tiles;
    const int n0 = P.ntaps * P.Cout * P.Cin, n1 = P.Cout;
    const int nsl = gx;
    if (P.ws && ((long long)nsl * (n0 + n1) > P.ws_slice || nsl < 128)) P.ws = nullptr;   // too small / atomics are as fast
    P.ws_slice = n0 + n1;
    unsigned long long* dbuf = nullptr;
#ifdef ISTGCN_EXPERIMENT
    { const char* e = getenv("ISTGCN_WGRAD_ABL"); P.abl = e ? atoi(e) : 0; }
    if (getenv("ISTGCN_WGRAD_DBG")) {
      static unsigned long long* dbuf_s = nullptr;
      if (!dbuf_s) (void)hipMalloc(&dbuf_s, 16 * sizeof(unsigned long long));
      dbuf = dbuf_s;
      (void)hipMemsetAsync(dbuf, 0, 16 * sizeof(unsigned long long), stream);
      P.dbg = dbuf;
    }
#endif
    ISTGCN_LAUNCH(kfn, dim3(gx, blocks), dim3(WS_NTH), off, stream, P);
    ISTGCN_CHECK_LAUNCH();
    if (dbuf) {
      unsigned long long h[16];
      (void)hipMemcpyAsync(h, dbuf, sizeof(h), hipMemcpyDeviceToHost, stream);
      (void)hipStreamSynchronize(stream);
      fprintf(stderr, "wgrad_ws dbg Cin=%d Cout=%d tiles/wg=%llu capf=%d | compute: mfma %llu barrier %llu | memory: commit %llu issue %llu barrier %llu\n",
              P.Cin, P.Cout, h[7], P.capf, h[0], h[1], h[8], h[9], h[10]);
    }
    if (P.ws) {
      int ny = nsl / 16;
      ny = ny < 1 ? 1 : (ny > 16 ? 16 : ny);
      dim3 rgrid(ceil_div(n0 + (P.dbias ? n1 : 0), 1024), ny);
      ISTGCN_LAUNCH(wgrad_reduce_kernel, rgrid, dim3(256), 0, stream, (const float*)P.ws, P.ws_slice, nsl, P.dW, n0,
                    P.dbias, P.dbias ? n1 : 0);
      ISTGCN_CHECK_LAUNCH();
    }
    return ISTGCN_OK;
  }
}

template <typename T, int JT, int OT, int IT, int PS, bool AGG, bool VEC>
int launch_vec(TwgParams& P, int grid_cap, hipStream_t stream) {
  constexpr int TS = JT >= 9 ? 3 : 1;
  const int esz = sizeof(T);
  P.n_iblk = ceil_div(P.Cin, IT * CB);
  const int n_oblk = ceil_div(P.Cout, OT * CB);
  P.capf = P.Fin;
  P.urows = AGG ? P.ntaps * TR : P.Fin * P.V;
  size_t off = (size_t)2 * TR * 2;
  off = (off + 15) & ~(size_t)15; P.off_urow = (int)off;
  off += AGG ? (size_t)(P.ntaps * P.V + 1) * 4 : (size_t)TR * 2;
  const bool mfma_agg = AGG && esz == 2 && P.V <= 32;
  P.mfma_agg = mfma_agg;
  if (AGG) {
    if (!mfma_agg) {                               // compressed adjacency columns: only the VALU aggregation reads them
      off = (off + 15) & ~(size_t)15; P.off_csr_v = (int)off; off += P.nnz_cap;
      off = (off + 15) & ~(size_t)15; P.off_csr_a = (int)off; off += (size_t)P.nnz_cap * 4;
    }
    off = (off + 15) & ~(size_t)15; P.off_S = (int)off; off += (size_t)P.V * (OT * CB + 1) * 4;
  }
  // MFMA aggregation reads a 32-row k-range per frame: zero rows behind the last frame
  P.dz_rows = mfma_agg ? (TR > (P.F - 1) * P.V + 32 ? TR : (P.F - 1) * P.V + 32) : TR;
  if (mfma_agg) { off = (off + 15) & ~(size_t)15; P.off_afrag = (int)off; off += (size_t)P.ntaps * 2 * 64 * 16; }
  off = (off + 15) & ~(size_t)15; P.off_dz = (int)off; off += (size_t)(AGG && IT > OT ? IT : OT) * P.dz_rows * CB * esz;
  off = (off + 15) & ~(size_t)15; P.off_u = (int)off;
  if (!AGG && JT >= 9) {
    // one workgroup per CU anyway (accumulator registers): spend the LDS on slack for the sliding u window
    const int adv = P.in_mul * P.F;
    for (int n_adv = 3; n_adv >= 1; --n_adv)
      if (P.Fin > adv && (n_adv + 1) * adv >= P.Fin - adv &&      // move-back source and destination stay disjoint
          off + (size_t)IT * (P.Fin + n_adv * adv) * P.V * CB * esz <= 150 * 1024) {
        P.capf = P.Fin + n_adv * adv;
        P.urows = P.capf * P.V;
        break;
      }
  }
  off += (size_t)IT * P.urows * CB * esz;
  if (off > 160 * 1024 || P.urows > 65535) return ISTGCN_EINVAL;
  const int blocks = n_oblk * P.n_iblk;
  auto kfn = tconv_wgrad_kernel<T, JT, OT, IT, PS, AGG, TS, VEC>;
  static std::atomic<unsigned long long> optin{0};
  if (int ea_ = istgcn_lds_optin((const void*)kfn, optin)) return ea_;
  // persistent grid: exactly the workgroups that are resident at once (every extra one pays a full flush of its
  // accumulators through atomics and waits for a slot anyway)
  if (grid_cap < 1) grid_cap = istgcn_resident_blocks((const void*)kfn, 64 * OT * IT * PS * TS, off);
  int gx = grid_cap / blocks;
  if (gx < 1) gx = 1;
  if (gx > P.total_tiles) gx = P.total_tiles;
  dim3 grid(gx, blocks);
#ifdef ISTGCN_EXPERIMENT
  static const bool dbg = getenv("ISTGCN_DEBUG") != nullptr;
#else
  constexpr bool dbg = false;
#endif
  if (dbg)
    fprintf(stderr, "[istgcn] wgrad<%s JT=%d OT=%d IT=%d PS=%d TS=%d %s> Cin=%d Cout=%d taps=%d grid=(%d,%d) lds=%zu capf=%d Fin=%d\n",
            esz == 2 ? "16bit" : "f32", JT, OT, IT, PS, TS, AGG ? "agg" : "conv", P.Cin, P.Cout, P.ntaps, gx, blocks, off,
            P.capf, P.Fin);
  const int n0 = P.ntaps * P.Cout * P.Cin, n1 = AGG ? P.V * P.Cout : P.Cout;
  float* aux_dst = AGG ? P.S : P.dbias;
  const int nsl = gx * PS;
  if (P.ws && ((long long)nsl * (n0 + n1) > P.ws_slice || nsl < 128)) P.ws = nullptr;   // too small / atomics are as fast
  P.ws_slice = n0 + n1;
  ISTGCN_LAUNCH(kfn, grid, dim3(64 * OT * IT * PS * TS), off, stream, P);
  ISTGCN_CHECK_LAUNCH();
  if (P.ws) {
    int ny = nsl / 16;
    ny = ny < 1 ? 1 : (ny > 16 ? 16 : ny);
    dim3 rgrid(ceil_div(n0 + (aux_dst ? n1 : 0), 1024), ny);
    ISTGCN_LAUNCH(wgrad_reduce_kernel, rgrid, dim3(256), 0, stream, (const float*)P.ws, P.ws_slice, nsl, P.dW, n0,
                  aux_dst, aux_dst ? n1 : 0);
    ISTGCN_CHECK_LAUNCH();
  }
  return ISTGCN_OK;
}

template <typename T, int JT, int OT, int IT, int PS, bool AGG>
int launch_cfg(TwgParams& P, int grid_cap, hipStream_t stream) {
  // whole 16-byte channel vectors everywhere (every layer but the 3-channel input) or the element-wise variant
  constexpr int EPL = Elem<T>::EPL;
  if (P.Cin % EPL == 0 && P.Cout % EPL == 0) return launch_vec<T, JT, OT, IT, PS, AGG, true>(P, grid_cap, stream);
  return launch_vec<T, JT, OT, IT, PS, AGG, false>(P, grid_cap, stream);
}

template <typename T, int JT, bool AGG>
int launch_JT(TwgParams& P, int grid_cap, hipStream_t stream) {
  // channel block per workgroup: as wide as the layer and LDS allow (fewer redundant reads of dz / g, more MFMA work
  // per staged byte); narrow layers fall back to one 32x32 pair split four ways over positions.
  if constexpr (AGG) {
    // few accumulator tiles per wave (K <= 4): spend the registers on waves instead -- 8 waves, finer position slices
    if constexpr (sizeof(T) == 2) {
      if (P.Cout > 32 && P.Cin > 32) return launch_cfg<T, JT, 2, 2, 2, AGG>(P, grid_cap, stream);
    }
    if (P.Cout > 32) return launch_cfg<T, JT, 2, 1, 4, AGG>(P, grid_cap, stream);
    return launch_cfg<T, JT, 1, 1, 8, AGG>(P, grid_cap, stream);
  } else {
    if constexpr (sizeof(T) == 2) {
      if (P.Cout > 32 && P.Cin > 32) return launch_cfg<T, JT, 2, 2, 1, AGG>(P, grid_cap, stream);
    }
    if (P.Cout > 32) return launch_cfg<T, JT, 2, 1, 2, AGG>(P, grid_cap, stream);
    return launch_cfg<T, JT, 1, 1, 4, AGG>(P, grid_cap, stream);
  }
}

template <typename T>
int launch_T(TwgParams& P, int grid_cap, hipStream_t stream) {
  int mn = P.tap_off[0], mx = P.tap_off[0];
  for (int j = 1; j < P.ntaps; ++j) { mn = P.tap_off[j] < mn ? P.tap_off[j] : mn; mx = P.tap_off[j] > mx ? P.tap_off[j] : mx; }
  P.min_off = mn;
  P.F = TR / P.V;
  P.Fin = P.in_mul * (P.F - 1) + (mx - mn) + 1;
  P.tiles_per_seq = ceil_div(P.Tz, P.F);
  P.total_tiles = P.NM * P.tiles_per_seq;
  if (P.ntaps <= 1) return launch_JT<T, 1, false>(P, grid_cap, stream);
  if (P.ntaps <= 3) return launch_JT<T, 3, false>(P, grid_cap, stream);
  if (P.ntaps <= 9) {
    if (P.ntaps > 3) {
      TwgParams Q = P;
      const int rc = launch_ws<T, 9>(Q, grid_cap, stream);
      if (rc >= 0) return rc;
    }
    return launch_JT<T, 9, false>(P, grid_cap, stream);
  }
  return launch_JT<T, 15, false>(P, grid_cap, stream);
}

// ======================================================================================================================
// Wave-specialised graph-conv weight gradient (round 2): dW[k][c][i] += sum_p dz[p][c] * u_k[p][i],  u_k = A_k-aggregated x,
// S[w][c] += sum_f dz[(f,w)][c].  16-bit storage, V <= 32, 64 x 64 channel block per workgroup, K <= 4.  ONE 8-wave
// workgroup per CU, two phases per tile of 128 positions (F frames):
//   A  all eight waves: aggregation of the staged x tile into the K images  (units = (frame, 32-channel tile, partition):
//      x^T by ds_read_b64_tr_b16, A_k fragments from LDS, 2 MFMAs, convert, swizzled image rows)
//   B  waves 0-3: contraction of the images with the dz tile (K MFMAs per k-step of 16 positions, compile-time operand
//      offsets);  waves 4-7: x / dz tiles of tile k+1 registers -> LDS (double-buffered), tile k+2's loads, S column sums
// The round-1 kernel did the same work on two 8-wave workgroups per CU, every wave through every phase in lock-step.
// ======================================================================================================================
template <typename T, int KT, int OTW>
__global__ __launch_bounds__(WS_NTH, 2) void gwg_ws_kernel(const TwgParams P) {
  // OTW = o-tiles per compute wave: the workgroup's channel block is (64 * OTW) x 64.  Two o-tiles halve how often an x
  // block is aggregated (once per o-block) for layers with >= 128 output channels.
  constexpr int OB = 64 * OTW;                             // output channels per workgroup
  constexpr int NZQ = OB / 8;                              // 16-byte vectors per dz row
  constexpr int UZ = TR * NZQ / WS_NROLE;                  // dz vectors per memory thread and tile
  using E = Elem<T>;
  constexpr int EPL = E::EPL;
  static_assert(EPL == 8, "16-bit storage only");
  typedef typename E::frag frag_t;
  constexpr int RB = CB * (int)sizeof(T);                  // bytes per sub-tile row (64)
  constexpr int SLS = OB + 1;                              // S_l row stride (odd: joints land in different banks)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* S_l = reinterpret_cast<float*>(smem + P.off_S);                          // [V][SLS]
  T* afrag = reinterpret_cast<T*>(smem + P.off_afrag);                            // [K][2][64][8] fragments of A_k
  T* img = reinterpret_cast<T*>(smem + P.off_u);                                  // [2 sub][K][TR][CB]
  const int tid = (int)(threadIdx.x ^ ISTGCN_ROLE_FLIP), lane = tid & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool is_compute = wave8 < 4;
  const int ltid = tid & (WS_NROLE - 1);
  const int V = P.V, K = P.ntaps;
  const int oblk = blockIdx.y / P.n_iblk, iblk = blockIdx.y - oblk * P.n_iblk;
  const int o0 = oblk * OB, i0 = iblk * 64;
  const int dz_sub = TR * CB, x_sub = P.dz_rows * CB;      // elements per 32-channel sub-tile
  const int dz_half = 2 * OTW * dz_sub * (int)sizeof(T);   // bytes of one dz half ([2*OTW sub][TR][CB])

  for (int idx = tid; idx < V * SLS; idx += WS_NTH) S_l[idx] = 0.f;
  // B-operand fragments of the adjacency: lane (w = lane&31, h = lane>>5), k-step s, element j = A[k][v = 16s + 8h + j][w]
  for (int idx = tid; idx < K * 2 * 64; idx += WS_NTH) {
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
  {
    // zeros once: the x halves (the aggregation reads a 32-row range per frame: rows behind the tile must be finite and
    // meet zero adjacency rows), the dz halves, the images (rows behind a short tile meet zero dz rows)
    frag_t z;
    zero_frag<T>(z);
    T* x0 = reinterpret_cast<T*>(smem + P.off_dz);          // [dz0][dz1][x0][x1][img] are contiguous
    const int nv = (int)((P.off_u - P.off_dz) / 16) + 2 * K * dz_sub / EPL;
    for (int i = tid; i < nv; i += WS_NTH) *reinterpret_cast<frag_t*>(x0 + i * EPL) = z;
  }
  __syncthreads();

  const T* dzg = reinterpret_cast<const T*>(P.dz);
  const T* xg = reinterpret_cast<const T*>(P.g);
  const int chunk = (P.total_tiles + gridDim.x - 1) / gridDim.x;
  const int t_begin = blockIdx.x * chunk, t_end = min(P.total_tiles, t_begin + chunk);
  const int ntile = t_end > t_begin ? t_end - t_begin : 0;
  struct TPos { int n, mq; };
  auto tpos_first = [&]() __attribute__((always_inline)) { TPos c; c.n = t_begin / P.tiles_per_seq; c.mq = t_begin - c.n * P.tiles_per_seq; return c; };
  auto tpos_next = [&](TPos c) __attribute__((always_inline)) { if (++c.mq == P.tiles_per_seq) { c.mq = 0; ++c.n; } return c; };
  auto tile_nf = [&](const TPos& c) __attribute__((always_inline)) { return min(P.F, P.Tz - c.mq * P.F); };

  // ---- phase A: aggregation, units dealt round-robin to the eight waves ----
  const int a_grp = lane >> 4, a_h = a_grp >> 1, a_cblk = (a_grp & 1) * 16, a_q4 = (lane & 15) >> 2, a_pp = lane & 3;
  const int a_w = lane & 31;
  const int a_src = (8 * a_h + a_q4) * CB + a_cblk + 4 * a_pp;       // element offset in an x sub-tile (frame 0, k-step 0)
  const int NC = 2 * K;                                              // unit columns (32-channel tile ct, partition kk)
  const unsigned rcpNC = (65536u + NC - 1) / NC, rcpK = (65536u + K - 1) / K;
  auto aggregate = [&](int half, int nf) __attribute__((always_inline)) {
    const T* xs = reinterpret_cast<const T*>(smem + (half ? P.off_x1 : P.off_dz + 2 * dz_half));
    const int nunit = nf * NC;
    for (int u = wave8; u < nunit; u += 8) {
      const int f = (int)(((unsigned)u * rcpNC) >> 16), c = u - f * NC;
      const int ct = (int)(((unsigned)c * rcpK) >> 16), kk = c - ct * K;
      const T* r0 = xs + ct * x_sub + f * V * CB + a_src;
      const frag_t x0 = tr_pair<T>(r0, r0 + 4 * CB);
      const frag_t x1 = tr_pair<T>(r0 + 16 * CB, r0 + 20 * CB);
      const T* af = afrag + kk * (2 * 64 * EPL) + lane * EPL;
      const frag_t b0 = *reinterpret_cast<const frag_t*>(af);
      const frag_t b1 = *reinterpret_cast<const frag_t*>(af + 64 * EPL);
      f32x16 d;
#pragma unroll
      for (int r = 0; r < 16; ++r) d[r] = 0.f;
      mma_kgroup(d, x0, b0);
      mma_kgroup(d, x1, b1);
      if (a_w < V) {
        // a lane owns one image row; rows are 64 bytes apart (what the contraction's transposed reads want), so sixteen
        // lanes storing the same 16-byte block would share two banks: the block index is XOR-swizzled with row bits 1-2
        const int row = f * V + a_w, sw = (row >> 1) & 3;
        T* dst = img + ((ct * K + kk) * TR + row) * CB + 4 * (lane >> 5);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float v4[4] = {d[4 * g], d[4 * g + 1], d[4 * g + 2], d[4 * g + 3]};
          store4(dst + 8 * (g ^ sw), v4);
        }
      }
    }
  };

  f32x16 acc[OTW][KT];
  unsigned long long tacc[5] = {0, 0, 0, 0, 0}, tlast = 0;
#define GSTAMP(i) if (X_DBG(P)) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); tacc[i] += now_ - tlast; tlast = now_; }
#ifdef ISTGCN_X_PRIO        /* experiment build: issue priority per role (1: compute waves high, 2: memory waves high) */
  if ((ISTGCN_X_PRIO == 1) == is_compute) __builtin_amdgcn_s_setprio(3);
#endif
  if (is_compute) {
    // =========================================== compute waves ===========================================
    const int ot = wave8 & 1, it = wave8 >> 1;
#pragma unroll
    for (int o = 0; o < OTW; ++o)
#pragma unroll
      for (int j = 0; j < KT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[o][j][r] = 0.f;
    const int grp = lane >> 4, h = grp >> 1, cblk = (grp & 1) * 16;
    const int q = (lane & 15) >> 2, pp = lane & 3;
    const int coff = cblk + 4 * pp;
    const int lrow = 8 * h + q;
    // swizzled column offsets of this lane's two image rows (8h + q and + 4, mod 16, in every k-step)
    const int coff_u0 = (((coff >> 3) ^ ((lrow >> 1) & 3)) << 3) + (coff & 7);
    const int coff_u1 = (((coff >> 3) ^ (((lrow + 4) >> 1) & 3)) << 3) + (coff & 7);
    // indicator fragments (B operand: lane (w = lane & 31, kg = lane >> 5) holds rows p = 16 ks + 8 kg + j, j = 0..7)
    constexpr int NKS = TR / 16;
    const bool s_wave = it == 0 && P.S != nullptr;
    frag_t ind[NKS];
    f32x16 accS[OTW];
#pragma unroll
    for (int o = 0; o < OTW; ++o)
#pragma unroll
      for (int r = 0; r < 16; ++r) accS[o][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
#pragma unroll
      for (int jj = 0; jj < EPL; ++jj) {
        const int p = 16 * ks + 8 * (lane >> 5) + jj;
        ind[ks][jj] = E::from_f((p % V) == (lane & 31) ? 1.f : 0.f);
      }
    }
    TPos c = tpos_first();
    ws_barrier();                                           // tile 0 staged
    tlast = __builtin_amdgcn_s_memtime();
    for (int k = 0; k < ntile; ++k) {
      aggregate(k & 1, tile_nf(c));
      c = tpos_next(c);
      GSTAMP(0)
      ws_barrier();                                         // A done: images complete
      GSTAMP(1)
      // this wave's o-tiles are sub-tiles ot * OTW + o of the dz tile
      const unsigned char* ap = smem + ((k & 1) ? P.off_dz1 : P.off_dz) + (ot * OTW * dz_sub + coff) * (int)sizeof(T) + lrow * RB;
      const unsigned char* u0 = smem + P.off_u + (it * K * dz_sub + coff_u0) * (int)sizeof(T) + lrow * RB;
      const unsigned char* u1 = smem + P.off_u + (it * K * dz_sub + coff_u1) * (int)sizeof(T) + (lrow + 4) * RB;
      constexpr int NK = TR / 16;
      frag_t a0[OTW], a1[OTW], b0[KT], b1[KT];
      auto load_k = [&](int ks, frag_t (&a)[OTW], frag_t (&b)[KT]) __attribute__((always_inline)) {
#pragma unroll
        for (int o = 0; o < OTW; ++o)
          a[o] = tr_pair<T>(reinterpret_cast<const T*>(ap + o * TR * RB + ks * 16 * RB), reinterpret_cast<const T*>(ap + o * TR * RB + ks * 16 * RB + 4 * RB));
#pragma unroll
        for (int j = 0; j < KT; ++j) {
          const int jv = j < K ? j : 0;                     // padding partitions alias partition 0 (computed, never flushed)
          b[j] = tr_pair<T>(reinterpret_cast<const T*>(u0 + jv * TR * RB + ks * 16 * RB), reinterpret_cast<const T*>(u1 + jv * TR * RB + ks * 16 * RB));
        }
      };
      auto mma_k = [&](int ks, const frag_t (&a)[OTW], const frag_t (&b)[KT]) __attribute__((always_inline)) {
#pragma unroll
        for (int o = 0; o < OTW; ++o) {
#pragma unroll
          for (int j = 0; j < KT; ++j) mma_kgroup(acc[o][j], a[o], b[j]);
          if (s_wave) mma_kgroup(accS[o], a[o], ind[ks]);
        }
      };
      load_k(0, a0, b0);
#pragma unroll
      for (int ks = 0; ks < NK; ks += 2) {
        load_k(ks + 1, a1, b1);
        __builtin_amdgcn_sched_barrier(0);
        mma_k(ks, a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        if (ks + 2 < NK) load_k(ks + 2, a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        mma_k(ks + 1, a1, b1);
        __builtin_amdgcn_sched_barrier(0);
      }
      GSTAMP(2)
      ws_barrier();                                         // B done: images free, tile k+1 staged
      GSTAMP(3)
    }
    if (X_DBG(P) && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) { for (int i = 0; i < 4; ++i) X_DBG(P)[i] = tacc[i]; X_DBG(P)[7] = (unsigned long long)ntile; }
    if (s_wave && (lane & 31) < V) {
      // D tile rows = output channel (registers), cols = joint (lanes) -> S_l[w][c] for the common flush below
#pragma unroll
      for (int o = 0; o < OTW; ++o)
#pragma unroll
        for (int r = 0; r < 16; ++r) S_l[(lane & 31) * SLS + (ot * OTW + o) * CB + mfma_row(r, lane)] = accS[o][r];
    }
  } else {
    // =========================================== memory waves ============================================
    // thread maps: x rows are 8 vectors (64 channels) wide, dz rows NZQ vectors (64 * OTW channels)
    const int qx = ltid & 7, rx0 = ltid >> 3;
    const int qz = ltid & (NZQ - 1), rz0 = ltid / NZQ;
    constexpr int RXS = WS_NROLE / 8, RZS = WS_NROLE / NZQ;   // rows per sweep
    const bool zlive_q = o0 + qz * EPL < P.Cout, xlive_q = i0 + qx * EPL < P.Cin;
    unsigned zoff[UZ], xoff[WS_UZ];
#pragma unroll
    for (int u = 0; u < UZ; ++u) zoff[u] = (unsigned)((rz0 + u * RZS) * P.Cout + qz * EPL);
#pragma unroll
    for (int u = 0; u < WS_UZ; ++u) xoff[u] = (unsigned)((rx0 + u * RXS) * P.Cin + qx * EPL);
    // a constant UZ + 4 loads per tile (dead slots read the tile's first vector and are zeroed on commit)
    auto issue = [&](int k, const TPos& c, u32x4 (&RZ)[UZ], u32x4 (&RX)[WS_UZ]) __attribute__((always_inline)) {
      const bool valid = k < ntile;
      const int rows = valid ? tile_nf(c) * V : 0;
      const size_t pos0 = valid ? (size_t)(c.n * P.Tz + c.mq * P.F) * V : 0;
      const T* zb = dzg + pos0 * P.Cout + (valid ? o0 : 0);
      const T* xb = xg + pos0 * P.Cin + (valid ? i0 : 0);
#pragma unroll
      for (int u = 0; u < UZ; ++u)
        RZ[u] = TWG_LD_DZ(zb + ((zlive_q && rz0 + u * RZS < rows) ? zoff[u] : 0u));
#pragma unroll
      for (int u = 0; u < WS_UZ; ++u)
        RX[u] = TWG_LD_U(xb + ((xlive_q && rx0 + u * RXS < rows) ? xoff[u] : 0u));
    };
    auto commit = [&](int k, const TPos& c, u32x4 (&RZ)[UZ], u32x4 (&RX)[WS_UZ]) __attribute__((always_inline)) {
      if (k >= ntile) return;
      const int rows = tile_nf(c) * V;
      T* dzs = reinterpret_cast<T*>(smem + ((k & 1) ? P.off_dz1 : P.off_dz)) + (qz >> 2) * dz_sub + (qz & 3) * EPL;
      T* xs = reinterpret_cast<T*>(smem + ((k & 1) ? P.off_x1 : P.off_dz + 2 * dz_half)) + (qx >> 2) * x_sub + (qx & 3) * EPL;
#pragma unroll
      for (int u = 0; u < UZ; ++u) {
        const int r = rz0 + u * RZS;
        frag_t vz = __builtin_bit_cast(frag_t, RZ[u]);
        if (!(zlive_q && r < rows)) zero_frag<T>(vz);
        *reinterpret_cast<frag_t*>(dzs + r * CB) = vz;
      }
#pragma unroll
      for (int u = 0; u < WS_UZ; ++u) {
        const int r = rx0 + u * RXS;
        frag_t vx = __builtin_bit_cast(frag_t, RX[u]);
        if (!(xlive_q && r < rows)) zero_frag<T>(vx);
        *reinterpret_cast<frag_t*>(xs + r * CB) = vx;
      }
    };
    u32x4 ZA[UZ], XA[WS_UZ], ZB[UZ], XB[WS_UZ];
    const TPos c0 = tpos_first();
    TPos ck = c0, c1 = tpos_next(c0), c2 = c1;
    issue(0, c0, ZA, XA);
    issue(1, c1, ZB, XB);
    __builtin_amdgcn_sched_barrier(0);
    commit(0, c0, ZA, XA);
    ws_barrier();                                           // tile 0 staged
    tlast = __builtin_amdgcn_s_memtime();
    auto iteration = [&](int k, u32x4 (&Zn)[UZ], u32x4 (&Xn)[WS_UZ], u32x4 (&Zf)[UZ], u32x4 (&Xf)[WS_UZ]) __attribute__((always_inline)) {
      aggregate(k & 1, tile_nf(ck));                        // phase A, this role's share
      GSTAMP(0)
      ws_barrier();
      GSTAMP(1)
      c2 = tpos_next(c1);
      issue(k + 2, c2, Zf, Xf);
      __builtin_amdgcn_sched_barrier(0);
      commit(k + 1, c1, Zn, Xn);
      GSTAMP(2)
      ck = c1; c1 = c2;
      ws_barrier();
      GSTAMP(4)
    };
    for (int k = 0; k < ntile; k += 2) {
      iteration(k, ZB, XB, ZA, XA);
      if (k + 1 < ntile) iteration(k + 1, ZA, XA, ZB, XB);
    }
    if (X_DBG(P) && blockIdx.x == 0 && blockIdx.y == 0 && ltid == 0) for (int i = 0; i < 5; ++i) X_DBG(P)[8 + i] = tacc[i];
  }
#undef GSTAMP
  __syncthreads();

  // ---- flush (as the convolution variant; the aux slots are S[w][c] of this o-block, reported by i-block 0) ----
  const bool aux_wg = iblk == 0;
  if (P.ws) {
    float* sl = P.ws + (size_t)blockIdx.x * P.ws_slice;
    if (is_compute) {
      const int ot = wave8 & 1, it = wave8 >> 1;
#pragma unroll
      for (int ow = 0; ow < OTW; ++ow)
#pragma unroll
        for (int j = 0; j < KT; ++j) {
          if (j < K) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int o = o0 + (ot * OTW + ow) * CB + mfma_row(r, lane), i = i0 + it * CB + (lane & 31);
              if (o < P.Cout && i < P.Cin) sl[((size_t)j * P.Cout + o) * P.Cin + i] = acc[ow][j][r];
            }
          }
        }
    } else if (aux_wg && P.S) {
      float* aux = sl + (size_t)K * P.Cout * P.Cin;
      for (int idx = ltid; idx < V * OB; idx += WS_NROLE) {
        const int w = idx / OB, cc = idx - w * OB;
        if (o0 + cc < P.Cout) aux[w * P.Cout + o0 + cc] = S_l[w * SLS + cc];
      }
    }
  } else {
    if (is_compute) {
      const int ot = wave8 & 1, it = wave8 >> 1;
#pragma unroll
      for (int ow = 0; ow < OTW; ++ow)
#pragma unroll
        for (int j = 0; j < KT; ++j) {
          if (j < K) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int o = o0 + (ot * OTW + ow) * CB + mfma_row(r, lane), i = i0 + it * CB + (lane & 31);
              if (o < P.Cout && i < P.Cin) atomicAdd(P.dW + ((size_t)j * P.Cout + o) * P.Cin + i, acc[ow][j][r]);
            }
          }
        }
    } else if (aux_wg && P.S) {
      for (int idx = ltid; idx < V * OB; idx += WS_NROLE) {
        const int w = idx / OB, cc = idx - w * OB;
        if (o0 + cc < P.Cout) atomicAdd(P.S + w * P.Cout + o0 + cc, S_l[w * SLS + cc]);
      }
    }
  }
}

template <typename T, int KT, int OTW>
int launch_gws_o(TwgParams& P, int grid_cap, hipStream_t stream) {
  if constexpr (sizeof(T) != 2) return -1;
  else {
    static const int forced = [] { const char* e = getenv("ISTGCN_WGRAD_WS"); return e ? atoi(e) : -1; }();
    if (forced == 0) return -1;
    if (P.V > 32 || P.Cin % 8 || P.Cout % 8 || P.Cin < 64 || P.Cout < 64 * OTW || P.ntaps > KT) return -1;
    P.n_iblk = ceil_div(P.Cin, 64);
    const int n_oblk = ceil_div(P.Cout, 64 * OTW);
    const int esz = 2, K = P.ntaps;
    P.dz_rows = (P.F - 1) * P.V + 32 > TR ? (P.F - 1) * P.V + 32 : TR;      // x sub-tile rows (32-row k-range of the last frame)
    size_t off = 0;
    P.off_S = (int)off; off += (size_t)P.V * (64 * OTW + 1) * 4;
    off = (off + 15) & ~(size_t)15; P.off_afrag = (int)off; off += (size_t)K * 2 * 64 * 16;
    const size_t dzb = (size_t)2 * OTW * TR * CB * esz, xb = (size_t)2 * P.dz_rows * CB * esz;
    off = (off + 15) & ~(size_t)15; P.off_dz = (int)off; off += dzb;
    P.off_dz1 = (int)off; off += dzb;
    /* x half 0 sits right behind the dz halves */ off += xb;
    P.off_x1 = (int)off; off += xb;
    P.off_u = (int)off; off += (size_t)2 * K * TR * CB * esz;
    if (off > 160 * 1024) return -1;
    if (OTW == 2 && P.total_tiles * n_oblk * P.n_iblk < 2 * 256) return -1;   // too few tiles to keep every CU busy with wide blocks
    const int blocks = n_oblk * P.n_iblk;
    auto kfn = gwg_ws_kernel<T, KT, OTW>;
    static std::atomic<unsigned long long> optin{0};
    if (int ea_ = istgcn_lds_optin((const void*)kfn, optin)) return ea_;
    if (grid_cap < 1) grid_cap = istgcn_resident_blocks((const void*)kfn, WS_NTH, off);
    int gx = grid_cap / blocks;
    if (gx < 1) gx = 1;
    if (gx > P.total_tiles) gx = P.total_tiles;
    const int n0 = K * P.Cout * P.Cin, n1 = P.V * P.Cout;
    const int nsl = gx;
    if (P.ws && ((long long)nsl * (n0 + n1) > P.ws_slice || nsl < 128)) P.ws = nullptr;
    P.ws_slice = n0 + n1;
    unsigned long long* dbuf = nullptr;
#ifdef ISTGCN_EXPERIMENT
    if (getenv("ISTGCN_WGRAD_DBG")) {
      static unsigned long long* dbuf_s = nullptr;
      if (!dbuf_s) (void)hipMalloc(&dbuf_s, 16 * sizeof(unsigned long long));
      dbuf = dbuf_s;
      (void)hipMemsetAsync(dbuf, 0, 16 * sizeof(unsigned long long), stream);
      P.dbg = dbuf;
    }
#endif
    ISTGCN_LAUNCH(kfn, dim3(gx, blocks), dim3(WS_NTH), off, stream, P);
    ISTGCN_CHECK_LAUNCH();
    if (dbuf) {
      unsigned long long h[16];
      (void)hipMemcpyAsync(h, dbuf, sizeof(h), hipMemcpyDeviceToHost, stream);
      (void)hipStreamSynchronize(stream);
      fprintf(stderr, "gwgrad_ws dbg Cin=%d Cout=%d tiles/wg=%llu | compute: agg %llu barA %llu contract %llu barB %llu | memory: agg %llu barA %llu issue+commit %llu ssums %llu barB %llu\n",
              P.Cin, P.Cout, h[7], h[0], h[1], h[2], h[3], h[8], h[9], h[10], h[11], h[12]);
    }
    if (P.ws) {
      int ny = nsl / 16;
      ny = ny < 1 ? 1 : (ny > 16 ? 16 : ny);
      dim3 rgrid(ceil_div(n0 + (P.S ? n1 : 0), 1024), ny);
      ISTGCN_LAUNCH(wgrad_reduce_kernel, rgrid, dim3(256), 0, stream, (const float*)P.ws, P.ws_slice, nsl, P.dW, n0,
                    P.S, P.S ? n1 : 0);
      ISTGCN_CHECK_LAUNCH();
    }
    return ISTGCN_OK;
  }
}

template <typename T, int KT>
int launch_gws(TwgParams& P, int grid_cap, hipStream_t stream) {
  if (P.Cout >= 128) {                                    // 128 x 64 blocks: each x block aggregated once per 128 output channels
    TwgParams Q = P;
    const int rc = launch_gws_o<T, KT, 2>(Q, grid_cap, stream);
    if (rc >= 0) return rc;
  }
  return launch_gws_o<T, KT, 1>(P, grid_cap, stream);
}

template <typename T>
int launch_agg_T(TwgParams& P, int grid_cap, hipStream_t stream) {
  P.min_off = 0;
  P.F = TR / P.V;
  P.Fin = P.F;
  P.tiles_per_seq = ceil_div(P.Tz, P.F);
  P.total_tiles = P.NM * P.tiles_per_seq;
  if (P.ntaps == 3) {
    TwgParams Q = P;
    const int rc = launch_gws<T, 3>(Q, grid_cap, stream);
    if (rc >= 0) return rc;
  }
  if (P.ntaps <= 1) return launch_JT<T, 1, true>(P, grid_cap, stream);
  if (P.ntaps <= 3) return launch_JT<T, 3, true>(P, grid_cap, stream);
  return launch_JT<T, 4, true>(P, grid_cap, stream);
}

}  // namespace

extern "C" int istgcn_tconv_wgrad_rc_ok(int V, int Cin, int Cout, int ntaps, const int* tap_off, int in_mul, int dtype);
extern "C" int istgcn_tconv_wgrad_rc(const void* dz, const void* g, const float* pre, int pre_relu, float* dW, float* dbias,
                                     int NM, int Tin, int Tz, int V, int Cin, int Cout, int ntaps, const int* tap_off,
                                     int in_mul, int dtype, int grid_cap, float* ws, long long ws_floats, void* stream);

extern "C" int istgcn_tconv_wgrad(const void* dz, const void* g, const float* pre, int pre_relu, float* dW,
                                  float* dbias, int NM, int Tin, int Tz, int V, int Cin, int Cout, int ntaps,
                                  const int* tap_off, int in_mul, int dtype, int grid_cap, float* ws,
                                  long long ws_floats, void* stream) {
  if (!dz || !g || !dW || !tap_off) return ISTGCN_EINVAL;
  if (ntaps < 1 || ntaps > 15 || V < 1 || V > 128 || Cin < 1 || Cout < 1 || in_mul < 1 || NM < 0 || Tz < 0)
    return ISTGCN_EINVAL;
  if (!istgcn_dtype_ok(dtype)) return ISTGCN_EINVAL;
  if (NM == 0 || Tz == 0) return ISTGCN_OK;
  // lean form of twg_ws (tconv_wgrad_lean.hip) for the 16-bit trunk shapes; a conv-bias gradient, when asked for (the training
  // step does not: functional.py), is a column-sum kernel of its own next to it
  if (twg_lean_ok(V, Cin, Cout, ntaps, tap_off, in_mul, dtype, Tin, Tz) && (!dbias || (Cout <= 256 && 256 % (Cout / 8) == 0))) {
    const int rc = twg_lean_launch(dz, g, pre, pre_relu, dW, NM, Tin, Tz, V, Cin, Cout, ntaps, tap_off, in_mul, dtype, grid_cap, ws,
                                   ws_floats, (hipStream_t)stream);
    if (rc == ISTGCN_OK && dbias) return twg_lean_dbias(dz, dbias, NM, Tz, V, Cout, dtype, (hipStream_t)stream);
    if (rc >= 0) return rc;      // (-1: the LDS plan does not fit -- decided before anything was launched)
  }
  {
    // frame-tiled kernel (tconv_rc_wgrad.hip): consecutive taps, stride 1 or 2, 64-channel blocks, 16-bit storage.  Measured
    // (tools/twg_exp.py, bf16, NM = 128): it wins where the round-2 kernels have no wave-specialised form for the 9-tap
    // stride-2 layers (128 ch T=300: 307 -> 265 us, 256 ch T=150: 537 -> 478); at stride 1 twg_ws is 5-8 % faster (the
    // 32-row frame tiles cost 28 % more matrix work than its 125-of-128-row tiles), and the 16-tap variant has no
    // registers left for its staging pipeline -- those stay with the round-2 kernels.
    // ISTGCN_TWG_RC=0: round-2 kernels everywhere (A/B timing); =2: the frame-tiled kernel wherever it applies
    static const int mode = [] { const char* e = getenv("ISTGCN_TWG_RC"); return e ? atoi(e) : 1; }();   // dispatch override, read once
    if (mode != 0 && istgcn_tconv_wgrad_rc_ok(V, Cin, Cout, ntaps, tap_off, in_mul, dtype) &&
        (mode == 2 || (in_mul == 2 && ntaps <= 10)))
      return istgcn_tconv_wgrad_rc(dz, g, pre, pre_relu, dW, dbias, NM, Tin, Tz, V, Cin, Cout, ntaps, tap_off, in_mul, dtype,
                                   grid_cap, ws, ws_floats, stream);
  }
  TwgParams P{};
  P.dz = dz; P.g = g; P.pre = pre; P.dW = dW; P.dbias = dbias;
  P.NM = NM; P.Tin = Tin; P.Tz = Tz; P.V = V; P.Cin = Cin; P.Cout = Cout; P.ntaps = ntaps; P.in_mul = in_mul;
  P.pre_relu = pre_relu;
  P.ws = ws_floats > 0 ? ws : nullptr; P.ws_slice = ws_floats;     // capacity until the launcher sets the slice length
  for (int j = 0; j < ntaps; ++j) P.tap_off[j] = tap_off[j];
  for (int j = ntaps; j < MAX_TAPS; ++j) P.tap_off[j] = tap_off[0];    // padding taps: valid addresses, never flushed
  if (dtype == 0) return launch_T<float>(P, grid_cap, (hipStream_t)stream);
  if (dtype == 2) return launch_T<_Float16>(P, grid_cap, (hipStream_t)stream);
  return launch_T<__bf16>(P, grid_cap, (hipStream_t)stream);
}

// Graph-conv weight gradient: same kernel, the K "taps" being the K adjacency partitions (aggregated images of x).
// The partial-sum reduce as its own entry for the register-chained graph-conv weight gradient (gcn_rc_wgrad.hip):
// d0[e] += sum_s ws[s*slice + e] for e < n0, d1[e - n0] likewise for the n1 entries behind.
extern "C" int istgcn_wgrad_reduce(const float* ws, long long slice, int nsl, float* d0, int n0, float* d1, int n1,
                                   void* stream) {
  int ny = nsl / 16;
  ny = ny < 1 ? 1 : (ny > 16 ? 16 : ny);
  dim3 rgrid(ceil_div(n0 + (d1 ? n1 : 0), 1024), ny);
  ISTGCN_LAUNCH(wgrad_reduce_kernel, rgrid, dim3(256), 0, (hipStream_t)stream, ws, slice, nsl, d0, n0, d1, d1 ? n1 : 0);
  ISTGCN_CHECK_LAUNCH();
  return ISTGCN_OK;
}

extern "C" int istgcn_gcn_wgrad_rc_ok(int V, int Cin, int Cout, int K, int dtype);
extern "C" int istgcn_gcn_wgrad_rc_f32(const void* dy, const void* x, const float* A, float* dW, float* S, int NM, int T,
                                       int V, int Cin, int Cout, int K, int grid_cap, float* ws, long long ws_floats,
                                       void* stream);
extern "C" int istgcn_gcn_wgrad_rc(const void* dy, const void* x, const float* A, float* dW, float* S, int NM, int T, int V,
                                   int Cin, int Cout, int K, int dtype, int grid_cap, float* ws, long long ws_floats,
                                   void* stream);

extern "C" int istgcn_gcn_wgrad(const void* dy, const void* x, const float* A, float* dW, float* S, int NM, int T, int V,
                                int Cin, int Cout, int K, int nnz_cap, int dtype, int grid_cap, float* ws,
                                long long ws_floats, void* stream) {
  if (!dy || !x || !A || !dW) return ISTGCN_EINVAL;
  if (V < 1 || V > 128 || Cin < 1 || Cout < 1 || K < 1 || K > 4 || NM < 0 || T < 0) return ISTGCN_EINVAL;
  if (nnz_cap < 1 || nnz_cap > K * V * V) return ISTGCN_EINVAL;
  if (!istgcn_dtype_ok(dtype)) return ISTGCN_EINVAL;
  if (NM == 0 || T == 0) return ISTGCN_OK;
  {
    // dispatch override ISTGCN_GCN_RC=0: the round-2 kernels (A/B timing, one process per setting), read once
    static const bool rc_on = [] { const char* e = getenv("ISTGCN_GCN_RC"); return !e || atoi(e) != 0; }();
    if (rc_on && istgcn_gcn_wgrad_rc_ok(V, Cin, Cout, K, dtype))
      return istgcn_gcn_wgrad_rc(dy, x, A, dW, S, NM, T, V, Cin, Cout, K, dtype, grid_cap, ws, ws_floats, stream);
    if (rc_on && dtype == 0 && V <= 32 && V >= 20 && Cin >= 64 && Cin % 64 == 0 && Cout >= 64 && Cout % 64 == 0 && K <= 3)
      return istgcn_gcn_wgrad_rc_f32(dy, x, A, dW, S, NM, T, V, Cin, Cout, K, grid_cap, ws, ws_floats, stream);
  }
  TwgParams P{};
  P.dz = dy; P.g = x; P.dW = dW; P.A = A; P.S = S; P.nnz_cap = nnz_cap;
  P.NM = NM; P.Tin = T; P.Tz = T; P.V = V; P.Cin = Cin; P.Cout = Cout; P.ntaps = K; P.in_mul = 1;
  P.ws = ws_floats > 0 ? ws : nullptr; P.ws_slice = ws_floats;
  if (dtype == 0) return launch_agg_T<float>(P, grid_cap, (hipStream_t)stream);
  if (dtype == 2) return launch_agg_T<_Float16>(P, grid_cap, (hipStream_t)stream);
  return launch_agg_T<__bf16>(P, grid_cap, (hipStream_t)stream);
}

#ifdef ISTGCN_STAMP
extern "C" int istgcn_debug_stamps_wgrad(unsigned long long* out16, int reset) {
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_stamp_wg), 16 * sizeof(unsigned long long)) != hipSuccess) return ISTGCN_ELAUNCH;
  if (reset) {
    unsigned long long z[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_wg), z, sizeof(z)) != hipSuccess) return ISTGCN_ELAUNCH;
  }
  return ISTGCN_OK;
}
#endif
