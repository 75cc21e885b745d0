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
  if (P.ntaps <= 9) return launch_JT<T, 9, false>(P, grid_cap, stream);
  return launch_JT<T, 15, false>(P, grid_cap, stream);
}

template <typename T>
int launch_agg_T(TwgParams& P, int grid_cap, hipStream_t stream) {
  P.min_off = 0;
  P.F = TR / P.V;
  P.Fin = P.F;
  P.tiles_per_seq = ceil_div(P.Tz, P.F);
  P.total_tiles = P.NM * P.tiles_per_seq;
  if (P.ntaps <= 1) return launch_JT<T, 1, true>(P, grid_cap, stream);
  if (P.ntaps <= 3) return launch_JT<T, 3, true>(P, grid_cap, stream);
  return launch_JT<T, 4, true>(P, grid_cap, stream);
}

}  // namespace

extern "C" int istgcn_tconv_wgrad(const void* dz, const void* g, const float* pre, int pre_relu, float* dW,
                                  float* dbias, int NM, int Tin, int Tz, int V, int Cin, int Cout, int ntaps,
                                  const int* tap_off, int in_mul, int dtype, int grid_cap, float* ws,
                                  long long ws_floats, void* stream) {
  if (!dz || !g || !dW || !tap_off) return ISTGCN_EINVAL;
  if (ntaps < 1 || ntaps > 15 || V < 1 || V > 128 || Cin < 1 || Cout < 1 || in_mul < 1 || NM < 0 || Tz < 0)
    return ISTGCN_EINVAL;
  if (!istgcn_dtype_ok(dtype)) return ISTGCN_EINVAL;
  if (NM == 0 || Tz == 0) return ISTGCN_OK;
  // the lean kernel (tconv_wgrad_lean.hip) for the 16-bit trunk shapes; a conv-bias gradient, when asked for (the training
  // step does not: functional.py), is a column-sum kernel of its own next to it
  if (twg_lean_ok(V, Cin, Cout, ntaps, tap_off, in_mul, dtype, Tin, Tz) && (!dbias || (Cout <= 256 && 256 % (Cout / 8) == 0))) {
    const int rc = twg_lean_launch(dz, g, pre, pre_relu, dW, NM, Tin, Tz, V, Cin, Cout, ntaps, tap_off, in_mul, dtype, grid_cap, ws,
                                   ws_floats, (hipStream_t)stream);
    if (rc == ISTGCN_OK && dbias) return twg_lean_dbias(dz, dbias, NM, Tz, V, Cout, dtype, (hipStream_t)stream);
    if (rc >= 0) return rc;      // (-1: the LDS plan does not fit -- decided before anything was launched)
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

// fewest joints for which the fp32 register-chained weight gradient (gcn_rc_f32.hip: 32-row frame tiles) is dispatched; below it
// the 32 / V padding of the frame tile costs more matrix work than the round-1 kernel's flattened 128-row tiles save elsewhere
#ifndef RCF32_WG_MINV
#define RCF32_WG_MINV 20
#endif
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
    if (rc_on && dtype == 0 && V <= 32 && V >= RCF32_WG_MINV && Cin >= 64 && Cin % 64 == 0 && Cout >= 64 && Cout % 64 == 0 && K <= 3)
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
