// Graph-convolution unit, forward: y = einsum('nkctv,kvw->nctw', conv1x1(x), A)  (+ per-joint bias term)
// computed aggregate-first:
//     xa[p=(t,w)][(k,i)] = sum_v A[k][v][w] * x[(t,v)][i]          LDS -> LDS (16-bit: MFMA, fp32: sparse VALU pass)
//     y[p][c]            = sum_(k,i) Wr[c][(k,i)] * xa[p][(k,i)]   MFMA 32x32, fp32 accumulate
// Replaces (reference file:line): net/utils/tgcn.py:76-89, net/utils/tgcn_multi3_fix_3A.py:76-92,
// net/utils/inceptionv2_gcn.py:64-89 (all variants fold into one effective adjacency A, host side),
// and -- with K=1, A=I and a frame stride -- the residual 1x1 strided Conv2d of st_gcnold.py:186-193.
// The same kernel run on dy with A^T and the transposed weights is the unit's data gradient.
//
// A workgroup owns tiles of F = floor(128/V) whole frames of one sequence (<= 128 rows of the NTVC tensor, contiguous
// in HBM) x up to 128 output channels (its grid.y block) and walks tiles in a grid-stride loop so that the adjacency
// tables, BatchNorm partial sums and the weight working set are amortised.  Output channels are the MFMA "row" axis,
// positions the "column" axis, so each lane ends up with 4 consecutive channels of one row per register quad.
//
// Wave specialisation (round 2; the scheme of tconv.hip).  Round 1 ran stage -> barrier -> aggregate -> barrier ->
// contract -> barrier -> epilogue in sequence on two 4-wave workgroups per CU: its SQ counters showed the MFMA pipe 15 %
// busy and half of all wave-cycles parked on memory or barriers.  Now ONE workgroup of EIGHT waves per CU:
//   waves 0-3 "compute": the channel contraction; weight fragments come from L2 through a register ring that runs
//             continuously across chunks and tiles, activation fragments two steps ahead;
//   waves 4-7 "memory":  the x chunk of the item after next is in flight in registers, the next item's chunk goes
//             registers -> LDS (double-buffered), the previous tile's output image goes LDS -> HBM (+ addend, BN sums);
//   all eight share the aggregation (LDS -> LDS), which is a latency chain per (frame, channel tile, k) unit.
// An ITEM is one (tile, input-channel chunk) pair.  Two barriers per item:
//     B1  chunk staged, previous contraction done     memory: store image, prefetch; then all: aggregate chunk -> xa
//     B2  xa complete                                  compute: contract (+ image at the     memory: next chunk -> LDS
//                                                               tile's last chunk)
// Measured with the ISTGCN_GCN_ABL switches (64->64, T=300, bf16; DESIGN.md has the table): the first wave-specialised
// build spent 40 us of 153 in the one-time setup (two passes of V dependent global loads per adjacency column) and
// ~45 scalar bookkeeping instructions per contraction step (ring position, chunk wrap, ghost steps); hence the LDS copy
// of A below and the round-based ring whose loads carry immediate offsets from two uniform base pointers.
#include "common.hpp"
// Diagnostic hooks (ablation masks whose results are WRONG, in-kernel cycle stamps with their debug buffer) exist only in
// experiment builds (-DISTGCN_EXPERIMENT through tools/build_variant.sh); the shipped library reads no such switch.
#ifdef ISTGCN_EXPERIMENT
#define X_ABL(P) ((P).abl)
#define X_DBG(P) ((P).dbg)
#else
#define X_ABL(P) 0
#define X_DBG(P) ((unsigned long long*)nullptr)
#endif

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
  int gx_div, gx_mod;    // gridDim.x = gx_div * tiles_per_seq + gx_mod (item cursors advance without dividing)
  int a_lds;             // the adjacency fits the work buffers: setup reads it from an LDS copy
  unsigned long long* dbg;  // diagnostic: per-phase cycle sums of workgroup 0 (ISTGCN_GCN_DBG)
  int abl;               // diagnostic ablation mask (ISTGCN_GCN_ABL; results are then wrong): 1 no aggregation, 2 no contraction,
                         // 256 return at once (launch cost), 512 return after the setup.  (The switches around the memory
                         // waves' loads and stores are gone: a uniform branch around memory operations inside the loop makes
                         // the compiler's wait counts path-dependent even when it is never taken.)
  int off_csr_v, off_csr_a, off_stat, off_rows, off_afrag, off_bterm, off_xs0, off_xs1, off_xa, off_o;  // LDS byte offsets
};

constexpr int TILE_ROWS = 128;
constexpr int NROLE = 256;           // threads per role (4 waves)
constexpr int NTH = 2 * NROLE;
constexpr int UL = 4;                // 16-byte vectors of a chunk per memory-wave thread (128 rows x <= 8 vectors / 256)

struct Tile { int n, t0, nf, rows; bool valid; };

// Depth of the weight-fragment ring in steps: DA-1 steps of matrix work must outlast an L2 round trip.  A step is
// MH*NTW = 4 MFMA groups with four channel tiles but only 2 (MT = 2) or 1 (MT = 1), so those rings are twice as deep
// (and their fragments are half as many per step: the same registers).  The geometry pads the contraction length to whole
// rounds of DA steps.
template <int MT> struct GcnRing { static constexpr int DA = MT >= 4 ? 6 : 12; };
inline int gcn_ring_steps(int MT) { return MT >= 4 ? 6 : 12; }

__device__ static inline void lds_barrier() {
  // LDS traffic of this wave retired, then the workgroup barrier.  NOT __syncthreads(): its fence drains vmcnt(0), which
  // would wait for the prefetches both roles deliberately keep in flight across barriers.
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <typename T, int MT, bool VEC_IN, bool VEC_OUT>
__global__ __launch_bounds__(NTH, 2) void gcn_fwd_kernel(const GcnFwdParams P) {
  using E = Elem<T>;
  constexpr int EPL = E::EPL;
  constexpr int KGS = E::KGS;
  typedef typename E::frag frag_t;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if (X_ABL(P) & 256) return;

  int* csr_off = reinterpret_cast<int*>(smem);                       // [K*V+1]
  unsigned char* csr_v = smem + P.off_csr_v;                         // [nnz_cap]
  float* csr_a = reinterpret_cast<float*>(smem + P.off_csr_a);       // [nnz_cap]
  float* stat = reinterpret_cast<float*>(smem + P.off_stat);         // [2][MT*32]
  unsigned char* row_f = smem + P.off_rows;                          // [128]
  unsigned char* row_w = row_f + TILE_ROWS;                          // [128]
  unsigned char* col_k = row_w + TILE_ROWS;                          // [K*V]
  unsigned char* col_w = col_k + P.K * P.V;                          // [K*V]
  T* afrag = reinterpret_cast<T*>(smem + P.off_afrag);               // 16-bit only: [K][2][64][8] MFMA fragments of A_k
  float* bterm_l = reinterpret_cast<float*>(smem + P.off_bterm);     // [V][BTS] bias term of this channel block
  constexpr int BTS = MT * 32 + 4;                                   // row stride: 4 banks apart (an unpadded row put all 32 joints on one bank)
  // the two halves of the x chunk buffer, [xs_rows][xs_stride] each: always smem + offset (a select between two POINTERS
  // makes the compiler lose the LDS address space and emit flat loads)
  auto xsbuf = [&](int half) __attribute__((always_inline)) {
    return reinterpret_cast<T*>(smem + (half ? P.off_xs1 : P.off_xs0));
  };
  T* xa = reinterpret_cast<T*>(smem + P.off_xa);                     // [128][xa_stride]
  T* outs = reinterpret_cast<T*>(smem + P.off_o);                    // [128][OSTR]
  constexpr int OSTR = MT * 32 + EPL;                                // image row stride in elements

  const int tid = (int)(threadIdx.x ^ ISTGCN_ROLE_FLIP);
  const int lane = tid & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);       // wave-uniform by construction; tell the compiler
  const bool is_compute = wave8 < 4;
  const int ltid = tid & (NROLE - 1), wave = wave8 & 3;
  const int V = P.V, K = P.K;
  const int KV = K * V;
  const int mt0 = blockIdx.y * MT;
  const int cbase_blk = mt0 * 32;
  const int Q = P.CCeff / EPL;            // channel vectors per chunk row (<= 8)
  const int NV = P.KKp / EPL;             // vectors per xa row (incl. zero padding)

  // ---- one-time setup: adjacency -> LDS scratch (ONE coalesced round trip; the work buffers are not in use yet), then
  //      the CSR lists of its columns, the MFMA fragments of A_k, row tables, stat accumulators, bias term.  (Reading A
  //      from global memory inside the per-column loops -- two passes of V dependent loads -- cost 30 us per launch.) ----
  float* Asc = reinterpret_cast<float*>(smem + P.off_xs0);
  if (P.a_lds) for (int idx = tid; idx < KV * V; idx += NTH) Asc[idx] = P.A[idx];
  if (tid == 0) csr_off[0] = 0;
  for (int c = tid; c < 2 * MT * 32; c += NTH) stat[c] = 0.f;
  for (int r = tid; r < TILE_ROWS; r += NTH) {
    int f = r / V;
    row_f[r] = (unsigned char)f;
    row_w[r] = (unsigned char)(r - f * V);
  }
  if (P.bterm) {
    for (int idx = tid; idx < V * MT * 32; idx += NTH) {
      const int w = idx / (MT * 32), c = idx - w * (MT * 32);
      bterm_l[w * BTS + c] = (cbase_blk + c < P.Cout) ? P.bterm[w * P.Cout + cbase_blk + c] : 0.f;
    }
  }
  __syncthreads();
  auto build_tables = [&](auto Aat) __attribute__((always_inline)) {
    for (int col = tid; col < KV; col += NTH) {
      int k = col / V, w = col - k * V, cnt = 0;
      col_k[col] = (unsigned char)k;
      col_w[col] = (unsigned char)w;
      for (int v = 0; v < V; ++v) cnt += (Aat((k * V + v) * V + w) != 0.f);
      csr_off[col + 1] = cnt;
    }
    __syncthreads();
    if (wave8 == 0) {
      // exclusive scan of the column counts, 64 columns per step (reads of a step precede its writes: one wave, in order)
      int carry = 0;
      for (int base = 0; base < KV; base += 64) {
        const int c = base + lane;
        const int v = c < KV ? csr_off[c + 1] : 0;
        int incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const int tsh = __shfl_up(incl, o);
          if (lane >= o) incl += tsh;
        }
        const int tot = __shfl(incl, 63);
        if (c < KV && c > 0) csr_off[c] = carry + incl - v;
        carry += tot;
      }
      if (lane == 0) {
        csr_off[KV] = carry;
        if (carry > P.nnz_cap && P.status) *P.status = 1;
      }
    }
    __syncthreads();
    for (int col = tid; col < KV; col += NTH) {
      int k = col / V, w = col - k * V, e = csr_off[col];
      for (int v = 0; v < V; ++v) {
        float a = Aat((k * V + v) * V + w);
        if (a != 0.f) {
          if (e < P.nnz_cap) { csr_v[e] = (unsigned char)v; csr_a[e] = a; }
          ++e;
        }
      }
    }
    if constexpr (sizeof(T) == 2) {
      // B-operand fragments of the adjacency for the MFMA aggregation: lane (w = lane&31, h = lane>>5), k-step s,
      // element j holds A[k][v = 16s + 8h + j][w] (zero outside the V x V block)
      for (int idx = tid; idx < K * 2 * 64; idx += NTH) {
        const int ln = idx & 63, sstep = (idx >> 6) & 1, k = idx >> 7;
        const int w = ln & 31, h = ln >> 5;
        frag_t fr;
#pragma unroll
        for (int j = 0; j < EPL; ++j) {
          const int v = 16 * sstep + 8 * h + j;
          fr[j] = E::from_f((v < V && w < V) ? Aat((k * V + v) * V + w) : 0.f);
        }
        *reinterpret_cast<frag_t*>(afrag + idx * EPL) = fr;
      }
    }
  };
  // two instantiations, not one accessor that selects between an LDS and a global pointer (that becomes a flat load)
  if (P.a_lds) build_tables([&](int idx) __attribute__((always_inline)) { return Asc[idx]; });
  else build_tables([&](int idx) __attribute__((always_inline)) { return P.A[idx]; });
  __syncthreads();                                                    // the scratch copy of A is dead from here
  {
    // both chunk halves start as zeros: the rows behind the 128 data rows (the 16-bit aggregation reads a 32-row range
    // per frame) are never written again; everything else is rewritten per item
    const int nvec = 2 * P.xs_rows * (P.xs_stride / EPL);
    T* x0 = xsbuf(0);                                                 // the halves are adjacent
    for (int idx = tid; idx < nvec; idx += NTH) {
      frag_t z;
      zero_frag<T>(z);
      *reinterpret_cast<frag_t*>(x0 + idx * EPL) = z;
    }
    // contraction padding columns of xa (K * CCeff is padded to whole ring rounds) must be finite: zeroed once
    if (NV > K * Q) {
      const int padv = NV - K * Q;
      for (int idx = tid; idx < TILE_ROWS * padv; idx += NTH) {
        const int r = idx / padv, c = idx - r * padv;
        frag_t z;
        zero_frag<T>(z);
        *reinterpret_cast<frag_t*>(xa + r * P.xa_stride + (K * Q + c) * EPL) = z;
      }
    }
  }
  __syncthreads();
  if (X_ABL(P) & 512) return;

  const T* xg = reinterpret_cast<const T*>(P.x);
  const T* Wp = reinterpret_cast<const T*>(P.Wp);
  T* yg = reinterpret_cast<T*>(P.y);
  const T* addg = reinterpret_cast<const T*>(P.addend);

  // ---- item cursors.  An ITEM is (tile, chunk); a workgroup's tiles are blockIdx.x + k * gridDim.x.  Every phase of the
  //      loop used to re-derive (sequence, first frame, rows) of its item by two integer divisions -- ~40 vector
  //      instructions each, several times per item and role.  A cursor is advanced with a handful of scalar operations
  //      instead: gridDim.x = gx_div * tiles_per_seq + gx_mod (host side). ----
  struct Cur { int n, tq, ch; };
  const int nch = P.nch;
  const int ntile_w = (P.total_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;   // tiles of this workgroup
  const int total_items = ntile_w * nch;
  auto cur_first = [&]() __attribute__((always_inline)) {
    Cur c;
    c.n = (int)blockIdx.x / P.tiles_per_seq;
    c.tq = (int)blockIdx.x - c.n * P.tiles_per_seq;
    c.ch = 0;
    return c;
  };
  auto cur_next_tile = [&](Cur& c) __attribute__((always_inline)) {
    c.n += P.gx_div;
    c.tq += P.gx_mod;
    if (c.tq >= P.tiles_per_seq) { c.tq -= P.tiles_per_seq; ++c.n; }
  };
  auto cur_next = [&](Cur& c) __attribute__((always_inline)) {
    if (++c.ch == nch) { c.ch = 0; cur_next_tile(c); }
  };
  auto cur_nf = [&](const Cur& c) __attribute__((always_inline)) {      // frames of the tile; 0 beyond the last tile
    return c.n < P.NM ? min(P.F, P.Tlog - c.tq * P.F) : 0;
  };

  // Wave decomposition of the contraction: with an even number of channel tiles the four compute waves split 2 (64-row
  // halves) x 2 (channel-tile halves), so a weight fragment feeds TWO MFMAs -- one fragment per MFMA asks the vector
  // L1 for 128 B/clk per CU, twice what it delivers.  (Odd MT: one 32-row slab and all tiles per wave.)
  constexpr bool MSPLIT = (MT % 2) == 0;
  constexpr int MH = MSPLIT ? MT / 2 : MT;           // channel tiles per wave
  constexpr int NTW = MSPLIT ? 2 : 1;                // 32-row tiles per wave

  // ---------------- aggregation xs -> xa, shared by all eight waves (w8 = 0..7) ----------------
  // lane constants of the 16-bit path (addresses are lane part + a wave-uniform part per unit)
  const int a_grp = lane >> 4, a_h = a_grp >> 1, a_cblk = (a_grp & 1) * 16, a_q4 = (lane & 15) >> 2, a_pp = lane & 3;
  const int a_w = lane & 31;
  const int a_src = (8 * a_h + a_q4) * P.xs_stride + a_cblk + 4 * a_pp;       // element offset in the chunk (k-step 0, low half)
  const int a_dst = a_w * P.xa_stride + 4 * (lane >> 5);                      // element offset in xa (frame 0, k 0, channel tile 0)
  const int CT = (P.CCeff + 31) >> 5;
  const int NC = CT * K;                                                      // unit columns (channel tile, partition)
  const unsigned rcpK = (65536u + K - 1) / K;                                 // c / K for c < 256 as (c * rcpK) >> 16
  auto aggregate = [&](const T* xs, int nf, int w8) __attribute__((always_inline)) {
    if (X_ABL(P) & 1) return;
    if constexpr (sizeof(T) == 2) if (V <= 32) {
      // 16-bit: aggregation on the matrix cores.  Per unit (column c = (channel tile ct, partition k), frame f):
      // D[i][w] = sum_v x[(f,v)][i] * A_k[v][w], x^T read straight from the row-major tile with ds_read_b64_tr_b16 (rows
      // v beyond the frame multiply zero adjacency rows and are kept finite by the memory waves), A_k fragments from LDS;
      // the VALU version of this pass cost ~2500 instructions per wave and tile (conversions + addressing) against 24
      // MFMAs of real work.  The phase is bound by INSTRUCTION ISSUE (SQ counters: ~2200 vector + ~2200 scalar
      // instructions per item and workgroup when every unit decoded its own (c, f) and re-read the A_k fragments), so
      // the work is cut into TASKS = (column, half of the frames): a task reads its two A_k fragments once and then walks
      // its frames with five address increments each, three stages (LDS reads / MFMAs / convert + write) one frame
      // apart.  Twelve task slots per round: two per compute wave (one short half and one long half each), one per
      // memory wave, which enters this phase late, behind its commit and stores.
      const int T2 = NC * 2;
      const int h0 = nf >> 1;                               // frames [0, h0) and [h0, nf)
      const int sstep = V * P.xs_stride, dstep = V * P.xa_stride;
      auto task = [&](int t) __attribute__((always_inline)) {
        const int c = t >> 1, hf = t & 1;
        const int f0 = hf ? h0 : 0, n = hf ? nf - h0 : h0;
        if (n <= 0) return;
        const int ct = (int)(((unsigned)c * rcpK) >> 16), kk = c - ct * K;
        const T* af = afrag + kk * (2 * 64 * EPL) + lane * EPL;
        const frag_t b0 = *reinterpret_cast<const frag_t*>(af);
        const frag_t b1 = *reinterpret_cast<const frag_t*>(af + 64 * EPL);
        const bool full = P.CCeff - ct * 32 >= 32;
        // channels of the chunk at or above this lane's first quad (quad g is channel i0 + 8g); adjacency columns
        // w >= V have nothing to write
        const int lim = a_w < V ? P.CCeff - ct * 32 - 4 * (lane >> 5) : 0;
        int srce = f0 * sstep + ct * 32 + a_src;
        int dste = f0 * dstep + kk * P.CCeff + ct * 32 + a_dst;
        auto rd = [&](frag_t& x0, frag_t& x1) __attribute__((always_inline)) {
          const T* r0 = xs + srce;
          x0 = tr_pair<T>(r0, r0 + 4 * P.xs_stride);
          x1 = tr_pair<T>(r0 + 16 * P.xs_stride, r0 + 20 * P.xs_stride);
          srce += sstep;
        };
        auto mm = [&](const frag_t& x0, const frag_t& x1, f32x16& d) __attribute__((always_inline)) {
#pragma unroll
          for (int r = 0; r < 16; ++r) d[r] = 0.f;
          mma_kgroup(d, x0, b0);
          mma_kgroup(d, x1, b1);
        };
        auto put = [&](const f32x16& d) __attribute__((always_inline)) {
          if (full) {
            if (lim > 0) {
#pragma unroll
              for (int g = 0; g < 4; ++g) {
                float v4[4] = {d[4 * g], d[4 * g + 1], d[4 * g + 2], d[4 * g + 3]};
                store4(xa + dste + 8 * g, v4);
              }
            }
          } else {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              if (8 * g < lim) {
                float v4[4] = {d[4 * g], d[4 * g + 1], d[4 * g + 2], d[4 * g + 3]};
                store4(xa + dste + 8 * g, v4);
              }
            }
          }
          dste += dstep;
        };
        frag_t xa0, xa1, xb0, xb1;
        f32x16 d0, d1;
        rd(xa0, xa1);
        if (n > 1) rd(xb0, xb1);
        __builtin_amdgcn_sched_barrier(0);
        mm(xa0, xa1, d0);
        __builtin_amdgcn_sched_barrier(0);
        for (int i = 0; i < n; i += 2) {
          if (i + 2 < n) rd(xa0, xa1);
          __builtin_amdgcn_sched_barrier(0);
          if (i + 1 < n) mm(xb0, xb1, d1);
          __builtin_amdgcn_sched_barrier(0);
          put(d0);
          __builtin_amdgcn_sched_barrier(0);
          if (i + 1 >= n) break;
          if (i + 3 < n) rd(xb0, xb1);
          __builtin_amdgcn_sched_barrier(0);
          if (i + 2 < n) mm(xa0, xa1, d0);
          __builtin_amdgcn_sched_barrier(0);
          put(d1);
          __builtin_amdgcn_sched_barrier(0);
        }
      };
      if (w8 < 4) {
        for (int base = 0; base < T2; base += 12) {
          if (base + w8 < T2) task(base + w8);
          if (base + 7 - w8 < T2) task(base + 7 - w8);
        }
      } else {
        for (int t = 4 + w8; t < T2; t += 12) task(t);
      }
      return;
    }
    // sparse aggregation.  A wave owns adjacency columns col = w8, w8+8, ... (their compressed lists are wave-uniform:
    // no divergence, LDS broadcast reads); lanes span (frame, channel vector).  Rows >= rows are never written:
    // they only feed output rows that are never stored.
    const int npair = nf * Q;
    for (int col = w8; col < KV; col += 8) {
      const int kk = col_k[col], w = col_w[col];
      const int e0 = csr_off[col], e1 = min(csr_off[col + 1], P.nnz_cap);
      for (int pr = lane; pr < npair; pr += 64) {
        const int f = pr / Q, q = pr - f * Q;
        const T* xrow = xs + (f * V) * P.xs_stride + q * EPL;
        float sum[EPL];
#pragma unroll
        for (int j = 0; j < EPL; ++j) sum[j] = 0.f;
        for (int e = e0; e < e1; ++e) {
          const float av = csr_a[e];
          const frag_t xv = *reinterpret_cast<const frag_t*>(xrow + csr_v[e] * P.xs_stride);
#pragma unroll
          for (int j = 0; j < EPL; ++j) sum[j] += av * E::to_f(xv[j]);
        }
        frag_t o;
#pragma unroll
        for (int j = 0; j < EPL; ++j) o[j] = E::from_f(sum[j]);
        *reinterpret_cast<frag_t*>(xa + (f * V + w) * P.xa_stride + kk * P.CCeff + q * EPL) = o;
      }
    }
  };

  unsigned long long tacc[6] = {0, 0, 0, 0, 0, 0}, tlast = 0;
#define STAMP(i) if (X_DBG(P)) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); tacc[i] += now_ - tlast; tlast = now_; }

  if (is_compute) {
    // =========================================== compute waves ===========================================
    const int ph = MSPLIT ? (wave & 1) : wave, mh = MSPLIT ? (wave >> 1) : 0;
    // The weight ring.  One step = one k-group = MH*NTW MFMA groups; a ROUND is DA steps and an item is a whole number
    // of rounds (the geometry pads K*CCeff to that).  Slot d holds step d of the current round; step d of a round
    // issues the load of the slot just freed: step DA-1 of this round (d = 0) or step d-1 of the NEXT round -- whatever
    // chunk or tile that belongs to -- so every fragment is requested DA-1 steps ahead and the ring never drains, not
    // even across the aggregation phase.  Addresses are two wave-uniform bases (this round, next round) + the lane's
    // 32-bit offset + an immediate: the per-step bookkeeping of the first version (position, wrap, ghost steps: ~45
    // scalar instructions against four MFMAs) is now ~10 scalar instructions per round.
    constexpr int DA = GcnRing<MT>::DA;
    constexpr int DB = 3;                                  // activation ring: DB-1 steps ahead; divides DA
    constexpr int PD = DB - 1;
    constexpr unsigned FRAGB = 64 * EPL * sizeof(T);       // bytes of one fragment (1 KB)
    f32x16 acc[MH][NTW];
    // fragments live in the rings as four raw dwords (loop-carried arrays of 8 x 16-bit vectors get scalarised and
    // re-packed element by element); they become MFMA operands by a bit cast
    u32x4 a[DA][MH], b[DB][NTW];
    const int nit = P.NKG;                                 // steps per item
    const int R = nit / DA;                                // rounds per item
    const size_t chstride_b = (size_t)P.MTtot * P.NKG * FRAGB;      // bytes between the weights of consecutive chunks
    const char* wblk = reinterpret_cast<const char*>(Wp) + (size_t)(mt0 + mh * MH) * P.NKG * FRAGB;   // wave-uniform
    unsigned voff[MH];
#pragma unroll
    for (int m = 0; m < MH; ++m) voff[m] = (unsigned)lane * 16u + (unsigned)m * (unsigned)P.NKG * FRAGB;
    auto ldw = [&](const char* base, int step, u32x4 (&dst)[MH]) __attribute__((always_inline)) {
#pragma unroll
      for (int m = 0; m < MH; ++m) dst[m] = *reinterpret_cast<const u32x4*>(base + step * FRAGB + (size_t)voff[m]);
    };
    int ch_n = 0, r_n = 0;                                 // chunk / round of `nxt`
    auto advance = [&]() __attribute__((always_inline)) {
      if (++r_n == R) { r_n = 0; if (++ch_n == nch) ch_n = 0; }
      return wblk + (size_t)ch_n * chstride_b + (size_t)r_n * (DA * FRAGB);
    };
    const char* cur = wblk;
#pragma unroll
    for (int d = 0; d < DA - 1; ++d) {                     // in flight while the first chunk is being staged
      ldw(cur, d, a[d]);
      // keep the issue order slot 0, 1, ...: the loop's waits count loads YOUNGER than the slot they need, and the
      // compiler takes the minimum over the loop's entry paths -- a reordered prologue turned them all into vmcnt(1)
      __builtin_amdgcn_sched_barrier(0);
    }
    const char* nxt = advance();
    const T* brow[NTW];
#pragma unroll
    for (int tt = 0; tt < NTW; ++tt) brow[tt] = xa + (ph * 32 * NTW + tt * 32 + (lane & 31)) * P.xa_stride + (lane >> 5) * EPL;
    // bias-term rows of this lane's positions (tile-independent: position p is joint p mod V of some frame)
    const float* brp[NTW];
#pragma unroll
    for (int tt = 0; tt < NTW; ++tt) brp[tt] = bterm_l + (int)row_w[ph * 32 * NTW + tt * 32 + (lane & 31)] * BTS + 4 * (lane >> 5) + mh * MH * 32;
    lds_barrier();                                          // B1(0): item 0 staged (the memory waves' prologue)
    tlast = __builtin_amdgcn_s_memtime();
    Cur c = cur_first();
    int nf = cur_nf(c);
    for (int it = 0; it < total_items; ++it) {
      if (c.ch == 0) {
        // tile start: accumulators = bias term bterm[w][c] of their (row, channel)
        const int rows = nf * V;
#pragma unroll
        for (int tt = 0; tt < NTW; ++tt) {
          const int p = ph * 32 * NTW + tt * 32 + (lane & 31);
          const bool rowb = P.bterm && p < rows;
#pragma unroll
          for (int m = 0; m < MH; ++m) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
              if (rowb) b4 = *reinterpret_cast<const f32x4*>(brp[tt] + m * 32 + 8 * g);
#pragma unroll
              for (int j = 0; j < 4; ++j) acc[m][tt][4 * g + j] = b4[j];
            }
          }
        }
      }
      STAMP(0)                                              // tile start
      // ---------------- P1: aggregation ----------------
      aggregate(xsbuf(it & 1), nf, wave);
      STAMP(1)
      lds_barrier();                                        // B2: xa complete
      STAMP(2)
      // ---------------- P2: channel contraction ----------------
      // A step's loads and its MFMAs are independent: sched_group_barrier asks for them INTERLEAVED, one MFMA and then
      // the loads in its shadow, instead of a clump of load instructions during which the matrix pipe idles.
#pragma unroll
      for (int d = 0; d < PD; ++d)
#pragma unroll
        for (int tt = 0; tt < NTW; ++tt) b[d][tt] = *reinterpret_cast<const u32x4*>(brow[tt] + d * KGS);
#define GCN_STEP(D)                                                                                      \
      {                                                                                                  \
        if ((D) == 0) ldw(cur, DA - 1, a[DA - 1]); else ldw(nxt, (D) - 1, a[(D) - 1]);                   \
        _Pragma("unroll") for (int tt = 0; tt < NTW; ++tt)                                               \
          b[((D) + PD) % DB][tt] = *reinterpret_cast<const u32x4*>(bp[tt] + ((D) + PD) * KGS);           \
        _Pragma("unroll") for (int m = 0; m < MH; ++m)                                                   \
          _Pragma("unroll") for (int tt = 0; tt < NTW; ++tt) mma_kgroup(acc[m][tt], __builtin_bit_cast(frag_t, a[D][m]), __builtin_bit_cast(frag_t, b[(D) % DB][tt])); \
        _Pragma("unroll") for (int i_ = 0; i_ < MH * NTW; ++i_) {                                        \
          __builtin_amdgcn_sched_group_barrier(0x008, sizeof(T) == 4 ? 4 : 1, 0);                        \
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                             \
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                             \
          __builtin_amdgcn_sched_group_barrier(0x006, 3, 0);                                             \
        }                                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                               \
      }
      const int nround = (X_ABL(P) & 2) ? 0 : R;
      for (int r = 0; r < nround; ++r) {
        const T* bp[NTW];
#pragma unroll
        for (int tt = 0; tt < NTW; ++tt) bp[tt] = brow[tt] + r * (DA * KGS);
        GCN_STEP(0) GCN_STEP(1) GCN_STEP(2) GCN_STEP(3) GCN_STEP(4) GCN_STEP(5)
        if constexpr (DA == 12) { GCN_STEP(6) GCN_STEP(7) GCN_STEP(8) GCN_STEP(9) GCN_STEP(10) GCN_STEP(11) }
        cur = nxt;
        nxt = advance();
      }
#undef GCN_STEP
      STAMP(3)
      if (c.ch == nch - 1) {
        // tile end: accumulators -> LDS output image (row-major, channels innermost); the memory waves stream it out
        // during the next item's aggregation phase
#pragma unroll
        for (int tt = 0; tt < NTW; ++tt) {
          const int p = ph * 32 * NTW + tt * 32 + (lane & 31);
#pragma unroll
          for (int m = 0; m < MH; ++m) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const int cl = (mh * MH + m) * 32 + 8 * g + 4 * (lane >> 5);
              float v4[4] = {acc[m][tt][4 * g], acc[m][tt][4 * g + 1], acc[m][tt][4 * g + 2], acc[m][tt][4 * g + 3]};
              store4(outs + p * OSTR + cl, v4);
            }
          }
        }
      }
      cur_next(c);
      if (c.ch == 0) nf = cur_nf(c);
      STAMP(4)
      lds_barrier();                                        // B1(it+1): contraction done (+ image), next chunk staged
      STAMP(5)
    }
    if (X_DBG(P) && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0)
      for (int i = 0; i < 6; ++i) X_DBG(P)[i] = tacc[i];
  } else {
    // =========================================== memory waves ============================================
    // Everything a thread needs per item is a wave-uniform base (from the item's cursor) plus per-thread constants:
    // which rows / channel vector of a chunk it moves (UL slots), and which rows / channel vector of the image it stores.
    const int RS = NROLE / Q;                               // rows per sweep of the role's threads (>= 32)
    const int r0 = ltid / Q, q = ltid - r0 * Q;
    const bool tlive = r0 < RS;
    unsigned goff[UL];                                      // element offset of slot u's row in the tile (+ its channel vector)
    int xoff[UL];                                           // element offset in the LDS chunk, -1: no such row
    int xrow[UL];                                           // tile row of the slot (>= 128: none)
#pragma unroll
    for (int u = 0; u < UL; ++u) {
      const int r = r0 + u * RS;
      const bool ex = tlive && r < TILE_ROWS;
      xrow[u] = ex ? r : 1 << 20;
      goff[u] = ex ? (unsigned)(((int)row_f[r] * P.in_t_stride * V + (int)row_w[r]) * P.Cin + q * EPL) : 0u;
      xoff[u] = ex ? r * P.xs_stride + q * EPL : -1;
    }
    const size_t in_seq = (size_t)P.Tin * V * P.Cin;        // elements per input sequence
    const int in_tile = P.F * P.in_t_stride * V * P.Cin;    // elements between the first rows of consecutive tiles
    // ---- global loads of the cursor's item -> registers.  UNCONDITIONAL loads (dead slots read the tile's first vector
    //      -- the tensor's first for items beyond the end -- and are zeroed in `commit`): a constant number of loads per
    //      item lets the compiler wait for "all but the youngest UL" instead of draining the queue ----
    auto issue = [&](const Cur& c, u32x4 (&R)[UL]) __attribute__((always_inline)) {
      const int rows = cur_nf(c) * V;                       // 0 beyond the last tile
      const int cb = c.ch * P.CCeff;
      const T* base = xg + (rows > 0 ? (size_t)c.n * in_seq + (size_t)c.tq * in_tile + cb : 0);     // wave-uniform
      const bool qlive = q * EPL < P.Cin - cb;
#pragma unroll
      for (int u = 0; u < UL; ++u) {
        const bool live = qlive && xrow[u] < rows;
        const unsigned g = live ? goff[u] : 0u;
        if constexpr (VEC_IN) R[u] = *reinterpret_cast<const u32x4*>(base + g);
        else {
          frag_t v;
          zero_frag<T>(v);
          if (live) {
#pragma unroll
            for (int e = 0; e < EPL; ++e) if (cb + q * EPL + e < P.Cin) v[e] = base[g + e];
          }
          R[u] = __builtin_bit_cast(u32x4, v);
        }
      }
    };
    // ---- registers of the cursor's item -> LDS chunk (zeros in the rows behind the tile and in channel vectors beyond Cin) ----
    auto commit = [&](const Cur& c, u32x4 (&R)[UL], T* xs) __attribute__((always_inline)) {
      const int rows = cur_nf(c) * V;
      const bool qlive = q * EPL < P.Cin - c.ch * P.CCeff;
#pragma unroll
      for (int u = 0; u < UL; ++u) {
        if (xoff[u] >= 0) {
          frag_t v = __builtin_bit_cast(frag_t, R[u]);
          if (!(qlive && xrow[u] < rows)) zero_frag<T>(v);
          *reinterpret_cast<frag_t*>(xs + xoff[u]) = v;
        }
      }
    };
    // ---- output image -> HBM (+ addend) with the BatchNorm sums; a thread always copies out the same channel vector,
    //      so its sums stay in registers for the whole walk and are reduced once per workgroup ----
    constexpr int VPR = MT * 32 / EPL;                      // vectors per image row
    constexpr int RSTEP = NROLE / VPR;                      // rows per sweep
    constexpr int NR = TILE_ROWS / RSTEP;                   // rows per thread
    constexpr int UB = NR < 4 ? NR : 4;
    const int vq = ltid % VPR, pr0 = ltid / VPR;
    const int cg = cbase_blk + vq * EPL;
    unsigned ooff[NR];                                      // element offset of the thread's i-th row in the output tile
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const int p = pr0 + i * RSTEP;
      ooff[i] = (unsigned)(((int)row_f[p] * P.out_t_stride * V + (int)row_w[p]) * P.Cout + cg);
    }
    const T* img = outs + pr0 * OSTR + vq * EPL;
    const size_t out_seq = (size_t)P.Tout * V * P.Cout;
    const int out_tile = P.F * P.out_t_stride * V * P.Cout;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 st1[EPL / 2], st2[EPL / 2];                       // packed pairs: v_pk_add_f32 / v_pk_fma_f32
#pragma unroll
    for (int j = 0; j < EPL / 2; ++j) { st1[j] = f32x2{0.f, 0.f}; st2[j] = f32x2{0.f, 0.f}; }
    auto store_image = [&](int n, int tq) __attribute__((always_inline)) {
      if (cg >= P.Cout) return;
      const int rows = min(P.F, P.Tlog - tq * P.F) * V;
      const size_t tb = (size_t)n * out_seq + (size_t)tq * out_tile;           // wave-uniform
      T* yb = yg + tb;
      const T* ab = addg + tb;
#pragma unroll
      for (int i0 = 0; i0 < NR; i0 += UB) {
        if (pr0 + i0 * RSTEP < rows) {
          frag_t sv[UB], av[UB];
          bool ok[UB];
#pragma unroll
          for (int u = 0; u < UB; ++u) {
            ok[u] = pr0 + (i0 + u) * RSTEP < rows;
            sv[u] = *reinterpret_cast<const frag_t*>(img + (i0 + u) * RSTEP * OSTR);
            if (addg) {
              const unsigned g = ok[u] ? ooff[i0 + u] : ooff[i0];
              if constexpr (VEC_OUT) av[u] = *reinterpret_cast<const frag_t*>(ab + g);
              else {
#pragma unroll
                for (int j = 0; j < EPL; ++j) av[u][j] = (cg + j < P.Cout) ? ab[g + j] : E::from_f(0.f);
              }
            }
          }
#pragma unroll
          for (int u = 0; u < UB; ++u) {
            if (ok[u]) {
              frag_t o = sv[u];
              if (addg) {
#pragma unroll
                for (int j = 0; j < EPL; ++j) o[j] = E::from_f(E::to_f(sv[u][j]) + E::to_f(av[u][j]));
              }
#pragma unroll
              for (int j = 0; j < EPL / 2; ++j) {
                f32x2 fv = {E::to_f(o[2 * j]), E::to_f(o[2 * j + 1])};
                if (!VEC_OUT) { if (cg + 2 * j >= P.Cout) fv[0] = 0.f; if (cg + 2 * j + 1 >= P.Cout) fv[1] = 0.f; }
                st1[j] += fv;
                st2[j] += fv * fv;
              }
              if constexpr (VEC_OUT) *reinterpret_cast<frag_t*>(yb + ooff[i0 + u]) = o;
              else {
#pragma unroll
                for (int j = 0; j < EPL; ++j) if (cg + j < P.Cout) yb[ooff[i0 + u] + j] = o[j];
              }
            }
          }
        }
      }
    };

    // ---- the item loop.  Item j travels in register set j & 1 and lands in chunk-buffer half j & 1.  Per item `it`:
    //        P1  commit(it+1)   its loads were issued two items ago; half (it+1)&1 has been free since B2 of item it-1
    //            store image    of the tile that ended with item it-1
    //            this role's share of the aggregation
    //        B2
    //        P2  issue(it+3)    into the register set drained by the commit
    //        B1
    //      The memory counter retires in order, so the wait in `commit` also covers the image stores of the PREVIOUS item
    //      (one whole item old by then) but never the stores of this one or the youngest prefetch. ----
    u32x4 RA[UL], RB[UL];
    Cur c_it = cur_first();                                 // item it
    Cur c_cm = c_it;                                        // item it+1 (next commit)
    Cur c_is = c_it;                                        // item it+3 (next issue)
    issue(c_is, RA); cur_next(c_is);
    issue(c_is, RB); cur_next(c_is);
    __builtin_amdgcn_sched_barrier(0);
    commit(c_cm, RA, xsbuf(0));
    issue(c_is, RA); cur_next(c_is);
    cur_next(c_cm);
    lds_barrier();                                          // B1(0)
    tlast = __builtin_amdgcn_s_memtime();
    bool pending = false;
    int pend_n = 0, pend_tq = 0;
    auto iteration = [&](int it, u32x4 (&Rs)[UL]) __attribute__((always_inline)) {   // Rs: holds item it+1, refilled with it+3
      if (it + 1 < total_items) commit(c_cm, Rs, xsbuf((it + 1) & 1));
      __builtin_amdgcn_sched_barrier(0);
      STAMP(0)
      if (pending) { store_image(pend_n, pend_tq); pending = false; }
      __builtin_amdgcn_sched_barrier(0);
      STAMP(1)
      aggregate(xsbuf(it & 1), cur_nf(c_it), 4 + wave);
      STAMP(2)
      lds_barrier();                                        // B2
      STAMP(3)
      issue(c_is, Rs);                                      // (no run-time switch around the loads: a branch here would make the
                                                            //  compiler's wait counts path-dependent, i.e. conservative)
      __builtin_amdgcn_sched_barrier(0);
      if (c_it.ch == nch - 1) { pending = true; pend_n = c_it.n; pend_tq = c_it.tq; }
      cur_next(c_it); cur_next(c_cm); cur_next(c_is);
      STAMP(4)
      lds_barrier();                                        // B1(it+1)
      STAMP(5)
    };
    for (int it = 0; it < total_items; it += 2) {
      iteration(it, RB);
      if (it + 1 < total_items) iteration(it + 1, RA);
    }
    if (pending) store_image(pend_n, pend_tq);
    if (X_DBG(P) && blockIdx.x == 0 && blockIdx.y == 0 && ltid == 0)
      for (int i = 0; i < 6; ++i) X_DBG(P)[8 + i] = tacc[i];

    if (P.stats) {
#pragma unroll
      for (int j = 0; j < EPL; ++j) {
        float sa = st1[j / 2][j & 1], sb = st2[j / 2][j & 1];
#pragma unroll
        for (int msk = VPR; msk < 64; msk <<= 1) { sa += __shfl_xor(sa, msk); sb += __shfl_xor(sb, msk); }
        const int cl = vq * EPL + j;
        if (lane < VPR && cbase_blk + cl < P.Cout) {
          atomicAdd(&stat[cl], sa);
          atomicAdd(&stat[MT * 32 + cl], sb);
        }
      }
    }
  }
#undef STAMP

  if (P.stats) {
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

template <typename T, int MT>
int launch_mt(const GcnFwdParams& P0, int grid_cap, int gy, size_t lds, hipStream_t stream) {
  GcnFwdParams P = P0;
  constexpr int EPL = Elem<T>::EPL;
  const bool vin = (P.Cin % EPL) == 0, vout = (P.Cout % EPL) == 0;
#define GO(VI, VO)                                                                                          \
  do {                                                                                                      \
    auto kfn = gcn_fwd_kernel<T, MT, VI, VO>;                                                               \
    static std::atomic<unsigned long long> optin{0};                                                        \
    if (int ea_ = istgcn_lds_optin((const void*)kfn, optin)) return ea_;                                    \
    int gx = (grid_cap > 0 ? grid_cap : istgcn_resident_blocks((const void*)kfn, NTH, lds)) / gy;           \
    gx = gx < 1 ? 1 : (gx > P.total_tiles ? P.total_tiles : gx);                                            \
    P.gx_div = gx / P.tiles_per_seq; P.gx_mod = gx % P.tiles_per_seq;                                       \
    ISTGCN_LAUNCH(kfn, dim3(gx, gy), dim3(NTH), lds, stream, P);                                            \
  } while (0)
  if (vin && vout) GO(true, true);
  else if (vin) GO(true, false);
  else if (vout) GO(false, true);
  else GO(false, false);
#undef GO
  ISTGCN_CHECK_LAUNCH();
  return ISTGCN_OK;
}

// Tiling decision shared by the launcher and the geometry query (the host packs weights to match).  The chunk width is a
// function of (Cin, Cout, K, dtype) alone: the widest chunk whose V-independent buffers (two chunk halves, aggregated
// image, output image) leave 32 KB of the CU's 160 KB for the adjacency tables and the bias term.
struct GcnGeom { int CCeff, nch, KKp, NKG, MT, gy, MTtot; };

inline int gcn_geom(int Cin, int Cout, int K, int dtype, GcnGeom* G) {
  const int epl = dtype == 0 ? 4 : 8, esz = dtype == 0 ? 4 : 2, cc_max = dtype == 0 ? 32 : 64, kgs = 2 * epl;
  // at most 4 channel tiles per workgroup (accumulators + rings must stay within 256 VGPRs); wider layers use grid.y
  G->MT = Cout <= 32 ? 1 : Cout <= 64 ? 2 : 4;
  G->gy = ceil_div(Cout, G->MT * 32);
  G->MTtot = G->gy * G->MT;
  for (int cc = cc_max; cc >= epl; cc >>= 1) {
    const int cce = Cin >= cc ? cc : round_up(Cin, epl);
    const int kkp = round_up(K * cce, kgs * gcn_ring_steps(G->MT));       // whole ring rounds
    const size_t xs = dtype == 0 ? (size_t)2 * TILE_ROWS * (cce + epl) * esz : (size_t)2 * 160 * round_up(cce, 32) * esz;
    const size_t fixed = xs + (size_t)TILE_ROWS * (kkp + epl) * esz + (size_t)TILE_ROWS * (G->MT * 32 + epl) * esz;
    if ((fixed <= 128 * 1024 && kkp / epl <= 64) || cc == epl) {
      if (kkp / epl > 64) return ISTGCN_EINVAL;
      G->CCeff = cce; G->nch = ceil_div(Cin, cce); G->KKp = kkp; G->NKG = kkp / kgs;
      return ISTGCN_OK;
    }
  }
  return ISTGCN_EINVAL;
}

template <typename T>
int launch_T(GcnFwdParams& P, int grid_x_cap, hipStream_t stream) {
  using E = Elem<T>;
  constexpr int EPL = E::EPL;
  const int dtype = sizeof(T) == 4 ? 0 : 1;
  GcnGeom G;
  if (int rc = gcn_geom(P.Cin, P.Cout, P.K, dtype, &G)) return rc;
  P.CCeff = G.CCeff; P.nch = G.nch; P.KKp = G.KKp; P.NKG = G.NKG; P.MTtot = G.MTtot;
  if (P.CCeff / EPL > 2 * UL) return ISTGCN_EINVAL;       // 128 rows x Q vectors must fit the role's UL x 256 prefetch slots
  P.F = TILE_ROWS / P.V;
  P.tiles_per_seq = ceil_div(P.Tlog, P.F);
  P.total_tiles = P.NM * P.tiles_per_seq;
  const bool mfma_agg = sizeof(T) == 2 && P.V <= 32;
  P.xs_stride = mfma_agg ? round_up(P.CCeff, 32) : P.CCeff + EPL;
  P.xs_rows = mfma_agg ? (P.F - 1) * P.V + 32 : TILE_ROWS;
  if (P.xs_rows < TILE_ROWS) P.xs_rows = TILE_ROWS;
  P.xa_stride = P.KKp + EPL;
  P.out_stride = G.MT * 32 + EPL;
  size_t off = (size_t)(P.K * P.V + 1) * sizeof(int);
  off = (off + 15) & ~(size_t)15; P.off_csr_v = (int)off; off += P.nnz_cap;
  off = (off + 15) & ~(size_t)15; P.off_csr_a = (int)off; off += (size_t)P.nnz_cap * 4;
  off = (off + 15) & ~(size_t)15; P.off_stat = (int)off; off += (size_t)2 * G.MT * 32 * 4;
  off = (off + 15) & ~(size_t)15; P.off_rows = (int)off; off += 2 * TILE_ROWS + 2 * P.K * P.V;
  off = (off + 15) & ~(size_t)15; P.off_afrag = (int)off; off += sizeof(T) == 2 ? (size_t)P.K * 2 * 64 * 16 : 0;
  off = (off + 15) & ~(size_t)15; P.off_bterm = (int)off; off += P.bterm ? (size_t)P.V * (G.MT * 32 + 4) * 4 : 0;
  const size_t xsb = (((size_t)P.xs_rows * P.xs_stride * sizeof(T)) + 15) & ~(size_t)15;
  off = (off + 15) & ~(size_t)15; P.off_xs0 = (int)off; off += xsb;
  P.off_xs1 = (int)off; off += xsb;
  P.off_xa = (int)off; off += (((size_t)TILE_ROWS * P.xa_stride * sizeof(T)) + 15) & ~(size_t)15;
  P.off_o = (int)off; off += (size_t)TILE_ROWS * P.out_stride * sizeof(T);
  if (off > 160 * 1024) return ISTGCN_EINVAL;
  P.a_lds = (size_t)P.K * P.V * P.V * 4 <= off - (size_t)P.off_xs0 ? 1 : 0;
  if (P.total_tiles < 1) return ISTGCN_OK;
#ifdef ISTGCN_EXPERIMENT
  { const char* e = getenv("ISTGCN_GCN_ABL"); P.abl = e ? atoi(e) : 0; }
  if (getenv("ISTGCN_GCN_DBG")) {
    static unsigned long long* dbuf = nullptr;
    if (!dbuf) (void)hipMalloc(&dbuf, 16 * sizeof(unsigned long long));
    (void)hipMemsetAsync(dbuf, 0, 16 * sizeof(unsigned long long), stream);
    P.dbg = dbuf;
    int rc = G.MT == 1 ? launch_mt<T, 1>(P, grid_x_cap, G.gy, off, stream) : G.MT == 2 ? launch_mt<T, 2>(P, grid_x_cap, G.gy, off, stream)
                                                                                       : launch_mt<T, 4>(P, grid_x_cap, G.gy, off, stream);
    unsigned long long h[16];
    (void)hipMemcpyAsync(h, dbuf, sizeof(h), hipMemcpyDeviceToHost, stream);
    (void)hipStreamSynchronize(stream);
    fprintf(stderr, "gcn_fwd dbg Cin=%d Cout=%d items/wg~%d | compute: init %llu agg %llu B2 %llu contract %llu image %llu B1 %llu | memory: commit %llu store %llu issue %llu agg %llu B2 %llu B1 %llu\n",
            P.Cin, P.Cout, P.total_tiles * P.nch / 256, h[0], h[1], h[2], h[3], h[4], h[5], h[8], h[9], h[10], h[11], h[12], h[13]);
    return rc;
  }
#endif
  switch (G.MT) {
    case 1: return launch_mt<T, 1>(P, grid_x_cap, G.gy, off, stream);
    case 2: return launch_mt<T, 2>(P, grid_x_cap, G.gy, off, stream);
    default: return launch_mt<T, 4>(P, grid_x_cap, G.gy, off, stream);
  }
}

}  // namespace

// Round-1 kernel (gcn_fwd_small.hip), same arguments: serves fp32 storage (VALU aggregation either way; no LDS room for
// the wave-specialised layout at full chunk width).
extern "C" int istgcn_gcn_fwd_v1(const void* x, const float* A, const void* Wp, const float* bterm,
                                 const void* addend, void* y, double* stats, int stats_rep, int* status,
                                 int NM, int Tin, int Tout, int Tlog, int V, int Cin, int Cout, int K,
                                 int in_t_stride, int out_t_stride, int nnz_cap, int dtype, int grid_cap, void* stream);
extern "C" int istgcn_gcn_v1_geometry(int Cin, int Cout, int K, int dtype, int* CCeff, int* nch, int* KKp, int* MTtot, int* EPL);
// Register-chained kernel (gcn_rc.hip): 16-bit storage, 64/128/256 input channels, output channels a multiple of 64,
// K <= 4, V <= 32.  Its weight layout is a second section of the packed buffer (istgcn_gcn_rc_offset), so the choice can
// depend on launch-time arguments (V, stats + addend together) that the packer does not see.  ISTGCN_GCN_RC=0 disables it.
extern "C" int istgcn_gcn_rc_layout(int Cin, int Cout, int K, int dtype);
extern "C" long long istgcn_gcn_rc_offset(int Cin, int Cout, int K, int dtype);
extern "C" int istgcn_gcn_fwd_rc(const void* x, const float* A, const void* Wq, const float* bterm, const void* addend,
                                 void* y, double* stats, int stats_rep, int NM, int Tin, int Tout, int Tlog, int V,
                                 int Cin, int Cout, int K, int in_t_stride, int out_t_stride, int dtype, int grid_cap,
                                 void* stream);
extern "C" int istgcn_gcn_fwd_rc_f32(const void* x, const float* A, const void* Wq, const float* bterm, void* y, double* stats,
                                     int stats_rep, int NM, int Tin, int Tout, int Tlog, int V, int Cin, int Cout, int K,
                                     int in_t_stride, int out_t_stride, int grid_cap, void* stream);
static bool gcn_use_rc() {       // dispatch override ISTGCN_GCN_RC=0 (A/B timing: one process per setting), read once
  static const bool on = [] { const char* e = getenv("ISTGCN_GCN_RC"); return !e || atoi(e) != 0; }();
  return on;
}
static bool gcn_use_v1(int dtype) { return dtype == 0; }      // float32: the round-1 kernel (gcn_fwd_small.hip)

extern "C" int istgcn_gcn_fwd(const void* x, const float* A, const void* Wp, const float* bterm,
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
  // (the 3-channel first-layer form of the register-chained kernel has no addend path: the round-2 kernels serve that call)
  if (gcn_use_rc() && V <= 32 && !(stats && addend) && !(Cin == 3 && addend) && istgcn_gcn_rc_layout(Cin, Cout, K, dtype)) {
    const long long off = istgcn_gcn_rc_offset(Cin, Cout, K, dtype);
    if (off >= 0 && dtype != 0)
      return istgcn_gcn_fwd_rc(x, A, reinterpret_cast<const char*>(Wp) + (size_t)off * 2, bterm, addend, y, stats, stats_rep, NM, Tin,
                               Tout, Tlog, V, Cin, Cout, K, in_t_stride, out_t_stride, dtype, grid_cap, stream);
    // (float32 is bound by the matrix instruction: the 32-row frame tile costs 32 / V of the round-1 kernel's row work.
    //  Measured on config 3, V = 18: slower than round 1 -- 23.4 vs 19.8 ms per step; V = 25: 1.5-1.7x faster.)
    if (off >= 0 && dtype == 0 && !addend && V >= 20)
      return istgcn_gcn_fwd_rc_f32(x, A, reinterpret_cast<const char*>(Wp) + (size_t)off * 4, bterm, y, stats, stats_rep, NM, Tin, Tout,
                                   Tlog, V, Cin, Cout, K, in_t_stride, out_t_stride, grid_cap, stream);
  }
  if (gcn_use_v1(dtype))
    return istgcn_gcn_fwd_v1(x, A, Wp, bterm, addend, y, stats, stats_rep, status, NM, Tin, Tout, Tlog, V, Cin, Cout, K, in_t_stride,
                             out_t_stride, nnz_cap, dtype, grid_cap, stream);
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
extern "C" int istgcn_gcn_geometry(int Cin, int Cout, int K, int dtype, int* CCeff, int* nch, int* KKp,
                                   int* MTtot, int* EPL) {
  if (!istgcn_dtype_ok(dtype) || Cin < 1 || Cout < 1 || K < 1) return ISTGCN_EINVAL;
  if (gcn_use_v1(dtype)) return istgcn_gcn_v1_geometry(Cin, Cout, K, dtype, CCeff, nch, KKp, MTtot, EPL);
  GcnGeom G;
  if (int rc = gcn_geom(Cin, Cout, K, dtype, &G)) return rc;
  *CCeff = G.CCeff; *nch = G.nch; *KKp = G.KKp; *MTtot = G.MTtot; *EPL = dtype == 0 ? 4 : 8;
  return ISTGCN_OK;
}
