// Temporal convolution, wave-specialised kernel with a LEAN memory role (round 4) for the trunk layers in 16-bit storage:
// same arithmetic, same packed weights, same LDS geometry and the same compute role (static k-structure) as tconv.hip --
//
//   out[n, m, v, o] = epi( sum_j sum_i Wf[j][o][i] * pre(in[n, in_mul*m + tap_off[j], v, i]) )
//
// (net/st_gcnold.py:165-175, the 15-tap fold of net/st_gcn_multi3_fix_3A_mstcn.py:160-180,212-215 and their data gradients) --
// but the four memory waves execute about a third of the instructions.  Why that matters: the stamps of round 3
// (profiles/r03_tconv_role_stamps.txt) had the compute waves WAITING for the memory waves 20-38 % of every forward
// launch (`commit`: 52 vector instructions per 16-byte vector) and of every data gradient (`store_pass`), and a wave
// retires one instruction per ~4 cycles whatever its SIMD partner does: ~650 instructions per (tile, chunk) item are
// 2600 cycles against 2304 cycles of matrix work at 64 channels.  What changed:
//   * `pre` (BatchNorm affine + ReLU) is six instructions per dword -- shift / and (unpack two bf16), two fma, one
//     v_cvt_pk_bf16_f32, one v_pk_max_i16 (ReLU on the packed pair: a negative float is a negative int16) -- written on
//     scalars so that the compiler cannot re-vectorise it element by element; rows outside the sequence are masked only
//     in tiles that touch a sequence edge (uniform branch);
//   * tile decode once per tile and stream (issue / commit / epilogue cursors advanced by adds), per-lane load offsets
//     fixed per tile (one add per load);
//   * the epilogue of a tile is SPREAD over the items of the next tile (EPP = 4 image rows per thread and item), so the
//     memory role's work per item is constant, the data gradient's `aux` rows are requested a whole `commit` ahead of
//     their use (they used to wait out a memory round trip per batch), and stores go through a buffer descriptor that
//     covers exactly the tile's rows (no row predicate);
//   * mode 1 accumulates sum(d) and sum(d * x) -- the centring and the 1/sigma of x-hat are applied to the two sums once
//     at the end (linear) -- and the ReLU mask is one fma + compare + select per element.
// Shapes: 16-bit storage, whole channel vectors, CC = 32 (two k-groups per tap), taps in threes, 256-row tiles, 64 / 128
// output channels per workgroup, out_mul = 1, modes 0 (forward) and 1 (data gradient).  Everything else: tconv.hip.
//
// hipcc-flags: -fno-slp-vectorize
// (the build passes these to hipcc for this file.  The SLP vectoriser pairs the memory role's scalar fp32 operations into
//  v_pk_fma_f32 / v_pk_add_f32; next to a wave issuing back-to-back MFMAs on the same SIMD a packed fp32 instruction
//  takes 22 cycles against 13 for a plain one (tools/valu_beside_mfma.hip), and the stamped kernel had `commit` at 5700
//  cycles per item with them, 2400 without: profiles/r04_tconv_lean_stamps.txt.)
#include "common.hpp"
#include "gcn_rc.hpp"     // rsrc_t / make_rsrc, pack2 / unpack2
#include "bn_tail.hpp"
#include "tconv_geom.hpp"
#include <cstdlib>

namespace {

using tconv_geo::NROLE;
using tconv_geo::UL;
using tconv_geo::TconvGeom;
constexpr int NTH = 2 * NROLE;
constexpr int MAX_TAPS = 16;
constexpr int EPP = 4;               // image rows per thread and item in the spread epilogue

struct TlParams {
  const void* in;
  const void* Wp;
  const float* bias;     // [Cout] or null
  const float* pre;      // [2][Cin] scale, shift or null
  const void* aux;       // mode 1: [NM][Tout][V][Cout]
  const float* maux;     // mode 1: [4][Cout] scale, shift, mean, rstd
  void* out;
  double* stats;         // [stats_rep][2][Cout] or null
  int NM, Tin, Tout, Mlog, V, Cin, Cout, ntaps;
  int in_mul, out_off, pre_relu, stats_rep;
  int tap_off[MAX_TAPS];
  int F, tiles_per_seq, total_tiles, nch, MTtot, min_off, Fin;
  unsigned tps_magic;
  int off_stat, off_u0, off_u1, off_o;
  BnTail tail;
  unsigned long long* dbg;   // experiment builds (-DISTGCN_TCONV_STAMP): cycle stamps of workgroup 0 (null otherwise)
};

#ifdef TL_X_NOMFMA          /* experiment build: the compute waves issue everything but the MFMAs (results wrong) */
#define TL_MMA(acc, a, b) { asm volatile("" :: "v"(a), "v"(b)); }
#else
#define TL_MMA(acc, a, b) mma_kgroup(acc, a, b)
#endif
#ifdef TL_X_PAD             /* experiment build: idle cycles in the compute wave after each MFMA (is the partner wave's VALU starved?) */
#define TL_PAD asm volatile("s_nop %0" :: "n"(TL_X_PAD));
#else
#define TL_PAD
#endif
#ifdef ISTGCN_TCONV_STAMP   /* experiment build: where the cycles of one compute wave and one memory wave of workgroup 0 go */
#define TSTAMP(i) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); tacc[i] += now_ - tlast; tlast = now_; }
#else
#define TSTAMP(i)
#endif

__device__ static inline void lds_barrier() {
  // LDS traffic of this wave retired, then the workgroup barrier; NOT __syncthreads() (its fence would drain the prefetch)
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// ReLU on two packed 16-bit floats (either format): sign-magnitude, so "negative" is the int16 sign bit
__device__ static inline uint32_t relu_pk(uint32_t w) {
  typedef short s16x2 __attribute__((ext_vector_type(2)));
  const s16x2 z = {0, 0};
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, w), z));
}

// two floats -> one packed pair with ONE conversion instruction (v_cvt_pk_bf16_f32; a pair built from two scalar casts
// compiles to two conversions and a v_perm_b32)
template <typename T> __device__ static inline uint32_t pk2(float a, float b);
template <> __device__ inline uint32_t pk2<__bf16>(float a, float b) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const f32x2 v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}
template <> __device__ inline uint32_t pk2<_Float16>(float a, float b) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const f32x2 v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2));
}

constexpr int US = 40;               // elements per staged row: CC + EPL (80 bytes: conflict-free 16-byte fragment reads)
constexpr int OS = 136;              // elements per image row: 128 channels + EPL

struct TileL {
  int n, m0, rows, in_rows, r_lo, r_hi;
  unsigned base;          // byte offset of the first staged row inside the sequence (wraps when the halo starts in front of it)
  bool valid, edge;
};

template <typename T, int MT, int MODE, int WM>
__global__ __launch_bounds__(NTH, 2) void tconv_lean_kernel(const TlParams P) {
  using E = Elem<T>;
  constexpr int EPL = 8, KGS = 16, CC = 32, NKG = 2;
  constexpr int TR = 256;
  // compute waves: WM channel groups x 4/WM row groups; a wave owns MT/WM channel tiles x 2*WM row tiles.  WM = 4 (one
  // channel tile per wave, every wave all 256 rows): each weight fragment is fetched ONCE per CU and step instead of by
  // both row groups -- the weight stream is vector-memory traffic (1 KB per wave instruction through a 64 B/clk path)
  static_assert(MT % WM == 0 && 4 % WM == 0, "wave layout");
  constexpr int MTW = MT / WM, NTW = 2 * WM;
  typedef typename E::frag frag_t;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* stat = reinterpret_cast<float*>(smem + P.off_stat);                 // [2][MT*32]
  float* bias_l = stat + 2 * MT * 32;                                        // [MT*32]
  float* pre_l = bias_l + MT * 32;                                           // [2][Cin]
  T* outs = reinterpret_cast<T*>(smem + P.off_o);                            // [TR][out_stride]

  const int tid = threadIdx.x, lane = tid & 63;
  // Roles by age: the MEMORY role runs on waves 0-3.  The two waves of a SIMD compete for its vector-issue port; the port
  // goes to the older wave (a workgroup's waves 4-7 are the younger half; s_setprio does not change it) -- measured in
  // isolation (tools/valu_beside_mfma.hip, profiles/r04_valu_beside_mfma.txt): next to a wave issuing back-to-back
  // MFMAs a YOUNGER vector wave gets one instruction per 13 cycles, an OLDER one per 6.5 (alone: 5.1), and the MFMA
  // wave runs at 32 cycles per MFMA either way.  The memory role is the one with the vector-ALU work (BatchNorm + ReLU
  // of the chunk, the epilogue sums); the compute role's stream is MFMAs + loads.
#ifndef TL_MEM_FIRST
#define TL_MEM_FIRST 1
#endif
  const bool is_compute = TL_MEM_FIRST ? tid >= NROLE : tid < NROLE;
  const int ltid = tid & (NROLE - 1), wave = ltid >> 6;
  const int V = P.V;
  const int mt0 = blockIdx.y * MT;
  const int cbase_blk = mt0 * 32;

  for (int c = tid; c < 2 * MT * 32; c += NTH) stat[c] = 0.f;
  for (int c = tid; c < MT * 32; c += NTH) bias_l[c] = P.bias ? P.bias[cbase_blk + c] : 0.f;
  for (int c = tid; c < 2 * P.Cin; c += NTH) {
    const int h = c / P.Cin;
    pre_l[c] = P.pre ? P.pre[c] : (h == 0 ? 1.f : 0.f);
  }

  const T* ing = reinterpret_cast<const T*>(P.in);
  const T* Wp = reinterpret_cast<const T*>(P.Wp);

  // XCD-affine persistent order (as tconv.hip): XCD x walks the contiguous tile range [x*chunk, (x+1)*chunk)
  const int G8 = gridDim.x >> 3;
  const int chunk = (P.total_tiles + 7) >> 3;
  const int xcd = blockIdx.x & 7, slot0 = blockIdx.x >> 3;
  const int slot_end = min(chunk, P.total_tiles - xcd * chunk);
  const int ntile_w = slot0 < slot_end ? (slot_end - slot0 + G8 - 1) / G8 : 0;
  const int nch = P.nch;
  const int total_items = ntile_w * nch;
  auto tile_of = [&](int k) __attribute__((always_inline)) {
    TileL t;
    t.valid = k < ntile_w;
    const int tile = xcd * chunk + slot0 + (t.valid ? k : 0) * G8;
    t.n = P.tiles_per_seq == 1 ? tile : (int)__umulhi((unsigned)tile, P.tps_magic);
    t.m0 = (tile - t.n * P.tiles_per_seq) * P.F;
    const int nf = min(P.F, P.Mlog - t.m0);
    t.rows = nf * V;
    const int fin0 = P.in_mul * t.m0 + P.min_off;                           // first staged input frame (may be < 0)
    t.in_rows = (P.in_mul * (nf - 1) + P.Fin - P.in_mul * (P.F - 1)) * V;   // frames actually needed
    t.r_lo = fin0 < 0 ? -fin0 * V : 0;
    t.r_hi = min(t.in_rows, (P.Tin - fin0) * V);
    t.base = (unsigned)(fin0 * V * P.Cin * (int)sizeof(T));
    t.edge = t.r_lo > 0 || t.r_hi < t.in_rows;
    return t;
  };
  lds_barrier();
#ifdef ISTGCN_TCONV_STAMP
  unsigned long long tacc[6] = {0, 0, 0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memtime();
#endif
#ifdef TL_X_PRIO            /* experiment build: issue priority per role (1: compute waves high, 2: memory waves high) */
  if ((TL_X_PRIO == 1) == is_compute) __builtin_amdgcn_s_setprio(3);
#endif

  if (is_compute) {
    // =========================================== compute waves ===========================================
    // (the static-k loop of tconv.hip: a six-step chunk = three taps x two k-groups; weight ring 5 steps ahead from L2,
    //  activation fragments 2 steps ahead from LDS; one MFMA, then the loads in its shadow)
    constexpr int DA = 6, DB = NTW >= 8 ? 2 : 3, PD = DB - 1;
    f32x16 acc[MTW][NTW];
    int brow[NTW];
    const int wr = wave / WM, wm = wave % WM;                // row group, channel group of this wave
    const int hoff = (lane >> 5) * EPL;
    const int nit = P.ntaps * NKG;
    const int roff0 = (P.tap_off[0] - P.min_off) * V;
    const int rstep = P.ntaps > 1 ? (P.tap_off[1] - P.tap_off[0]) * V : 0;
    const unsigned astr = (unsigned)(P.MTtot * 64 * EPL);    // elements between the fragments of consecutive steps
    const unsigned alim = (unsigned)(nch * nit) * astr;
    const T* abase = Wp + ((size_t)(mt0 + wm * MTW) * 64 + lane) * EPL;
    u32x4 a[DA][MTW], b[DB][NTW];
    unsigned ao = 0;
    auto load_as = [&](u32x4 (&dst)[MTW]) __attribute__((always_inline)) {
#ifdef TL_X_NOA             /* experiment build: no weight-fragment loads at all (results wrong) */
      return;
#endif
#ifdef TL_X_WL1             /* experiment build: every weight fragment is fragment 0 (L1 hits; results wrong) */
#pragma unroll
      for (int m = 0; m < MTW; ++m) dst[m] = *reinterpret_cast<const u32x4*>(abase + (unsigned)(m * 64 * EPL));
#else
#pragma unroll
      for (int m = 0; m < MTW; ++m) dst[m] = *reinterpret_cast<const u32x4*>(abase + ao + (unsigned)(m * 64 * EPL));
#endif
      const unsigned an = ao + astr;
      ao = an == alim ? 0u : an;
    };
    const T* us = reinterpret_cast<const T*>(smem + P.off_u0);
    const int ts1 = rstep * US;
    const int tsk[4] = {0, ts1, 2 * ts1, 3 * ts1};
    int soffc = 0;
    auto load_bs = [&](u32x4 (&dst)[NTW], int soff) __attribute__((always_inline)) {
#ifdef TL_X_NOB             /* experiment build: no activation-fragment reads (results wrong) */
      return;
#endif
#pragma unroll
      for (int tt = 0; tt < NTW; ++tt) dst[tt] = *reinterpret_cast<const u32x4*>(us + brow[tt] + soff);
    };
#pragma unroll
    for (int d = 0; d < DA - 1; ++d) load_as(a[d]);         // in flight while the first chunk is being staged
    // per-lane LDS element offset of each output row's fragment at tap offset 0 (row p of the tile = frame p / V, joint
    // p % V; input frame = in_mul * frame): the same for every tile.  Rows past a short last tile are NOT clamped: they read
    // rows of the staged buffer that nobody wrote for this tile (inside its UL * 64 rows) and produce garbage in their own
    // accumulator columns only, which the epilogue masks before the sums and whose stores fall outside the tile's descriptor
#pragma unroll
    for (int tt = 0; tt < NTW; ++tt) {
      const int p = wr * (32 * NTW) + tt * 32 + (lane & 31);
      const int f = p / V;
      brow[tt] = ((P.in_mul * f) * V + (p - f * V)) * US + hoff;
    }
    auto acc_init = [&]() __attribute__((always_inline)) {   // accumulators = conv bias (rows of the D tile = output channels)
#pragma unroll
      for (int m = 0; m < MTW; ++m) {
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias_l + (wm * MTW + m) * 32 + 8 * q4 + 4 * (lane >> 5));
#pragma unroll
          for (int tt = 0; tt < NTW; ++tt)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) acc[m][tt][4 * q4 + jj] = b4[jj];
        }
      }
    };
    // One step = this step's MFMAs plus the loads of later steps (weights DA-1 steps ahead, activations PD steps ahead);
    // sched_group_barrier interleaves one MFMA with the loads in its shadow.  LB = false: no activation prefetch (the last
    // PD steps of an item: their targets lie past the item).  MM = false: loads only -- the MFMAs of an item's LAST step
    // are issued after the item barrier, behind the first fragment reads of the NEXT item (TL_MMAS): the operands of that
    // step are in registers (weight ring slot DA-1, activation ring slot DB-1, which the next item's prologue does not
    // touch), so the matrix pipe works through them while the new item's first fragments are on their way from LDS --
    // that latency used to be exposed once per item (~10 % of a 64-channel item).
#define TL_MMAS(D)                                                                                       \
        _Pragma("unroll") for (int m = 0; m < MTW; ++m)                                                  \
          _Pragma("unroll") for (int tt = 0; tt < NTW; ++tt) { TL_MMA(acc[m][tt], __builtin_bit_cast(frag_t, a[D][m]), __builtin_bit_cast(frag_t, b[(D) % DB][tt])); TL_PAD }
#define TL_STEP(D, LB, MM)                                                                               \
      {                                                                                                  \
        load_as(a[((D) + DA - 1) % DA]);                                                                 \
        if (LB) load_bs(b[((D) + PD) % DB], soffc + tsk[((D) + PD) >> 1] + (((D) + PD) & 1) * KGS);      \
        if (MM) {                                                                                        \
          TL_MMAS(D)                                                                                     \
          _Pragma("unroll") for (int i_ = 0; i_ < MTW * NTW; ++i_) {                                     \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                           \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                           \
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                           \
            __builtin_amdgcn_sched_group_barrier(0x006, 5, 0);                                           \
          }                                                                                              \
        }                                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                               \
      }
    static_assert(PD <= 2 && (DA - 1) % DB == DB - 1, "the deferred step's activation slot is the one the prologue leaves alone");
    const int nchunk = nit / DA;
    lds_barrier();                                          // item 0 staged (the memory waves' prologue)
    int ch = 0, k = 0;
    if (total_items > 0) {
      acc_init();
      soffc = roff0 * US;
#pragma unroll
      for (int d = 0; d < PD; ++d) load_bs(b[d], soffc + tsk[d >> 1] + (d & 1) * KGS);
    }
    TSTAMP(5)
    for (int it = 0; it < total_items; ++it) {
      for (int c = 0; c < nchunk - 1; ++c) {
        TL_STEP(0, true, true) TL_STEP(1, true, true) TL_STEP(2, true, true) TL_STEP(3, true, true) TL_STEP(4, true, true) TL_STEP(5, true, true)
        soffc += tsk[3];
      }
      TL_STEP(0, true, true) TL_STEP(1, true, true) TL_STEP(2, true, true) TL_STEP(3, true, true)
      TL_STEP(4, PD < 2, true) TL_STEP(5, false, false)
      TSTAMP(1)                                             // the steps (all but the last one's MFMAs)
      us = reinterpret_cast<const T*>(smem + (((it + 1) & 1) ? P.off_u1 : P.off_u0));
      lds_barrier();                                        // item done: this half of the tile buffer may be refilled
      TSTAMP(2)                                             // wait at the item barrier
      const bool tile_end = ++ch == nch;
      const bool more = it + 1 < total_items;
      if (tile_end) {
        ch = 0;
        ++k;
      }
      // (the row bases are loop invariants now; made opaque per item, or the compiler materialises every `base + tap offset`
      //  sum of the item in its own register -- ~20 of them -- and spills them around the loop: reloads with vmcnt(0)
      //  waits at every item start, i.e. a drained weight ring)
#pragma unroll
      for (int tt = 0; tt < NTW; ++tt) asm volatile("" : "+v"(brow[tt]));
      if (more) {                                           // the next item's first fragment reads ...
        soffc = roff0 * US;
#pragma unroll
        for (int d = 0; d < PD; ++d) load_bs(b[d], soffc + tsk[d >> 1] + (d & 1) * KGS);
      }
      __builtin_amdgcn_sched_barrier(0);
      TL_MMAS(5)                                            // ... and, in their shadow, the MFMAs of this item's last step
      __builtin_amdgcn_sched_barrier(0);
      TSTAMP(0)                                             // item start
      if (tile_end) {
        // ---- tile end: accumulators -> LDS output image (row-major, channels innermost) ----
#pragma unroll
        for (int tt = 0; tt < NTW; ++tt) {
          const int sr = wr * (32 * NTW) + tt * 32 + (lane & 31);
#pragma unroll
          for (int m = 0; m < MTW; ++m) {
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
              const int cl = (wm * MTW + m) * 32 + 8 * q4 + 4 * (lane >> 5);
              float v4[4] = {acc[m][tt][4 * q4], acc[m][tt][4 * q4 + 1], acc[m][tt][4 * q4 + 2], acc[m][tt][4 * q4 + 3]};
              store4(outs + sr * OS + cl, v4);
            }
          }
        }
        TSTAMP(3)                                           // accumulators -> image
        lds_barrier();                                      // image complete
        acc_init();
        TSTAMP(4)
      }
    }
#undef TL_STEP
#undef TL_MMAS
#ifdef ISTGCN_TCONV_STAMP
    if (P.dbg && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) { for (int i = 0; i < 6; ++i) P.dbg[i] = tacc[i]; P.dbg[6] = (unsigned long long)total_items; }
#endif
  } else {
    // =========================================== memory waves ============================================
    constexpr int Q = CC / EPL;                             // 4 channel vectors per staged row
    const int q = ltid & (Q - 1), r0 = ltid >> 2;
    constexpr int RS = NROLE / Q;                           // 64 rows per sweep of the 256 threads
    const int winrows = P.Fin * V;
    const unsigned seq_bytes = (unsigned)(P.Tin * V * P.Cin * (int)sizeof(T));
    const size_t seq_elems = (size_t)P.Tin * V * P.Cin;
    // byte offset of this thread's u-th vector inside a staged window; rows past the window's capacity are permanently
    // out of range (0x80000000 + any tile base stays above 2^30 > the sequence: no traffic)
    // (the LDS half-buffers of this kernel hold UL * RS rows, so every vector has a slot and no store is predicated)
    unsigned voff[UL], vofft[UL];
#pragma unroll
    for (int u = 0; u < UL; ++u) {
      const int r = r0 + u * RS;
      voff[u] = r < winrows ? (unsigned)((r * P.Cin + q * EPL) * (int)sizeof(T)) : 0x80000000u;
    }
    const unsigned lds0 = (unsigned)((r0 * US + q * EPL) * (int)sizeof(T));
    const unsigned ldss = (unsigned)(RS * US * (int)sizeof(T));

    // ---- issue stream: item (ki, chi) -> registers.  One buffer descriptor per sequence: rows behind the sequence read
    //      as zeros without touching memory; rows in FRONT of it get a far offset once per tile (0xC0000000 + base never
    //      wraps below the 2^30 the launcher guarantees the sequence to be shorter than). ----
    TileL ti = tile_of(0);
    int ki = 0, chi = 0;
    auto set_issue_tile = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int u = 0; u < UL; ++u) vofft[u] = (r0 + u * RS >= ti.r_lo) ? voff[u] : 0xC0000000u;
    };
    set_issue_tile();
    auto issue = [&](u32x4 (&R)[UL]) __attribute__((always_inline)) {
#ifdef TL_X_NOLOAD          /* experiment build: empty descriptor, the chunk loads touch no memory (results wrong) */
      const rsrc_t rs = make_rsrc(ing + (size_t)ti.n * seq_elems, 0u);
#else
      const rsrc_t rs = make_rsrc(ing + (size_t)ti.n * seq_elems, ti.valid ? seq_bytes : 0u);
#endif
      const unsigned base = ti.base + (unsigned)(chi * CC * (int)sizeof(T));
#pragma unroll
      for (int u = 0; u < UL; ++u) R[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, vofft[u] + base, 0, 0));
      if (++chi == nch) { chi = 0; ++ki; ti = tile_of(ki); set_issue_tile(); }
    };

    // ---- commit stream: registers of item (kc, chc) -> `pre` -> LDS tile ----
    TileL tc = tile_of(0);
    int kc = 0, chc = 0;
    auto commit = [&](u32x4 (&R)[UL], unsigned ubytes) __attribute__((always_inline)) {
#ifdef TL_X_NOCOMMIT        /* experiment build: the staged chunk is never transformed / written (results wrong) */
      if (false) {
#else
      if (tc.valid) {
#endif
        unsigned char* dst = smem + ubytes + lds0;
        auto sweep = [&](auto has_pre, auto relu, auto edge) __attribute__((always_inline)) {
          float sc[EPL], sh[EPL];
          if constexpr (decltype(has_pre)::value) {
            const int cb = chc * CC + q * EPL;
#pragma unroll
            for (int e4 = 0; e4 < EPL; e4 += 4) {
              const f32x4 s4 = *reinterpret_cast<const f32x4*>(pre_l + cb + e4);
              const f32x4 h4 = *reinterpret_cast<const f32x4*>(pre_l + P.Cin + cb + e4);
#pragma unroll
              for (int e = 0; e < 4; ++e) { sc[e4 + e] = s4[e]; sh[e4 + e] = h4[e]; }
            }
          }
#pragma unroll
          for (int u = 0; u < UL; ++u) {
            uint32_t w[4] = {R[u][0], R[u][1], R[u][2], R[u][3]};
#ifndef TL_X_NOXFORM        /* experiment build: the chunk is staged untransformed (results wrong) */
            if constexpr (decltype(has_pre)::value) {
#else
            if constexpr (false) {
#endif
#pragma unroll
              for (int d = 0; d < 4; ++d) {
                float lo, hi;
                unpack2<T>(w[d], lo, hi);
                lo = __builtin_fmaf(lo, sc[2 * d], sh[2 * d]);
                hi = __builtin_fmaf(hi, sc[2 * d + 1], sh[2 * d + 1]);
                uint32_t p = pk2<T>(lo, hi);
                if constexpr (decltype(relu)::value) p = relu_pk(p);
                w[d] = p;
              }
            }
            if constexpr (decltype(edge)::value) {
              const int r = r0 + u * RS;
              const uint32_t keep = (r >= tc.r_lo && r < tc.r_hi) ? 0xffffffffu : 0u;
#pragma unroll
              for (int d = 0; d < 4; ++d) w[d] &= keep;
            }
            const u32x4 o = {w[0], w[1], w[2], w[3]};
            *reinterpret_cast<u32x4*>(dst + u * ldss) = o;
          }
        };
        using yes = std::integral_constant<bool, true>;
        using no = std::integral_constant<bool, false>;
        if (P.pre) {
          if (P.pre_relu) { if (tc.edge) sweep(yes{}, yes{}, yes{}); else sweep(yes{}, yes{}, no{}); }
          else { if (tc.edge) sweep(yes{}, no{}, yes{}); else sweep(yes{}, no{}, no{}); }
        } else sweep(no{}, no{}, no{});                      // (rows outside the sequence were loaded as zeros and stay zeros)
      }
      if (++chc == nch) { chc = 0; ++kc; tc = tile_of(kc); }
    };

    // ---- epilogue stream: the output image of tile `pend` -> HBM, EPP image rows per thread and item ----
    constexpr int CW = MT * 32;                             // channels of the image
    constexpr int VPR = CW / EPL;                           // vectors per image row
    constexpr int RSTEP = NROLE / VPR;                      // rows per sweep
    constexpr int NR = TR / RSTEP;                          // image rows per thread
    constexpr int NPARTS = NR / EPP;
    static_assert(NR % EPP == 0, "whole parts");
    const int vq = ltid % VPR, prow = ltid / VPR;
    const unsigned img0 = (unsigned)((prow * OS + vq * EPL) * (int)sizeof(T));
    const unsigned imgs = (unsigned)(RSTEP * OS * (int)sizeof(T));
    const unsigned gof0 = (unsigned)((prow * P.Cout + vq * EPL) * (int)sizeof(T));     // inside the tile's rows
    const unsigned gofs = (unsigned)(RSTEP * P.Cout * (int)sizeof(T));
    float s1[EPL], s2[EPL];
#pragma unroll
    for (int jj = 0; jj < EPL; ++jj) { s1[jj] = 0.f; s2[jj] = 0.f; }
    float msc[EPL], msh[EPL];
    if constexpr (MODE == 1) {
      const int cg = cbase_blk + vq * EPL;
#pragma unroll
      for (int jj = 0; jj < EPL; ++jj) { msc[jj] = P.maux[cg + jj]; msh[jj] = P.maux[P.Cout + cg + jj]; }
    }
    TileL pend = tile_of(0);
    int ppart = NPARTS;                                     // nothing pending
    T* outg = reinterpret_cast<T*>(P.out);
    const T* auxg = reinterpret_cast<const T*>(P.aux);
    auto tile_elem0 = [&](const TileL& t) __attribute__((always_inline)) {
      return ((size_t)(t.n * P.Tout + t.m0 + P.out_off) * V) * P.Cout + cbase_blk;
    };
    // mode 1: the part's `aux` rows, requested at the top of the iteration (a zero-size descriptor when nothing is pending:
    // the same number of loads on every path keeps the compiler's vmcnt bookkeeping exact)
    auto aux_issue = [&](u32x4 (&AV)[EPP]) __attribute__((always_inline)) {
      if constexpr (MODE == 1) {
        const bool act = ppart < NPARTS;
        const rsrc_t rs = make_rsrc(auxg + tile_elem0(pend), act ? (unsigned)(pend.rows * P.Cout * (int)sizeof(T)) : 0u);
        const unsigned vo = gof0 + (unsigned)(ppart * EPP) * gofs;     // (all of the offset in the VGPR: that is what the range check sees)
#pragma unroll
        for (int e = 0; e < EPP; ++e)
          AV[e] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo + (unsigned)e * gofs, 0, 2 /* nt: read once */));
      }
    };
    auto epi_part = [&](u32x4 (&AV)[EPP]) __attribute__((always_inline)) {
      if (ppart >= NPARTS) return;
#ifdef TL_X_NOEPI           /* experiment build: the output image is never streamed out (results wrong) */
      ++ppart;
      return;
#endif
      const rsrc_t ro = make_rsrc(outg + tile_elem0(pend), (unsigned)(pend.rows * P.Cout * (int)sizeof(T)));
      const int i0 = ppart * EPP;
      const unsigned vo = gof0 + (unsigned)i0 * gofs;
      const unsigned char* img = smem + P.off_o + img0 + (unsigned)i0 * imgs;
      auto body = [&](auto masked) __attribute__((always_inline)) {
#pragma unroll
        for (int e = 0; e < EPP; ++e) {
          const u32x4 sv = *reinterpret_cast<const u32x4*>(img + e * imgs);
          uint32_t w[4] = {sv[0], sv[1], sv[2], sv[3]};
          if constexpr (decltype(masked)::value) {
            const uint32_t keep = (prow + (i0 + e) * RSTEP < pend.rows) ? 0xffffffffu : 0u;
#pragma unroll
            for (int d = 0; d < 4; ++d) w[d] &= keep;
          }
          if constexpr (MODE == 1) {
            const uint32_t g[4] = {AV[e][0], AV[e][1], AV[e][2], AV[e][3]};
#pragma unroll
            for (int d = 0; d < 4; ++d) {
              float glo, ghi, zlo, zhi;
              unpack2<T>(g[d], glo, ghi);
              unpack2<T>(w[d], zlo, zhi);
              const float olo = __builtin_fmaf(glo, msc[2 * d], msh[2 * d]) > 0.f ? zlo : 0.f;
              const float ohi = __builtin_fmaf(ghi, msc[2 * d + 1], msh[2 * d + 1]) > 0.f ? zhi : 0.f;
              s1[2 * d] += olo;
              s1[2 * d + 1] += ohi;
              s2[2 * d] = __builtin_fmaf(olo, glo, s2[2 * d]);
              s2[2 * d + 1] = __builtin_fmaf(ohi, ghi, s2[2 * d + 1]);
              w[d] = pk2<T>(olo, ohi);
            }
          } else {
#pragma unroll
            for (int d = 0; d < 4; ++d) {
              float lo, hi;
              unpack2<T>(w[d], lo, hi);
              s1[2 * d] += lo;
              s1[2 * d + 1] += hi;
              s2[2 * d] = __builtin_fmaf(lo, lo, s2[2 * d]);
              s2[2 * d + 1] = __builtin_fmaf(hi, hi, s2[2 * d + 1]);
            }
          }
          const u32x4 o = {w[0], w[1], w[2], w[3]};
          __builtin_amdgcn_raw_buffer_store_b128(o, ro, vo + (unsigned)e * gofs, 0, 0);        // rows >= pend.rows: dropped
        }
      };
      using yes = std::integral_constant<bool, true>;
      using no = std::integral_constant<bool, false>;
      if ((i0 + EPP) * RSTEP <= pend.rows) body(no{});      // every row of the part exists
      else if (i0 * RSTEP < pend.rows) body(yes{});         // the part straddles the tile's last row
      ++ppart;
    };

    // ---- the item loop.  While the compute waves are on item `it`, item it+1 goes registers -> LDS (other half of the
    //      tile buffer) and item it+2 HBM -> registers; behind them one part of the previous tile's image goes LDS -> HBM. ----
    u32x4 RA[UL], RB[UL];
    issue(RA);
    issue(RB);
    __builtin_amdgcn_sched_barrier(0);
    commit(RA, (unsigned)P.off_u0);
    lds_barrier();                                          // item 0 staged
    int ch = 0, kt = 0;
    auto iteration = [&](int it, u32x4 (&Rn)[UL], u32x4 (&Rf)[UL]) __attribute__((always_inline)) {    // Rn: item it+1, Rf: free -> item it+2
      u32x4 AV[EPP];
      aux_issue(AV);
      __builtin_amdgcn_sched_barrier(0);
      issue(Rf);                                            // (past the last item: empty descriptor, same number of loads)
      __builtin_amdgcn_sched_barrier(0);
      TSTAMP(0)
#ifdef ISTGCN_TCONV_STAMP    /* the wait for item it+1's loads, stamped apart from the transform (slot 4 = tile-end barrier is tiny) */
      if constexpr (MODE == 1) asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      TSTAMP(4)
#endif
      commit(Rn, (unsigned)(((it + 1) & 1) ? P.off_u1 : P.off_u0));     // (past the last item: tc.valid is false)
      __builtin_amdgcn_sched_barrier(0);
      TSTAMP(1)
      epi_part(AV);
      TSTAMP(2)
      lds_barrier();                                        // item `it` computed, item it+1 staged
      TSTAMP(3)
      if (++ch == nch) {                                    // tile end: take over the output image
        ch = 0;
        lds_barrier();                                      // image written by the compute waves
        TSTAMP(3)
        // (NPARTS <= nch, checked by the launcher: the previous image has been streamed out completely)
        pend = tile_of(kt++);
        ppart = 0;
      }
    };
    TSTAMP(5)
    for (int it = 0; it < total_items; it += 2) {
      iteration(it, RB, RA);
      if (it + 1 < total_items) iteration(it + 1, RA, RB);
    }
    while (ppart < NPARTS) {                                // the last tile's image
      u32x4 AV[EPP];
      aux_issue(AV);
      epi_part(AV);
    }
#ifdef ISTGCN_TCONV_STAMP
    if (P.dbg && blockIdx.x == 0 && blockIdx.y == 0 && ltid == 0) for (int i = 0; i < 6; ++i) P.dbg[8 + i] = tacc[i];
#endif

    // ---- BatchNorm partial sums: registers -> lanes sharing a channel vector -> LDS ----
    if (P.stats) {
#pragma unroll
      for (int jj = 0; jj < EPL; ++jj) {
        float a = s1[jj], b = s2[jj];
        if constexpr (MODE == 1) {                          // sum d * xhat = (sum d * x - mean * sum d) * rstd
          const int cg = cbase_blk + vq * EPL + jj;
          b = (b - P.maux[2 * P.Cout + cg] * a) * P.maux[3 * P.Cout + cg];
        }
#pragma unroll
        for (int msk = VPR; msk < 64; msk <<= 1) { a += __shfl_xor(a, msk); b += __shfl_xor(b, msk); }
        const int cl = vq * EPL + jj;
        if (lane < VPR) {
          atomicAdd(&stat[cl], a);
          atomicAdd(&stat[MT * 32 + cl], b);
        }
      }
    }
  }

  if (P.stats) {
    lds_barrier();
    double* dst = P.stats + (size_t)(blockIdx.x % P.stats_rep) * 2 * P.Cout;
    for (int c = tid; c < MT * 32; c += NTH) {
      atomic_add_f64(dst + cbase_blk + c, (double)stat[c]);
      atomic_add_f64(dst + P.Cout + cbase_blk + c, (double)stat[MT * 32 + c]);
    }
  }
  bn_tail_run(P.tail, gridDim.x * gridDim.y, reinterpret_cast<unsigned*>(smem));
}

template <typename T, int MT, int MODE, int WM>
int launch_lean(const TlParams& P, int grid_cap, int gy, size_t lds, hipStream_t stream) {
  auto kfn = tconv_lean_kernel<T, MT, MODE, WM>;
  static std::atomic<unsigned long long> optin{0};
  if (int ea = istgcn_lds_optin((const void*)kfn, optin)) return ea;
  int gx = (grid_cap > 0 ? grid_cap : istgcn_resident_blocks((const void*)kfn, NTH, lds)) / gy;
  gx = round_up(gx < 1 ? 1 : (gx > P.total_tiles ? P.total_tiles : gx), 8);      // XCD-affine order: multiple of 8
  ISTGCN_LAUNCH(kfn, dim3(gx, gy), dim3(NTH), lds, stream, P);
  ISTGCN_CHECK_LAUNCH();
  return ISTGCN_OK;
}

}  // namespace

// LDS layout of the lean kernel: the tables of tconv.hip, two staged-chunk buffers of UL * 64 rows each (a slot for every
// vector a memory-wave thread handles: no predicated LDS store), the output image.
struct LeanLds { int off_stat, off_u0, off_u1, off_o, lds; };
static LeanLds lean_lds(const tconv_geo::TconvGeom& G, int Cin) {
  LeanLds L;
  size_t off = 0;
  L.off_stat = (int)off;
  off += (size_t)(3 * G.MT * 32 + 2 * Cin) * 4;
  off = (off + 15) & ~(size_t)15;
  const size_t ubytes = (size_t)UL * 64 * G.us_stride * 2;
  L.off_u0 = (int)off; off += ubytes;
  L.off_u1 = (int)off; off += ubytes;
  L.off_o = (int)off; off += (size_t)256 * G.out_stride * 2;
  L.lds = (int)off;
  return L;
}

// Does the lean kernel serve this launch?  (Called by istgcn_tconv with the geometry it has already decided.)
bool tconv_lean_ok(const tconv_geo::TconvGeom& G, int mode, int Tin, int V, int Cin, int Cout, int ntaps, int out_mul, int dtype) {
  static const bool off = [] { const char* e = getenv("ISTGCN_TCONV_LEAN"); return e && atoi(e) == 0; }();
  if (off || dtype == 0 || mode > 1 || out_mul != 1) return false;
  if (G.NT != 2 || (G.MT != 2 && G.MT != 4) || G.CC != 32 || G.NKG != 2 || G.us_stride != US || G.out_stride != OS) return false;
  if (ntaps % 3 != 0 || Cin % 32 != 0 || Cout % (G.MT * 32) != 0) return false;
  if (G.nch < G.MT) return false;                           // NPARTS = MT parts of the image, one per item of the next tile
  if ((long long)Tin * V * Cin * 2 >= (1ll << 30)) return false;
  if (G.Fin * V > UL * 64 || lean_lds(G, Cin).lds > 160 * 1024) return false;
  return true;
}

int tconv_lean_launch(const void* in, const void* Wp, const float* bias, const float* pre, int pre_relu, const void* aux,
                      const float* maux, void* out, double* stats, int stats_rep, int mode, int NM, int Tin, int Tout,
                      int Mlog, int V, int Cin, int Cout, int ntaps, const int* tap_off, int in_mul, int out_off, int dtype,
                      int grid_cap, const tconv_geo::TconvGeom& G, const BnTail& tail, hipStream_t stream) {
  TlParams P{};
  P.in = in; P.Wp = Wp; P.bias = bias; P.pre = pre; P.aux = aux; P.maux = maux; P.out = out; P.stats = stats;
  P.NM = NM; P.Tin = Tin; P.Tout = Tout; P.Mlog = Mlog; P.V = V; P.Cin = Cin; P.Cout = Cout; P.ntaps = ntaps;
  P.in_mul = in_mul; P.out_off = out_off; P.pre_relu = pre_relu; P.stats_rep = stats_rep < 1 ? 1 : stats_rep;
  for (int j = 0; j < ntaps; ++j) P.tap_off[j] = tap_off[j];
  P.F = G.F; P.nch = G.nch; P.MTtot = G.MTtot; P.min_off = G.min_off; P.Fin = G.Fin;
  const LeanLds L = lean_lds(G, Cin);
  P.off_stat = L.off_stat; P.off_u0 = L.off_u0; P.off_u1 = L.off_u1; P.off_o = L.off_o;
  P.tiles_per_seq = ceil_div(Mlog, G.F);
  P.total_tiles = NM * P.tiles_per_seq;
  P.tps_magic = (unsigned)(((1ull << 32) + P.tiles_per_seq - 1) / (unsigned long long)P.tiles_per_seq);
  P.tail = tail;
#ifdef ISTGCN_TCONV_STAMP
  { const char* e_dbg = getenv("ISTGCN_TCONV_DBG_PTR"); P.dbg = e_dbg ? reinterpret_cast<unsigned long long*>(strtoull(e_dbg, nullptr, 0)) : nullptr; }
#endif
#ifndef TL_WM4
#define TL_WM4 2            /* compute-wave layout of the 128-channel tiles: 2 (two channel groups x two row groups) or 4 */
#endif
#define CASE(TT, MTv, MD) if (G.MT == MTv && mode == MD) return launch_lean<TT, MTv, MD, (MTv == 4 ? TL_WM4 : 2)>(P, grid_cap, G.gy, (size_t)L.lds, stream)
  if (dtype == 2) { CASE(_Float16, 2, 0); CASE(_Float16, 2, 1); CASE(_Float16, 4, 0); CASE(_Float16, 4, 1); }
  else { CASE(__bf16, 2, 0); CASE(__bf16, 2, 1); CASE(__bf16, 4, 0); CASE(__bf16, 4, 1); }
#undef CASE
  return ISTGCN_EINVAL;
}
