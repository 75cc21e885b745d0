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
// A workgroup owns TR = 128*NT output rows (whole frames of one sequence) x up to 128 output channels; the input rows
// it needs (with the tap halo) are staged per channel chunk into LDS once and then re-read at a row offset per tap, so
// the activation is fetched from HBM/L2 once per chunk, not once per tap.  Tiles of one sequence are kept on one XCD
// (grid-stride order below) so halo re-reads hit that XCD's L2.
//
// Wave specialisation (round 2).  The workgroup has EIGHT waves in two roles, one workgroup per CU:
//   waves 0-3  "compute": the MFMA loop over (taps x k-groups) of the staged chunk, weight fragments streamed from L2
//              through a register ring; at the end of a tile they drop their accumulators into the LDS output image;
//   waves 4-7  "memory": everything that touches HBM -- they keep the global loads of the chunk TWO items ahead in
//              registers (they own no accumulators, so the registers are free), apply the BatchNorm affine + ReLU and
//              write the chunk one item ahead into the other half of a double-buffered LDS tile, and they stream the
//              previous tile's output image to HBM with the epilogue math (bias is already in the accumulators; BN sums /
//              ReLU mask / residual) while the compute waves are already on the next tile.
// The two roles meet at one barrier per (tile, chunk) item (+ the hand-over of the output image at a tile end), so the
// matrix cores see back-to-back MFMA work whenever a chunk's matrix time exceeds its memory time (128/256-channel layers)
// and the HBM stream never waits for arithmetic otherwise (64-channel layers).  In the single-role kernel this replaces
// (round 1: two 4-wave workgroups per CU, each running stage -> barrier -> MFMA -> barrier -> epilogue in sequence) 40 %
// of the time was un-overlapped staging and epilogue.
#include "common.hpp"
#include "gcn_rc.hpp"     // rsrc_t / make_rsrc
#include "bn_tail.hpp"
#include "tconv_geom.hpp"
// Diagnostic ablations (results WRONG) exist only in experiment builds (-DISTGCN_EXPERIMENT, tools/build_variant.sh);
// the cycle stamps only under -DISTGCN_TCONV_STAMP.  The shipped library reads neither switch.
#ifdef ISTGCN_EXPERIMENT
#define X_ABL(P) ((P).abl)
#else
#define X_ABL(P) 0
#endif
#include <cstdlib>
#include <type_traits>


namespace {

using tconv_geo::NROLE;               // threads per role (4 waves)
using tconv_geo::UL;
using tconv_geo::TconvGeom;
using tconv_geo::tconv_geom;
constexpr int NTH = 2 * NROLE;       // compute waves 0-3, memory waves 4-7
constexpr int MAX_TAPS = 16;

struct TconvParams {
  const void* in;
  const void* Wp;
  const float* bias;     // [Cout] or null
  const float* pre;      // [2][Cin] scale, shift or null
  const void* aux;       // epilogue mode 1 / 2: [NM][Tout][V][Cout]
  const float* maux;     // mode 1: [4][Cout] scale, shift, mean, rstd; mode 2: [2][Cout] scale, shift or null
  void* out;
  double* stats;         // [stats_rep][2][Cout] or null
  int NM, Tin, Tout, Mlog, V, Cin, Cout, ntaps;
  int in_mul, out_mul, out_off, pre_relu, mode, stats_rep;
  int tap_off[MAX_TAPS];
  // derived on the host
  int F, tiles_per_seq, total_tiles, CC, nch, NKG, MTtot, min_off, Fin;
  // ceil(2^32 / d) for the two run-time divisors of the item decode (x / d == umulhi(x, magic) for x * d < 2^32, d > 1):
  // a division by a run-time value is ~40 instructions, and the memory waves decode two items per iteration
  unsigned tps_magic, nch_magic;
  int us_stride, out_stride, off_stat, off_u0, off_u1, off_o;
  int cin_pad;           // nch * CC: length of the LDS copies of the `pre` rows
  unsigned long long* dbg;   // experiment builds (-DISTGCN_TCONV_STAMP): cycle stamps of workgroup 0 (null otherwise)
  int abl;               // diagnostic ablation (ISTGCN_TCONV_ABL): 1 = no input loads, 2 = no MFMAs; results are then wrong
  BnTail tail;           // "last workgroup finalises" the BatchNorm behind these sums (bn_tail.hpp), when the caller armed it
};

struct Tile {
  int n, m0, nf, rows, fin0, in_rows, r_lo, r_hi;
  long long row0;        // first staged input row (frame fin0, may be negative) in rows of the NTVC tensor
  bool valid;
};

__device__ static inline void lds_barrier() {
  // LDS traffic of this wave retired, then the workgroup barrier.  NOT __syncthreads(): its fence drains vmcnt(0), which
  // would wait for the memory waves' prefetch (two items of global loads deliberately left in flight across barriers).
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

#ifdef ISTGCN_TCONV_X_NOMFMA   /* experiment build: the compute waves issue everything but the MFMAs */
constexpr bool TCONV_X_NOMFMA = true;
#else
constexpr bool TCONV_X_NOMFMA = false;
#endif
#ifdef ISTGCN_TCONV_STAMP   /* experiment build: where the cycles of one compute wave and one memory wave of workgroup 0 go */
#define TSTAMP(i) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); tacc[i] += now_ - tlast; tlast = now_; }
#else
#define TSTAMP(i)
#endif

template <typename T, int MT, int NT, bool VEC, int MODE, int WM, bool SK>
__global__ __launch_bounds__(NTH, 2) void tconv_kernel(const TconvParams P) {
  using E = Elem<T>;
  constexpr int EPL = E::EPL;
  constexpr int KGS = E::KGS;
  constexpr int TR = 128 * NT;
  constexpr int OW = 256 / (int)sizeof(T);             // channels per epilogue pass (256-byte rows of the output image)
  constexpr int NPASS = (MT * 32 + OW - 1) / OW;       // 1 for the 16-bit types, 2 for fp32 with 128 output channels
  constexpr int MPP = OW / 32;                         // channel tiles per pass
  // compute-wave layout: WM channel groups x 4/WM row groups.  WM = 2: a wave owns MT/2 channel tiles x 2*NT row tiles, so
  // a weight fragment (vector L1 / L2) feeds 2*NT MFMAs instead of NT -- the L1 delivers 64 B/clk per CU, which one
  // fragment per two MFMAs on all four SIMDs already saturates
  static_assert(true && MT % WM == 0, "channel tiles split evenly over the channel groups");
  constexpr int MTW = MT / WM, NTW = NT * WM;
  typedef typename E::frag frag_t;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned short* row_f = reinterpret_cast<unsigned short*>(smem);          // [TR]
  unsigned short* row_v = row_f + TR;                                        // [TR]
  float* stat = reinterpret_cast<float*>(smem + P.off_stat);                 // [2][MT*32]
  float* bias_l = stat + 2 * MT * 32;                                        // [MT*32] conv bias of this channel block
  float* pre_l = bias_l + MT * 32;                                           // [2][cin_pad] BatchNorm affine of the input
  // the two halves of the staged-chunk buffer, [Fin*V][us_stride] each.  Always formed as smem + offset: selecting between
  // two POINTERS makes the compiler lose the LDS address space and emit flat loads, which count on vmcnt AND lgkmcnt, return
  // out of order and force vmcnt(0) waits in front of every MFMA group (found in the ISA of the first wave-specialised build)
  auto ubuf = [&](int half) __attribute__((always_inline)) { return reinterpret_cast<T*>(smem + (half ? P.off_u1 : P.off_u0)); };
  T* outs = reinterpret_cast<T*>(smem + P.off_o);                            // [TR][out_stride]

  const int tid = (int)(threadIdx.x ^ ISTGCN_ROLE_FLIP), lane = tid & 63;
  const bool is_compute = tid < NROLE;
  const int ltid = tid & (NROLE - 1), wave = ltid >> 6;
  const int V = P.V;
  const int mt0 = blockIdx.y * MT;
  const int cbase_blk = mt0 * 32;

  for (int r = tid; r < TR; r += NTH) {
    int f = r / V;
    row_f[r] = (unsigned short)f;
    row_v[r] = (unsigned short)(r - f * V);
  }
  for (int c = tid; c < 2 * MT * 32; c += NTH) stat[c] = 0.f;
  // per-channel constants into LDS once: read from there they cost LDS latency and -- unlike a global load -- no wait on
  // the vector-memory counter, behind which the waves' prefetched loads are queued
  for (int c = tid; c < MT * 32; c += NTH) bias_l[c] = (P.bias && cbase_blk + c < P.Cout) ? P.bias[cbase_blk + c] : 0.f;
  for (int c = tid; c < 2 * P.cin_pad; c += NTH) {
    const int h = c / P.cin_pad, i = c - h * P.cin_pad;
    pre_l[c] = (P.pre && i < P.Cin) ? P.pre[h * P.Cin + i] : (h == 0 ? 1.f : 0.f);
  }

  const T* ing = reinterpret_cast<const T*>(P.in);
  const T* Wp = reinterpret_cast<const T*>(P.Wp);
  const T* auxg = reinterpret_cast<const T*>(P.aux);
  T* outg = reinterpret_cast<T*>(P.out);

  // XCD-affine persistent order: XCD x (= blockIdx.x % 8 under round-robin dispatch; speed only) walks the contiguous
  // tile range [x*chunk, (x+1)*chunk), its workgroups taking neighbouring tiles at each step.  The k-th tile of this
  // workgroup is tile0 + k*G8; an ITEM is one (tile, input-channel chunk) pair, item = k*nch + ch.
  const int G8 = gridDim.x >> 3;
  const int chunk = (P.total_tiles + 7) >> 3;
  const int xcd = blockIdx.x & 7, slot0 = blockIdx.x >> 3;
  const int slot_end = min(chunk, P.total_tiles - xcd * chunk);
  const int ntile_w = slot0 < slot_end ? (slot_end - slot0 + G8 - 1) / G8 : 0;
  const int nch = P.nch;
  const int total_items = ntile_w * nch;
  auto udiv = [](int x, int d, unsigned magic) __attribute__((always_inline)) { return d == 1 ? x : (int)__umulhi((unsigned)x, magic); };
  auto tile_of = [&](int k) __attribute__((always_inline)) {
    Tile t;
    t.valid = k < ntile_w;
    const int tile = xcd * chunk + slot0 + (t.valid ? k : 0) * G8;
    t.n = udiv(tile, P.tiles_per_seq, P.tps_magic);
    t.m0 = (tile - t.n * P.tiles_per_seq) * P.F;
    t.nf = min(P.F, P.Mlog - t.m0);
    t.rows = t.nf * V;
    t.fin0 = P.in_mul * t.m0 + P.min_off;                              // first staged input frame (may be < 0)
    t.in_rows = (P.in_mul * (t.nf - 1) + P.Fin - P.in_mul * (P.F - 1)) * V;   // frames actually needed
    t.r_lo = t.fin0 < 0 ? -t.fin0 * V : 0;
    t.r_hi = min(t.in_rows, (P.Tin - t.fin0) * V);
    t.row0 = (long long)(t.n * P.Tin + t.fin0) * V;
    return t;
  };
  lds_barrier();

#ifdef ISTGCN_X_PRIO        /* experiment build: issue priority per role (1: compute waves high, 2: memory waves high) */
  if ((ISTGCN_X_PRIO == 1) == is_compute) __builtin_amdgcn_s_setprio(3);
#endif
#ifdef ISTGCN_TCONV_STAMP
  unsigned long long tacc[6] = {0, 0, 0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memtime();
#endif
  if (is_compute) {
    // =========================================== compute waves ===========================================
    // One wave per SIMD issues all the MFMAs, so nothing may stall it: the weight fragments come from L2 (500-900 cycles)
    // through a register ring DA steps deep that runs CONTINUOUSLY over the items of the workgroup -- the fragment
    // sequence of a tile is periodic (it does not depend on the staged data), so the ring keeps prefetching across the
    // item barriers and tile ends; the activation fragments (LDS, ~100 cycles) are read one step ahead inside an item.
    // (Round 1 re-primed a 3-4 deep ring per chunk: with two workgroups per CU the other workgroup covered the bubbles
    // and the too-shallow ring; with one wave per SIMD they were the kernel.)
    // Shape of the loop, chosen so that the compiler's own s_waitcnt bookkeeping stays exact (a data-dependent number of
    // loads on any path degrades every wait to vmcnt(0) = one exposed L2 round trip per step): an item is padded to
    // nitp = a multiple of DA steps, every (padded) step issues exactly MTW ring loads -- ghost steps and steps past the
    // last item read fragment 0 -- and only the MFMAs are predicated.  With nitp a multiple of DA the ring slot of a step
    // is its position in the unrolled chunk, and the first steps of the next item are already in flight when an item ends.
    constexpr int DA = 6;                                   // weight ring: 5 steps ahead
    constexpr int DB = NTW >= 8 ? 2 : 3;                    // activation ring: DB-1 steps ahead (divides DA: static slots)
    constexpr int PD = DB - 1;
    f32x16 acc[MTW][NTW];
    int brow[NTW];                                          // element offset of the lane's fragment at tap offset 0
    const int wr = wave / WM, wm = wave % WM;               // row group, channel group of this wave
    const int hoff = (lane >> 5) * EPL;
    const int nit = P.ntaps * P.NKG;                       // steps per item; NKG is a power of two
    const int nitp = (nit + DA - 1) / DA * DA;
    const int lkg = 31 - __builtin_clz(P.NKG);
    const int roff0 = (P.tap_off[0] - P.min_off) * V;
    const int rstep = P.ntaps > 1 ? (P.tap_off[1] - P.tap_off[0]) * V : 0;
    const int period = nch * nit;
    const size_t astride = (size_t)P.MTtot * 64 * EPL;      // elements between the fragments of consecutive steps
    const T* abase = Wp + ((size_t)(mt0 + wm * MTW) * 64 + lane) * EPL;
    // fragments live in the rings as four raw dwords: loop-carried arrays of 8 x 16-bit vectors get scalarised and re-packed
    // element by element (20 v_perm_b32 per step in the first build); they become MFMA operands by a bit cast
    u32x4 a[DA][MTW], b[DB][NTW];
    // running positions (a handful of scalar adds / selects per step; recomputing tap, k-group and offsets from the step
    // number cost ~20 scalar instructions per 4-8 MFMAs, and the 64-channel layers are instruction-issue bound)
    size_t aoff = 0;                                        // element offset of the next real ring fragment (wraps at alimit)
    const size_t alimit = (size_t)period * astride;
    int ppos = 0;                                           // padded step (mod nitp) the next ring load is for
    auto load_a = [&](u32x4 (&dst)[MTW]) __attribute__((always_inline)) {
      // branch-free: a ghost step (padding of an item to a multiple of DA steps) reads fragment 0 and does not advance
      const bool real = ppos < nit;
#ifdef ISTGCN_TCONV_X_WL1      /* experiment build: every weight fragment is fragment 0 (L1 hits) */
      const size_t off = 0;
#else
      const size_t off = real ? aoff : 0;
#endif
#pragma unroll
      for (int m = 0; m < MTW; ++m) dst[m] = *reinterpret_cast<const u32x4*>(abase + off + (size_t)m * 64 * EPL);
      const size_t an = aoff + (real ? astride : 0);
      aoff = an == alimit ? 0 : an;
      ppos = ppos + 1 == nitp ? 0 : ppos + 1;
    };
    const T* us = ubuf(0);
    // activation fragments of the NEXT not yet loaded step of the current item: running (k-group, offset) pair; past the
    // item's last step it stays there (the slot is reloaded with the same fragment and never used)
    const int tapstep = rstep * P.us_stride - (P.NKG - 1) * KGS;
    int sb = 0, kgb = 0, soffb = 0;                         // step, its k-group, its element offset in the tile
    auto load_b = [&](u32x4 (&dst)[NTW]) __attribute__((always_inline)) {
#ifndef ISTGCN_TCONV_X_NOB     /* experiment build: no activation fragment reads */
#pragma unroll
      for (int tt = 0; tt < NTW; ++tt) dst[tt] = *reinterpret_cast<const u32x4*>(us + brow[tt] + soffb);
#endif
      const bool adv = sb + 1 < nit;
      const bool wrap = kgb + 1 == P.NKG;
      soffb += adv ? (wrap ? tapstep : KGS) : 0;
      kgb = adv ? (wrap ? 0 : kgb + 1) : kgb;
      sb += adv ? 1 : 0;
    };
    // SK ("static k-structure"): two k-groups per tap and a whole number of six-step chunks per item (9 and 15 taps, every
    // trunk layer) -- a chunk is three taps x two k-groups, so the (tap, k-group) of a step is its position in the unrolled
    // chunk: the activation offset of a step is chunk base + {0,1,2,3} tap strides (four scalars) + an immediate, the weight
    // walk a 32-bit offset with one wrap test, and there are no ghost steps.  The generic path's running positions are
    // ~26 scalar instructions per step (tools/mfma_mix.hip: 6 % of the loop with eight MFMAs per step, twice that with four).
    // The activation prefetch of an item's last two steps runs one tap past the item: reads of staged-buffer / image bytes
    // inside the workgroup's LDS whose values are never used (the slots are reloaded by the next item's prologue).
    unsigned ao = 0;
    const unsigned astr = (unsigned)astride, alim = (unsigned)alimit;
    auto load_as = [&](u32x4 (&dst)[MTW]) __attribute__((always_inline)) {
#pragma unroll
      for (int m = 0; m < MTW; ++m) dst[m] = *reinterpret_cast<const u32x4*>(abase + ao + (unsigned)(m * 64 * EPL));
      const unsigned an = ao + astr;
      ao = an == alim ? 0u : an;
    };
    const int ts1 = rstep * P.us_stride;
    const int tsk[4] = {0, ts1, 2 * ts1, 3 * ts1};
    int soffc = 0;
    auto load_bs = [&](u32x4 (&dst)[NTW], int soff) __attribute__((always_inline)) {
#pragma unroll
      for (int tt = 0; tt < NTW; ++tt) dst[tt] = *reinterpret_cast<const u32x4*>(us + brow[tt] + soff);
    };
#pragma unroll
    for (int d = 0; d < DA - 1; ++d) { if constexpr (SK) load_as(a[d]); else load_a(a[d]); }   // in flight while the first chunk is being staged
    lds_barrier();                                          // item 0 staged (the memory waves' prologue)
    int ch = 0, k = 0;
    TSTAMP(5)
    for (int it = 0; it < total_items; ++it) {
      if (ch == 0) {
        // tile start: accumulators = conv bias (rows of the D tile = output channels: 4 consecutive per register quad),
        // and the per-lane LDS row of each output row at tap offset 0 (pad rows clamp to row 0: computed, never stored)
        const Tile t = tile_of(k);
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
#pragma unroll
        for (int tt = 0; tt < NTW; ++tt) {
          const int p = wr * (32 * NTW) + tt * 32 + (lane & 31);
          brow[tt] = (p < t.rows ? (P.in_mul * row_f[p]) * V + row_v[p] : 0) * P.us_stride + hoff;
        }
      }
      if constexpr (SK) {
        soffc = roff0 * P.us_stride;
#pragma unroll
        for (int d = 0; d < PD; ++d) load_bs(b[d], soffc + tsk[d >> 1] + (d & 1) * KGS);
        TSTAMP(0)                                           // tile start + ring prologue
#define TCONV_STEP_S(D)                                                                                  \
        {                                                                                                \
          load_as(a[((D) + DA - 1) % DA]);                                                               \
          load_bs(b[((D) + PD) % DB], soffc + tsk[((D) + PD) >> 1] + (((D) + PD) & 1) * KGS);            \
          _Pragma("unroll") for (int m = 0; m < MTW; ++m)                                                \
            _Pragma("unroll") for (int tt = 0; tt < NTW; ++tt) mma_kgroup(acc[m][tt], __builtin_bit_cast(frag_t, a[D][m]), __builtin_bit_cast(frag_t, b[(D) % DB][tt])); \
          _Pragma("unroll") for (int i_ = 0; i_ < MTW * NTW; ++i_) {                                     \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                           \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                           \
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                           \
            __builtin_amdgcn_sched_group_barrier(0x006, 5, 0);                                           \
          }                                                                                              \
          __builtin_amdgcn_sched_barrier(0);                                                             \
        }
        const int nchunk = nit / DA;
        for (int c = 0; c < nchunk; ++c) {
          TCONV_STEP_S(0) TCONV_STEP_S(1) TCONV_STEP_S(2) TCONV_STEP_S(3) TCONV_STEP_S(4) TCONV_STEP_S(5)
          soffc += tsk[3];
        }
#undef TCONV_STEP_S
      } else {
      sb = 0; kgb = 0; soffb = roff0 * P.us_stride;         // taps are an arithmetic progression (checked on the host)
#pragma unroll
      for (int d = 0; d < PD; ++d) load_b(b[d]);
      TSTAMP(0)                                             // tile start + ring prologue
      // One step = this step's MFMAs plus the loads of later steps (weights DA-1 steps ahead, activations two steps ahead),
      // which are independent of each other: sched_group_barrier asks for them to be INTERLEAVED -- one MFMA, then a few
      // of the other instructions in the 32-cycle shadow of that MFMA -- instead of a clump of ~50 address / load
      // instructions in front of the MFMA group, during which the matrix pipe idles (SQ counters of the first
      // wave-specialised build: MFMA pipe 45 % busy although the compute waves never waited on memory).
#define TCONV_STEP(D, PRED)                                                                              \
      {                                                                                                  \
        const int st_ = s0 + (D);                                                                        \
        load_a(a[((D) + DA - 1) % DA]);                                                                  \
        load_b(b[((D) + PD) % DB]);                                                                      \
        if (!TCONV_X_NOMFMA && (!(PRED) || st_ < nit)) {                                                 \
          _Pragma("unroll") for (int m = 0; m < MTW; ++m)                                                \
            _Pragma("unroll") for (int tt = 0; tt < NTW; ++tt) mma_kgroup(acc[m][tt], __builtin_bit_cast(frag_t, a[D][m]), __builtin_bit_cast(frag_t, b[(D) % DB][tt])); \
        }                                                                                                \
        _Pragma("unroll") for (int i_ = 0; i_ < MTW * NTW; ++i_) {                                       \
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   /* one MFMA */                            \
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   /* one LDS read */                        \
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   /* one global read */                     \
          __builtin_amdgcn_sched_group_barrier(0x006, 5, 0);   /* a few VALU / SALU */                   \
        }                                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                               \
      }
      const int nfull = nit / DA;
      int s0 = 0;
      for (int c = 0; c < nfull; ++c, s0 += DA) {
        TCONV_STEP(0, false) TCONV_STEP(1, false) TCONV_STEP(2, false) TCONV_STEP(3, false) TCONV_STEP(4, false) TCONV_STEP(5, false)
      }
      if (s0 < nit) {                                       // remainder chunk: ghost steps keep the ring's cadence
        TCONV_STEP(0, true) TCONV_STEP(1, true) TCONV_STEP(2, true) TCONV_STEP(3, true) TCONV_STEP(4, true) TCONV_STEP(5, true)
      }
#undef TCONV_STEP
      }
      TSTAMP(1)                                             // the steps
      us = ubuf((it + 1) & 1);
      lds_barrier();                                        // item done: this half of the tile buffer may be refilled
      TSTAMP(2)                                             // wait at the item barrier
      if (++ch == nch) {
        ch = 0;
        ++k;
        // ---- tile end: accumulators -> LDS output image (row-major, channels innermost), one hand-over per pass ----
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
#pragma unroll
          for (int tt = 0; tt < NTW; ++tt) {
            const int sr = wr * (32 * NTW) + tt * 32 + (lane & 31);        // image row = tile row
#pragma unroll
            for (int ml = 0; ml < MPP; ++ml) {
              const int mg = ps * MPP + ml;                // channel tile of this pass; held by channel group mg / MTW
              const int m = mg % MTW;
              if (mg < MT && mg / MTW == wm) {
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                  const int cl = ml * 32 + 8 * q4 + 4 * (lane >> 5);
                  float v4[4] = {acc[m][tt][4 * q4], acc[m][tt][4 * q4 + 1], acc[m][tt][4 * q4 + 2], acc[m][tt][4 * q4 + 3]};
                  store4(outs + sr * P.out_stride + cl, v4);
                }
              }
            }
          }
          TSTAMP(3)                                         // accumulators -> image
          lds_barrier();                                    // image of pass ps complete
          if (ps < NPASS - 1) lds_barrier();                // ... and streamed out by the memory waves: reusable
          TSTAMP(4)                                         // wait at the image barrier
        }
      }
    }
#ifdef ISTGCN_TCONV_STAMP
    if (P.dbg && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) { for (int i = 0; i < 6; ++i) P.dbg[i] = tacc[i]; P.dbg[6] = (unsigned long long)total_items; }
#endif
  } else {
    // =========================================== memory waves ============================================
    const int Q = P.CC / EPL;                               // 16-byte vectors per staged row: 2 or 4 (power of two)
    const int lq = 31 - __builtin_clz(Q);
    const int q = ltid & (Q - 1), r0 = ltid >> lq, RS = NROLE >> lq;
    // ---- global loads of item j -> registers: all UL loads of a thread are in flight together, and they stay in flight
    //      while the previous tile's output image is streamed out (the loads are only waited for in `commit`) ----
    // byte offset of this thread's u-th vector inside a staged window (row r0 + u*RS, channel vector q of the chunk)
    unsigned voff[UL];
#pragma unroll
    for (int u = 0; u < UL; ++u)      // (rows past the window's capacity are never staged: permanently out of range, no traffic)
      voff[u] = r0 + u * RS < P.Fin * V ? (unsigned)(((r0 + u * RS) * P.Cin + q * EPL) * (int)sizeof(T)) : 0x80000000u;
    auto issue = [&](int j, u32x4 (&R)[UL]) __attribute__((always_inline)) {
      const int k = udiv(j, nch, P.nch_magic), ch = j - k * nch;
      const Tile t = tile_of(k);
      const int cb = ch * P.CC;
      if constexpr (VEC) {
        // ONE buffer descriptor per sequence: window rows behind the sequence (and, by an explicit offset, in front of it)
        // fall outside the descriptor and read as zeros without touching memory;
        // past the last item (and under the no-loads ablation) the descriptor is empty.  A constant number of loads per item
        // is what lets the compiler wait for "all but the UL youngest" instead of for everything.  (Default cache policy
        // on purpose: the halo rows are re-read by the neighbouring tile from L2; streaming loads measured 14.30 vs 14.03
        // ms/step.)  A channel vector past Cin in the last chunk reads its row's neighbour; `commit` zeroes it.
        const unsigned bytes = (t.valid && X_ABL(P) != 1) ? (unsigned)(P.Tin * V * P.Cin * (int)sizeof(T)) : 0u;
        const rsrc_t rs = make_rsrc(ing + (size_t)t.n * P.Tin * V * P.Cin, bytes);
        const unsigned base = (unsigned)((t.fin0 * V * P.Cin + cb) * (int)sizeof(T));
        // (rows in FRONT of the sequence get an explicit far-out-of-range offset: their wrapped negative offsets end within
        //  16 bytes of 2^32, where offset + size overflows 32 bits -- not left to how the range check is evaluated)
#pragma unroll
        for (int u = 0; u < UL; ++u) {
          const unsigned off = r0 + u * RS >= t.r_lo ? voff[u] + base : 0x7ffffff0u;
          R[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
        }
      } else {
        const bool qlive = t.valid && (q * EPL < P.Cin - cb);
        const T* base = ing + (t.row0 * P.Cin + cb + q * EPL);
#pragma unroll
        for (int u = 0; u < UL; ++u) {
          const int r = r0 + u * RS;
          const bool live = qlive && r >= t.r_lo && r < t.r_hi;
          frag_t v;
          zero_frag<T>(v);
          if (live) {
            const T* p = base + (long long)r * P.Cin;
#pragma unroll
            for (int e = 0; e < EPL; ++e) if (cb + q * EPL + e < P.Cin) v[e] = p[e];
          }
          R[u] = __builtin_bit_cast(u32x4, v);
        }
      }
    };
    // ---- registers of item j -> BatchNorm affine + ReLU -> LDS tile (zero rows outside the sequence / chunk) ----
    // Branch-free per vector: every vector is transformed (pairs of elements on the packed-math pipe) and the rows that do
    // not exist are masked to zero afterwards -- they are rare (sequence edges), and a branch around the transform of each
    // vector was four branches and ~70 instructions per vector; which of (no transform | affine | affine + ReLU) applies
    // is decided once per item, outside the sweep
    auto commit = [&](int j, u32x4 (&R)[UL], T* us) __attribute__((always_inline)) {
      const int k = udiv(j, nch, P.nch_magic), ch = j - k * nch;
      const Tile t = tile_of(k);
      if (!t.valid) return;
      const int cb = ch * P.CC;
      const int c_lim = P.Cin - cb;
      const bool qlive = q * EPL < c_lim;
      float scv[EPL], shv[EPL];
#pragma unroll
      for (int e4 = 0; e4 < EPL; e4 += 4) {
        const f32x4 sc4 = *reinterpret_cast<const f32x4*>(pre_l + cb + q * EPL + e4);
        const f32x4 sh4 = *reinterpret_cast<const f32x4*>(pre_l + P.cin_pad + cb + q * EPL + e4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { scv[e4 + e] = sc4[e]; shv[e4 + e] = sh4[e]; }
      }
      auto sweep = [&](auto has_pre, auto relu) __attribute__((always_inline)) {
        typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int u = 0; u < UL; ++u) {
          const int r = r0 + u * RS;
          const bool live = qlive && r >= t.r_lo && r < t.r_hi;
          frag_t v = __builtin_bit_cast(frag_t, R[u]);
          if constexpr (decltype(has_pre)::value) {
#pragma unroll
            for (int e = 0; e < EPL; e += 2) {
              f32x2 fv = {E::to_f(v[e]), E::to_f(v[e + 1])};
              const f32x2 sc2 = {scv[e], scv[e + 1]}, sh2 = {shv[e], shv[e + 1]};
              fv = fv * sc2 + sh2;
              if constexpr (decltype(relu)::value) fv = __builtin_elementwise_max(fv, f32x2{0.f, 0.f});
              if (VEC || q * EPL + e < c_lim) v[e] = E::from_f(fv[0]);
              if (VEC || q * EPL + e + 1 < c_lim) v[e + 1] = E::from_f(fv[1]);
            }
          }
          u32x4 w = __builtin_bit_cast(u32x4, v);
          const unsigned keep = live ? 0xffffffffu : 0u;
          w &= u32x4{keep, keep, keep, keep};
          if (r < t.in_rows) *reinterpret_cast<u32x4*>(us + r * P.us_stride + q * EPL) = w;
        }
      };
      using yes = std::integral_constant<bool, true>;
      using no = std::integral_constant<bool, false>;
      if (P.pre) { if (P.pre_relu) sweep(yes{}, yes{}); else sweep(yes{}, no{}); }
      else sweep(no{}, no{});
    };

    // ---- output image of pass ps -> HBM with the epilogue math; per-channel sums stay in registers across tiles ----
    // thread -> (row, channel vector) map over the LIVE channels of a pass: with 64 output channels the image's 128-channel
    // rows are half empty, and a map over all 16 vectors left half of these threads idle in the sweep
    constexpr int CW = MT * 32 < OW ? MT * 32 : OW;         // live channels per pass
    constexpr int VPR = CW / EPL;                           // vectors per image row
    constexpr int RSTEP = NROLE / VPR;                      // rows per sweep
    const int vq = ltid % VPR;
    float st1[NPASS][EPL], st2[NPASS][EPL];
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps)
#pragma unroll
      for (int jj = 0; jj < EPL; ++jj) { st1[ps][jj] = 0.f; st2[ps][jj] = 0.f; }
    // modes 1 / 2 read `aux` (the producer's input / the residual) at the output positions: ALL rows of a thread are
    // fetched in one go and early -- before the chunk prefetch of the same iteration -- so the sweep itself only stores
    // (batches of four rows, each waiting out an HBM round trip, were most of the data-gradient kernel's memory side)
    constexpr int NR = TR / RSTEP;                          // image rows per thread
    // MEASURED AND SWITCHED OFF: with four channel tiles the compute waves' 128 accumulator registers set the kernel's
    // allocation and 64 more here spilled 100 registers; with two it fit (9 spills) but the 64-channel data gradient went
    // from 128 to 152 us -- the extra 16-32 KB of loads in flight per CU queue in front of the chunk prefetch.  Kept as a
    // compile-time switch; the sweep reads `aux` in batches of four rows.
    constexpr bool AUXPF = false && MODE >= 1 && VEC && MT <= 2;
    auto out_index = [&](const Tile& t, int pc, int cg) __attribute__((always_inline)) {
      if (P.out_mul == 1)                                   // the tile's output rows are then one contiguous run in HBM
        return ((size_t)(t.n * P.Tout + t.m0 * P.out_mul + P.out_off) * V + pc) * P.Cout + cg;
      return ((size_t)(t.n * P.Tout + (t.m0 + row_f[pc]) * P.out_mul + P.out_off) * V + row_v[pc]) * P.Cout + cg;
    };
    auto aux_prefetch = [&](const Tile& t, int ps, u32x4 (&AV)[AUXPF ? NR : 1]) __attribute__((always_inline)) {
      if constexpr (AUXPF) {
        const int cg = cbase_blk + ps * OW + vq * EPL;
        const bool col_live = (ps * OW + vq * EPL) < MT * 32 && cg < P.Cout;
#pragma unroll
        for (int i = 0; i < NR; ++i) {
          const int p = ltid / VPR + i * RSTEP;
          const bool ok = col_live && auxg && p < t.rows;
          const T* src = ok ? auxg + out_index(t, p, cg) : (auxg ? auxg : ing);   // dead slots: a valid address, ignored
          AV[i] = *reinterpret_cast<const u32x4*>(src);
        }
      }
    };
    auto store_pass = [&](const Tile& t, int ps, float (&s1)[EPL], float (&s2)[EPL], u32x4 (&AV)[AUXPF ? NR : 1]) __attribute__((always_inline)) {
      const bool dense_rows = P.out_mul == 1;               // the tile's output rows are then one contiguous run in HBM
      const size_t out_base = ((size_t)(t.n * P.Tout + t.m0 * P.out_mul + P.out_off) * V) * P.Cout;
      const int cg = cbase_blk + ps * OW + vq * EPL;
      const bool col_live = (ps * OW + vq * EPL) < MT * 32 && cg < P.Cout;
      float msc[EPL], msh[EPL], mmu[EPL], mrs[EPL];
#pragma unroll
      for (int jj = 0; jj < EPL; ++jj) { msc[jj] = MODE == 2 ? 1.f : 0.f; msh[jj] = 0.f; mmu[jj] = 0.f; mrs[jj] = 0.f; }
      if constexpr (MODE == 2) {
        if (P.maux && col_live) {
#pragma unroll
          for (int jj = 0; jj < EPL; ++jj)
            if (cg + jj < P.Cout) { msc[jj] = P.maux[cg + jj]; msh[jj] = P.maux[P.Cout + cg + jj]; }
        }
      }
      if constexpr (MODE == 1) {
        if (col_live) {
#pragma unroll
          for (int jj = 0; jj < EPL; ++jj) {
            if (cg + jj < P.Cout) {
              msc[jj] = P.maux[cg + jj]; msh[jj] = P.maux[P.Cout + cg + jj];
              mmu[jj] = P.maux[2 * P.Cout + cg + jj]; mrs[jj] = P.maux[3 * P.Cout + cg + jj];
            }
          }
        }
      }
      if (!col_live || X_ABL(P) == 4) return;
      // UB rows per batch: their LDS reads and (modes 1, 2) aux loads are all issued before the first is used
      constexpr int UB = 4;
      static_assert(NR % UB == 0 || NR < UB, "rows per thread come in whole batches");
#pragma unroll
      for (int i0 = 0; i0 < NR; i0 += UB) {
        const int p0 = ltid / VPR + i0 * RSTEP;
        if (p0 >= t.rows) break;
        frag_t sv[UB], av[UB];
        size_t g[UB];
        bool ok[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
          const int p = p0 + u * RSTEP;
          ok[u] = (i0 + u < NR) && p < t.rows;
          const int pc = ok[u] ? p : p0;
          if (dense_rows) g[u] = out_base + (size_t)pc * P.Cout + cg;
          else g[u] = ((size_t)(t.n * P.Tout + (t.m0 + row_f[pc]) * P.out_mul + P.out_off) * V + row_v[pc]) * P.Cout + cg;
          sv[u] = *reinterpret_cast<const frag_t*>(outs + pc * P.out_stride + vq * EPL);
          if constexpr (MODE >= 1) {
            if constexpr (AUXPF) {
              av[u] = __builtin_bit_cast(frag_t, AV[(i0 + u) < NR ? i0 + u : 0]);
            } else if (MODE == 1 || auxg) {
              // (read exactly once: streaming load, does not displace the weights / halo rows in L2)
              if (VEC) av[u] = __builtin_nontemporal_load(reinterpret_cast<const frag_t*>(auxg + g[u]));
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
    };

    // ---- the item loop.  While the compute waves are on item `it`, item it+1 goes registers -> LDS (other half of the
    //      tile buffer) and item it+2 HBM -> registers: two items of global loads are in flight all the time (these waves
    //      own no accumulators, the registers are free), so the stream never waits out a memory latency; behind them the
    //      previous tile's output image goes LDS -> HBM.  Item j lives in register set j & 1. ----
    u32x4 RA[UL], RB[UL];
    issue(0, RA);
    issue(1, RB);
    __builtin_amdgcn_sched_barrier(0);
    commit(0, RA, ubuf(0));
    lds_barrier();                                          // item 0 staged
    bool pending = false;
    Tile pend = tile_of(0);
    auto iteration = [&](int it, u32x4 (&Rn)[UL], u32x4 (&Rf)[UL]) __attribute__((always_inline)) {    // Rn: item it+1, Rf: free -> item it+2
      const int k = udiv(it, nch, P.nch_magic), ch = it - k * nch;
      u32x4 AV[AUXPF ? NR : 1];
      if (pending) aux_prefetch(pend, NPASS - 1, AV);       // oldest loads of the iteration: landed when the sweep starts
      __builtin_amdgcn_sched_barrier(0);
      issue(it + 2, Rf);                                    // (past the last item: dead slots, same number of loads)
      __builtin_amdgcn_sched_barrier(0);
      TSTAMP(0)
      if (it + 1 < total_items && X_ABL(P) != 4) commit(it + 1, Rn, ubuf((it + 1) & 1));
      __builtin_amdgcn_sched_barrier(0);
      TSTAMP(1)
      if (pending) { store_pass(pend, NPASS - 1, st1[NPASS - 1], st2[NPASS - 1], AV); pending = false; }
      TSTAMP(2)
      lds_barrier();                                        // item `it` computed, item it+1 staged
      TSTAMP(3)
      if (ch == nch - 1) {                                  // tile end: take over the output image
        const Tile t = tile_of(k);
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
          lds_barrier();                                    // image of pass ps written by the compute waves
          if (ps < NPASS - 1) {
            u32x4 AVp[AUXPF ? NR : 1];
            aux_prefetch(t, ps, AVp);
            store_pass(t, ps, st1[ps], st2[ps], AVp);
            lds_barrier();
          }
        }
        pending = true;
        pend = t;
        TSTAMP(4)
      }
    };
    TSTAMP(5)
    for (int it = 0; it < total_items; it += 2) {
      iteration(it, RB, RA);
      if (it + 1 < total_items) iteration(it + 1, RA, RB);
    }
    if (pending) {
      u32x4 AVl[AUXPF ? NR : 1];
      aux_prefetch(pend, NPASS - 1, AVl);
      store_pass(pend, NPASS - 1, st1[NPASS - 1], st2[NPASS - 1], AVl);
    }
#ifdef ISTGCN_TCONV_STAMP
    if (P.dbg && blockIdx.x == 0 && blockIdx.y == 0 && ltid == 0) for (int i = 0; i < 6; ++i) P.dbg[8 + i] = tacc[i];
#endif

    // ---- BatchNorm partial sums: registers -> lanes sharing a channel vector -> LDS ----
    if (P.stats) {
#pragma unroll
      for (int ps = 0; ps < NPASS; ++ps) {
#pragma unroll
        for (int jj = 0; jj < EPL; ++jj) {
          float a = st1[ps][jj], b = st2[ps][jj];
#pragma unroll
          for (int msk = VPR; msk < 64; msk <<= 1) { a += __shfl_xor(a, msk); b += __shfl_xor(b, msk); }
          const int cl = ps * OW + vq * EPL + jj;
          if (lane < VPR && cl < MT * 32 && cbase_blk + cl < P.Cout) {
            atomicAdd(&stat[cl], a);
            atomicAdd(&stat[MT * 32 + cl], b);
          }
        }
      }
    }
  }

  if (P.stats) {
    lds_barrier();
    double* dst = P.stats + (size_t)(blockIdx.x % P.stats_rep) * 2 * P.Cout;
    for (int c = tid; c < MT * 32; c += NTH) {
      if (cbase_blk + c < P.Cout) {
        atomic_add_f64(dst + cbase_blk + c, (double)stat[c]);
        atomic_add_f64(dst + P.Cout + cbase_blk + c, (double)stat[MT * 32 + c]);
      }
    }
  }
  bn_tail_run(P.tail, gridDim.x * gridDim.y, reinterpret_cast<unsigned*>(smem));
}

template <typename T, int MT, int NT, int WM>
int launch4(const TconvParams& P, int grid_cap, int gy, size_t lds, hipStream_t stream) {
  const bool vec = (P.Cin % Elem<T>::EPL) == 0 && (P.Cout % Elem<T>::EPL) == 0;
  // static k-structure (see the kernel): 16-bit storage, whole channel vectors, two k-groups per tap, taps in threes
  // Where it pays (measured, us generic -> SK at NM=128, bf16; profiles/r03_tconv_static_k.txt): the data gradient at every
  // width (64 ch 136 -> 126, 128 ch 183 -> 171, 256 ch 287 -> 270) and the 64-channel forward (117 -> 113).  NOT the
  // 128 / 256-channel forward (145 -> 152, 258 -> 272): there the memory waves' BatchNorm + ReLU transform is vector-ALU
  // work on the same SIMDs, the denser MFMA stream starves it (stamps: `commit` 2297 -> 3857 ticks at 256 channels) and
  // the compute waves end up waiting longer at the item barrier than the leaner loop saved.
  const bool sk = sizeof(T) == 2 && vec && P.NKG == 2 && P.ntaps % 3 == 0 && (P.mode == 1 || (P.mode == 0 && MT == 2));
#define GO(VV, MD, SKV)                                                                                      \
  do {                                                                                                      \
    auto kfn = tconv_kernel<T, MT, NT, VV, MD, WM, SKV>;                                                    \
    static std::atomic<unsigned long long> optin{0};                                                        \
    if (int ea_ = istgcn_lds_optin((const void*)kfn, optin)) return ea_;                                    \
    int gx = (grid_cap > 0 ? grid_cap : istgcn_resident_blocks((const void*)kfn, NTH, lds)) / gy;           \
    gx = round_up(gx < 1 ? 1 : (gx > P.total_tiles ? P.total_tiles : gx), 8);  /* XCD-affine order: multiple of 8 */ \
    ISTGCN_LAUNCH(kfn, dim3(gx, gy), dim3(NTH), lds, stream, P);                                            \
  } while (0)
#define GOV(MD)                                                                                              \
  do {                                                                                                      \
    if constexpr (sizeof(T) == 2 && MT >= 2 && (MD == 1 || (MD == 0 && MT == 2))) { if (sk) { GO(true, MD, true); break; } } \
    if (vec) GO(true, MD, false); else GO(false, MD, false);                                                \
  } while (0)
  if (P.mode == 1) GOV(1);
  else if (P.mode == 2) GOV(2);
  else GOV(0);
#undef GOV
#undef GO
  ISTGCN_CHECK_LAUNCH();
  return ISTGCN_OK;
}

template <typename T>
int launch_T(TconvParams& P, const TconvGeom& G, int grid_cap, hipStream_t stream) {
  P.F = G.F; P.CC = G.CC; P.nch = G.nch; P.NKG = G.NKG; P.MTtot = G.MTtot; P.min_off = G.min_off; P.Fin = G.Fin;
  P.us_stride = G.us_stride; P.out_stride = G.out_stride; P.off_stat = G.off_stat; P.cin_pad = G.nch * G.CC;
  P.nch_magic = (unsigned)(((1ull << 32) + G.nch - 1) / (unsigned long long)G.nch);
  P.off_u0 = G.off_u0; P.off_u1 = G.off_u1; P.off_o = G.off_o;
  P.tiles_per_seq = ceil_div(P.Mlog, P.F);
  P.total_tiles = P.NM * P.tiles_per_seq;
  P.tps_magic = (unsigned)(((1ull << 32) + P.tiles_per_seq - 1) / (unsigned long long)P.tiles_per_seq);
  const size_t lds = G.lds;
#ifdef ISTGCN_EXPERIMENT
  { const char* e_abl = getenv("ISTGCN_TCONV_ABL"); P.abl = e_abl ? atoi(e_abl) : 0; }
#endif
#ifdef ISTGCN_TCONV_STAMP
  { const char* e_dbg = getenv("ISTGCN_TCONV_DBG_PTR"); P.dbg = e_dbg ? reinterpret_cast<unsigned long long*>(strtoull(e_dbg, nullptr, 0)) : nullptr; }
#endif
  // compute-wave layout: two channel groups x two row groups wherever there are two channel tiles to split
#define CASE(MTv, NTv, WMv) if (G.MT == MTv && G.NT == NTv) return launch4<T, MTv, NTv, WMv>(P, grid_cap, G.gy, lds, stream)
#ifdef ISTGCN_TCONV_ONE          /* ISA inspection builds: one instantiation */
  if constexpr (sizeof(T) == 2 && !std::is_same<T, _Float16>::value) { CASE(2, 2, 2); CASE(4, 2, 2); }
#else
  // (four channel tiles: one per wave, every wave all rows -- each weight fragment is then fetched ONCE per CU and step)
  // (one wave per channel tile x all rows -- each weight fragment fetched once per CU -- measured 5-7 % SLOWER for four
  //  channel tiles: 8 activation fragments per step and wave from LDS instead of 4)
  CASE(1, 1, 1); CASE(1, 2, 1); CASE(2, 1, 2); CASE(4, 1, 2); CASE(2, 2, 2); CASE(4, 2, 2);
#endif
#undef CASE
  return ISTGCN_EINVAL;
}

}  // namespace

// Round-1 kernel (tconv_small.hip), same arguments: serves the shapes without matrix work to overlap.
extern "C" int istgcn_tconv_v1_geometry(int V, int Cin, int Cout, int ntaps, const int* tap_off, int in_mul, int dtype,
                                        int* CC, int* nch, int* MTtot, int* EPL);
extern "C" int istgcn_tconv_v1(const void* in, const void* Wp, const float* bias, const float* pre, int pre_relu,
                               const void* aux, const float* maux, void* out, double* stats, int stats_rep, int mode,
                               int NM, int Tin, int Tout, int Mlog, int V, int Cin, int Cout, int ntaps,
                               const int* tap_off, int in_mul, int out_mul, int out_off, int dtype, int grid_cap,
                               void* stream);
// Which kernel serves a shape (the packed-weight geometry follows the same decision).
static bool tconv_use_v1(int Cin, int Cout) {
  // measured on config 5's bottleneck (fp16, NM=256; us new / round-1 kernel): 64->8 244/201, 128->11 831/552, 256->16
  // 207/140, but 8->64 163/194, 16->256 131/182, 8->8 (15 taps) 160/195, 11->11 235/188: few OUTPUT channels behind many
  // input channels (or unaligned ones) stream better from eight loading waves
  return Cout <= 32 && (Cin > 32 || (Cin & 7) != 0);
}

extern "C" int istgcn_tconv_geometry(int V, int Cin, int Cout, int ntaps, const int* tap_off, int in_mul, int dtype,
                                     int* CC, int* nch, int* MTtot, int* EPL) {
  if (!istgcn_dtype_ok(dtype)) return ISTGCN_EINVAL;
  if (!tap_off || ntaps < 1 || ntaps > MAX_TAPS || V < 1 || V > 128 || Cin < 1 || Cout < 1 || in_mul < 1) return ISTGCN_EINVAL;
  if (tconv_use_v1(Cin, Cout)) return istgcn_tconv_v1_geometry(V, Cin, Cout, ntaps, tap_off, in_mul, dtype, CC, nch, MTtot, EPL);
  tconv_geo::LeanGeom L;
  if (tconv_geo::tconv_lean_geom(V, Cin, Cout, ntaps, tap_off, in_mul, dtype, &L)) {     // the lean kernel's packing (tconv_lean.hip)
    *CC = 32; *nch = L.nch; *MTtot = L.MTtot; *EPL = 8;
    return ISTGCN_OK;
  }
  TconvGeom G;
  int rc = tconv_geom(V, Cin, Cout, ntaps, tap_off, in_mul, dtype, &G);
  if (rc) return rc;
  *CC = G.CC; *nch = G.nch; *MTtot = G.MTtot; *EPL = dtype == 0 ? 4 : 8;
  return ISTGCN_OK;
}

extern "C" int istgcn_tconv(const void* in, const void* Wp, const float* bias, const float* pre, int pre_relu,
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
  // (the staging loads address one sequence through a 32-bit buffer descriptor)
  if ((long long)Tin * V * Cin * (dtype == 0 ? 4 : 2) >= (1ll << 31)) return ISTGCN_EINVAL;
  if (tconv_use_v1(Cin, Cout))
    return istgcn_tconv_v1(in, Wp, bias, pre, pre_relu, aux, maux, out, stats, stats_rep, mode, NM, Tin, Tout, Mlog, V, Cin, Cout,
                           ntaps, tap_off, in_mul, out_mul, out_off, dtype, grid_cap, stream);
  TconvParams P{};
  P.in = in; P.Wp = Wp; P.bias = bias; P.pre = pre; P.aux = aux; P.maux = maux; P.out = out; P.stats = stats;
  P.NM = NM; P.Tin = Tin; P.Tout = Tout; P.Mlog = Mlog; P.V = V; P.Cin = Cin; P.Cout = Cout; P.ntaps = ntaps;
  P.in_mul = in_mul; P.out_mul = out_mul; P.out_off = out_off; P.pre_relu = pre_relu; P.mode = mode;
  P.stats_rep = stats_rep < 1 ? 1 : stats_rep;
  for (int j = 0; j < ntaps; ++j) P.tap_off[j] = tap_off[j];
  {
    tconv_geo::LeanGeom L;
    if (tconv_geo::tconv_lean_geom(V, Cin, Cout, ntaps, tap_off, in_mul, dtype, &L) && (ntaps > 5 || tconv_lean_serves(mode, ntaps, L.UL))) {
      // (9 / 15 taps: the weights are packed for the lean kernel, every mode goes there; 4 / 5 taps: the data gradient)
      BnTail tail{};
      if (stats) istgcn_bn_tail_take(stats, &tail);
      return tconv_lean_launch(in, Wp, bias, pre, pre_relu, aux, maux, out, stats, stats_rep, mode, NM, Tin, Tout, Mlog, V, Cin, Cout,
                               ntaps, tap_off, in_mul, out_mul, out_off, dtype, grid_cap, L, tail, (hipStream_t)stream);
    }
  }
  TconvGeom G;
  int rc = tconv_geom(V, Cin, Cout, ntaps, tap_off, in_mul, dtype, &G);
  if (rc) return rc;
  if (G.NT == 2) {
    // small launches (config 1: 4 sequences): fewer 256-row tiles than half the CUs -> 128-row tiles, twice the workgroups
    // each with half the serial work (the weight layout must not change: same chunk width, same channel tiling)
    int cus = 0, dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
    const long tiles2 = (long)NM * ceil_div(Mlog, G.F) * G.gy;
    if (2 * tiles2 <= cus) {
      TconvGeom G1;
      if (tconv_geom(V, Cin, Cout, ntaps, tap_off, in_mul, dtype, &G1, 1) == ISTGCN_OK && G1.NT == 1 && G1.CC == G.CC &&
          G1.nch == G.nch && G1.MT == G.MT && G1.MTtot == G.MTtot)
        G = G1;
    }
  }
  if (stats) istgcn_bn_tail_take(stats, &P.tail);
  if (dtype == 0) return launch_T<float>(P, G, grid_cap, (hipStream_t)stream);
  if (dtype == 2) return launch_T<_Float16>(P, G, grid_cap, (hipStream_t)stream);
  return launch_T<__bf16>(P, G, grid_cap, (hipStream_t)stream);
}

