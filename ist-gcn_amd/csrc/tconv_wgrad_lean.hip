// Weight gradient of the temporal convolution, trunk layers in 16-bit storage: the lean form (round 4) of round 2's
// wave-specialised kernel `twg_ws` (deleted in round 5; its description is in DESIGN_HISTORY.md)
// (tconv_wgrad.hip) -- the autograd of the (k,1) Conv2d of net/st_gcnold.py:165-175 with BatchNorm + ReLU in front of it:
//
//   dW[j][o][i] += sum_{n,m,v} dz[n, m, v, o] * pre(g[n, m + tap_off[j], v, i])          (stride 1, zero padding)
//
// Same contraction layout as twg_ws: one 8-wave workgroup per CU owns a 64 x 64 (o x i) channel block for ALL taps and
// walks consecutive 128-position tiles; four compute waves hold one (o-tile, i-tile) pair x JT tap accumulators each and
// read both operands transposed (ds_read_b64_tr_b16) from row-major LDS sub-tiles; four memory waves stage the dz tile
// (double-buffered) and the NEW frames of a halo window that slides through an LDS region (a FRESH window is staged whole at
// the front when the region is used up or a sequence starts).  One raw s_barrier per tile.
//
// What changed is the memory role, by the rules tconv_lean.hip measured (profiles/r04_valu_beside_mfma.txt): next to an
// MFMA wave a vector wave gets one plain instruction per 8-13 cycles, so its instruction count IS its time.
//   * twg_ws: ~400 vector + 150 scalar instructions per tile and memory wave, in ~45-instruction basic blocks separated by
//     per-vector predicates (row exists? row inside the sequence? channel vector inside the block?).
//   * here: no per-vector predicate at all.  dz is loaded through a buffer descriptor that ENDS with the tile's last row
//     (rows 125..127 of a 5-frame tile and everything behind a short last tile read as zeros without traffic); the window's
//     frames through a descriptor of the sequence, with unsigned row offsets, so frames in front of / behind the sequence
//     are zeros as well; every thread stores every vector it loaded (the LDS buffers have a slot for each).  Rows that are
//     padding of the conv must be zeros AFTER BatchNorm + ReLU: tiles that touch a sequence edge take a masked copy of the
//     transform (uniform branch, 2 tiles of 60).  The transform itself: shift / and, two fma, one packed conversion, ReLU
//     as v_pk_max_i16 per dword; this file is compiled without the SLP vectoriser (packed fp32 math costs 22 cycles per
//     instruction here).
//   * no conv-bias column sums: the launcher takes this kernel only when the caller does not ask for dbias (the training
//     step does not: a bias in front of a batch-statistics BatchNorm has sum_p dz = 0 identically, functional.py).
//
// Shapes: 16-bit storage, stride 1, C_in % 64 == 0, C_out % 64 == 0, 4..9 taps per launch (10..15 taps -- the 15-tap fold of
// the Inception-TCN -- as two launches over the first 8 and the remaining taps), the first Fin - F frames of a window within
// 224 rows: twg_lean_ok().  Everything else: tconv_wgrad.hip.
//
// hipcc-flags: -fno-slp-vectorize
#include "common.hpp"
#include "gcn_rc.hpp"     // rsrc_t / make_rsrc
#include "tconv_wgrad_lean.hpp"
#include <cstdlib>

extern "C" int istgcn_wgrad_reduce(const float* ws, long long slice, int nsl, float* d0, int n0, float* d1, int n1, void* stream);

namespace {

constexpr int TR = 128;              // positions (rows) of a tile
constexpr int CB = 32;               // channels of an LDS sub-tile
constexpr int RB = CB * 2;           // bytes of a sub-tile row
constexpr int NROLE = 256, NTH = 2 * NROLE;
constexpr int UZ = 4;                // dz vectors per memory thread and tile: 128 rows x 8 vectors / 256
constexpr int US = 4;                // vectors of a window's new frames: F * V <= 128 rows
constexpr int UX = 7;                // vectors of the first Fin - F frames of a fresh window: <= 224 rows
constexpr int SWEEP = NROLE / 8;     // rows one sweep of the 256 threads covers (8 vectors per 64-channel row)
// LDS plan (bytes): the two halves of the double-buffered dz tile (2 sub-tiles each), then the u region (2 sub-tiles of urows rows)
constexpr int OFF_DZ = 0, OFF_DZ1 = 2 * TR * RB, OFF_U = 4 * TR * RB;

struct TwlParams {
  const void* dz;        // [NM][Tz][V][Cout]
  const void* g;         // [NM][Tin][V][Cin]
  const float* pre;      // [2][Cin] scale, shift or null (identity)
  float* dW;             // [ntaps][Cout][Cin] fp32, caller-zeroed
  float* ws;             // partial-sum workspace (one slice per blockIdx.x) or null: flush with atomics
  long long ws_slice;
  int NM, Tin, Tz, V, Cin, Cout, ntaps, pre_relu;
  int tap_off[16];
  int F, Fin, min_off, tiles_per_seq, total_tiles, n_iblk;
  int urows, capf;       // rows of a u sub-tile region; frames the halo window can slide through
  int fs, fbase;         // the input frames the kernel walks are a VIEW of g: view frame f is frame fs * f + fbase of the sequence
                         // (stride-2 convs: one launch per tap parity, each a unit-stride problem on every second frame)
  int tap_dst0, tap_dstep;   // tap j of this launch is tap tap_dst0 + j * tap_dstep of dW
  unsigned long long* dbg;   // experiment builds (-DISTGCN_TWG_STAMP): cycle stamps of workgroup 0
};

#ifdef TWL_X_NOMFMA         /* experiment build: the compute waves issue everything but the MFMAs (results wrong) */
#define TWL_MMA(acc, a, b) { asm volatile("" :: "v"(a), "v"(b)); }
#else
#define TWL_MMA(acc, a, b) mma_kgroup(acc, a, b)
#endif
#ifdef ISTGCN_TWG_STAMP
#define WSTAMP(i) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); tacc[i] += now_ - tlast; tlast = now_; }
#else
#define WSTAMP(i)
#endif

__device__ static inline void lds_barrier() {
  // LDS traffic of this wave retired, then the workgroup barrier; NOT __syncthreads() (its fence drains the prefetches)
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <typename T> __device__ static inline void unpk2(uint32_t p, float& lo, float& hi);
template <> __device__ inline void unpk2<__bf16>(uint32_t p, float& lo, float& hi) {
  lo = __builtin_bit_cast(float, p << 16);
  hi = __builtin_bit_cast(float, p & 0xffff0000u);
}
template <> __device__ inline void unpk2<_Float16>(uint32_t p, float& lo, float& hi) {
  const f16x2 v = __builtin_bit_cast(f16x2, p);
  lo = (float)v[0];
  hi = (float)v[1];
}
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
// max on two packed int16: with floor 0 it is ReLU on two packed 16-bit floats of either format (a negative float is a
// negative int16), with floor -32768 it is the identity
__device__ static inline uint32_t max_pk(uint32_t w, uint32_t floor2) {
  typedef short s16x2 __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, w), __builtin_bit_cast(s16x2, floor2)));
}

struct TPos { int n, mq; };

template <int I, int N, typename F> __device__ static inline void static_for(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}
// scheduling pattern of one step: M MFMAs with R LDS reads spread between them
// (+ E scalar / vector ALU instructions behind each MFMA: address arithmetic that would otherwise sit in one lump)
template <int M, int R, int E = 0> __device__ static inline void interleave_mr() {
  if constexpr (M > 0) {
    constexpr int r = (R + M - 1) / M;
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    if constexpr (r > 0) __builtin_amdgcn_sched_group_barrier(0x100, r, 0);
    if constexpr (E > 0) __builtin_amdgcn_sched_group_barrier(0x006, E, 0);
    interleave_mr<M - 1, R - r, E>();
  }
}

template <typename T, int JT>
__global__ __launch_bounds__(NTH, 2) void twg_lean_kernel(const TwlParams P) {
  using E = Elem<T>;
  typedef typename E::frag frag_t;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool is_compute = wave8 < 4;
  const int ltid = tid & (NROLE - 1);
  const int V = P.V;
  const int oblk = blockIdx.y / P.n_iblk, iblk = blockIdx.y - oblk * P.n_iblk;
  const int o0 = oblk * 64, i0 = iblk * 64;
  const int dz_sub = TR * RB, u_sub = P.urows * RB;        // bytes per 32-channel sub-tile

  {
    // both dz halves and the whole u region start as zeros: everything the contraction can reach is finite from the first
    // tile on (pad positions behind a tile's frames meet zero dz rows and whatever finite u rows lie there)
    const u32x4 z = {0u, 0u, 0u, 0u};
    const int n0 = 2 * 2 * dz_sub / 16, n1 = 2 * u_sub / 16;
    for (int i = tid; i < n0; i += NTH) *reinterpret_cast<u32x4*>(smem + OFF_DZ + i * 16) = z;
    for (int i = tid; i < n1; i += NTH) *reinterpret_cast<u32x4*>(smem + OFF_U + i * 16) = z;
  }
  __syncthreads();

  const int chunk = (P.total_tiles + gridDim.x - 1) / gridDim.x;
  const int t_begin = blockIdx.x * chunk, t_end = min(P.total_tiles, t_begin + chunk);
  const int ntile = t_end > t_begin ? t_end - t_begin : 0;
  const int adv = P.F, keep = P.Fin - adv;                 // frames a window advances by / shares with its predecessor
  // rows at the front of the region a FRESH window's staging writes: its Fin frames, and at least the UX sweeps of the first
  // keep frames (which run on into the slab: the same values to the same places, but rows a live window must not own)
  const int front_rows = max(P.Fin * V, UX * SWEEP);
  // window schedule, computed identically by both roles: tile k is FRESH (window at frame 0 of the region, staged whole) at
  // the start of the walk, at a sequence start, or when sliding on would leave the region; otherwise its window is adv
  // frames further on.  Tiles of a workgroup are consecutive: (sequence, tile in sequence) advance by increments.
  auto tpos_first = [&]() __attribute__((always_inline)) { TPos c; c.n = t_begin / P.tiles_per_seq; c.mq = t_begin - c.n * P.tiles_per_seq; return c; };
  auto tpos_next = [&](TPos c) __attribute__((always_inline)) { if (++c.mq == P.tiles_per_seq) { c.mq = 0; ++c.n; } return c; };
  auto next_window = [&](int k, const TPos& c, int w_prev, bool& fresh) __attribute__((always_inline)) {
    fresh = k == 0 || c.mq == 0 || w_prev + adv + P.Fin > P.capf;
    return fresh ? 0 : w_prev + adv;
  };

  // A compute wave: BOTH o-tiles x one i-tile x half of the taps (wave8 & 1 = i-tile, wave8 >> 1 = tap group): per k-step
  // 2 dz fragments + JW u fragments feed 2 * JW MFMAs (one pair per wave x all taps: 1 + JT fragments for JT MFMAs -- that
  // form is bound by the transposed LDS reads: 640 ds_read_b64_tr_b16 = 2560 LDS cycles per tile against 2304 of MFMA).
  // An odd tap count leaves a middle tap: group 0 contracts it over the first half of a tile's k-steps, group 1 over the
  // second half -- with group 1 walking the k-steps rotated by half a tile, so both groups run the same instruction stream
  // (the shared slot is live in the first NK / 2 steps of the walk); the two partial sums meet in the flush.
  constexpr int JW = (JT + 1) / 2;                          // tap slots of a wave
  constexpr bool SHARED = (JT & 1) != 0;
  constexpr int NK = TR / 16;
  f32x16 acc[2][JW];
#ifdef ISTGCN_TWG_STAMP
  unsigned long long tacc[4] = {0, 0, 0, 0}, tlast = 0;
#endif
  const int it_w = wave8 & 1, grp_w = (wave8 >> 1) & 1;
  if (is_compute) {
    // =========================================== compute waves ===========================================
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
      for (int j = 0; j < JW; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[o][j][r] = 0.f;
    int toff[JW];                                           // byte offset of a slot's tap rows in a u sub-tile
#pragma unroll
    for (int j = 0; j < JW; ++j) {
      const int tap = grp_w ? JT - 1 - j : j;               // slot JW - 1 of an odd JT is the middle tap in both groups
      const int jv = tap < P.ntaps ? tap : 0;               // padding taps alias tap 0 (computed, never flushed)
      toff[j] = (P.tap_off[jv] - P.min_off) * V * RB;
    }
    const int grp = lane >> 4, h = grp >> 1, cblk = (grp & 1) * 16;
    const int q = (lane & 15) >> 2, pp = lane & 3;
    const int coff = (cblk + 4 * pp) * 2;                   // bytes
    const int lrow = 8 * h + q;                             // this lane addresses rows 16*ks + lrow and + 4 of every k-step
    const int rot = (SHARED && grp_w) ? (NK / 2) * 16 * RB : 0;   // bytes the k-step walk of group 1 is rotated by
    lds_barrier();                                          // tile 0 staged (the memory waves' prologue)
#ifdef ISTGCN_TWG_STAMP
    tlast = __builtin_amdgcn_s_memtime();
    const unsigned long long clk0 = tlast, real0 = __builtin_amdgcn_s_memrealtime();
#endif
    int w = 0;
    bool fresh = true;
    TPos cpos = tpos_first();
    // LDS byte offsets of this lane's first rows in the dz half and at every slot's tap of the window, for the first and the
    // second half of the walk (step x reads k-step (x + rot) mod NK: x * 1024 on top of the `lo` bases for x < NK / 2, of the
    // `hi` bases behind).  Offsets, never pointers selected at run time: those decay to flat loads.
    unsigned ap_lo = 0, ap_hi = 0, up_lo[JW], up_hi[JW];
    bool late = false;                                      // the tile set up last is a fresh one whose front window overlaps its predecessor's
    auto setup = [&](int k) __attribute__((always_inline)) {
      const int w_prev = w;
      w = next_window(k, cpos, w, fresh);
      late = k > 0 && k < ntile && fresh && w_prev * V < front_rows;
      cpos = tpos_next(cpos);
      const unsigned ab = (unsigned)(((k & 1) ? OFF_DZ1 : OFF_DZ) + coff + lrow * RB);
      const unsigned ub = (unsigned)(OFF_U + it_w * u_sub + coff + w * V * RB + lrow * RB);
      ap_lo = ab + (unsigned)rot; ap_hi = ab - (unsigned)rot;
#pragma unroll
      for (int j = 0; j < JW; ++j) { up_lo[j] = ub + (unsigned)(toff[j] + rot); up_hi[j] = ub + (unsigned)(toff[j] - rot); }
    };
    frag_t a0[2], a1[2], b0[JW], b1[JW];
    auto nslots = [](int x) constexpr { return (SHARED && x >= NK / 2) ? JW - 1 : JW; };
    auto load_x = [&](auto xc, frag_t (&a)[2], frag_t (&b)[JW]) __attribute__((always_inline)) {
      constexpr int X = decltype(xc)::value;
      const unsigned ab = X < NK / 2 ? ap_lo : ap_hi;
#pragma unroll
      for (int o = 0; o < 2; ++o)
        a[o] = tr_pair<T>(reinterpret_cast<const T*>(smem + ab + o * dz_sub + X * 16 * RB), reinterpret_cast<const T*>(smem + ab + o * dz_sub + X * 16 * RB + 4 * RB));
#pragma unroll
      for (int j = 0; j < nslots(X); ++j) {
        const unsigned ubj = X < NK / 2 ? up_lo[j] : up_hi[j];
        b[j] = tr_pair<T>(reinterpret_cast<const T*>(smem + ubj + X * 16 * RB), reinterpret_cast<const T*>(smem + ubj + X * 16 * RB + 4 * RB));
      }
    };
    auto mma_x = [&](auto xc, const frag_t (&a)[2], const frag_t (&b)[JW]) __attribute__((always_inline)) {
      constexpr int X = decltype(xc)::value;
#pragma unroll
      for (int j = 0; j < nslots(X); ++j)
#pragma unroll
        for (int o = 0; o < 2; ++o) TWL_MMA(acc[o][j], a[o], b[j]);
    };
    using x0 = std::integral_constant<int, 0>;
    using xl = std::integral_constant<int, NK - 1>;
    if (ntile > 0) { setup(0); load_x(x0{}, a0, b0); }
    for (int k = 0; k < ntile; ++k) {
      // (the fragments of this tile's first step are in flight.)  The transposed reads of a step are issued BETWEEN the
      // MFMAs of the step before it, spread evenly (twg_ws: one burst per k-step with the MFMA pipe idle meanwhile).
      static_for<0, NK - 1>([&](auto xc) __attribute__((always_inline)) {
        constexpr int X = decltype(xc)::value;
        using xn = std::integral_constant<int, X + 1>;
        if constexpr (X % 2 == 0) { load_x(xn{}, a1, b1); mma_x(xc, a0, b0); }
        else { load_x(xn{}, a0, b0); mma_x(xc, a1, b1); }
        // the last reads of this tile are in this step: the addresses of the next tile are computed here as well, a few
        // instructions behind each MFMA (behind the barrier they were 250 cycles of scalar work with the MFMA pipe idle;
        // past the last tile the values are unused)
        if constexpr (X == NK - 2) { setup(k + 1); interleave_mr<2 * nslots(X), 2 * (2 + nslots(X + 1)), 8>(); }
        else interleave_mr<2 * nslots(X), 2 * (2 + nslots(X + 1))>();
        __builtin_amdgcn_sched_barrier(0);
      });
      // (Measured and dropped: a commit count of the memory waves in LDS that lets a compute wave fetch the next tile's first
      //  fragments BEFORE its barrier when that tile is already staged -- the steps got slower by what the tail got faster.)
      WSTAMP(0)
#ifdef ISTGCN_TWG_STAMP
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      WSTAMP(2)
      asm volatile("s_barrier" ::: "memory");
      WSTAMP(3)
#else
      lds_barrier();                                        // tile k read (its last MFMAs are still to come), tile k+1 staged (unless it is a late fresh one)
#endif
      if (k + 1 < ntile) {
        if (late) lds_barrier();                            // fresh tile whose front window overlaps window k: staged now
        load_x(x0{}, a0, b0);                               // the next tile's first fragments ...
      }
      __builtin_amdgcn_sched_barrier(0);
      mma_x(xl{}, a1, b1);                                  // ... and, in their shadow, this tile's last step (registers only)
      __builtin_amdgcn_sched_barrier(0);
      WSTAMP(1)
    }
#ifdef ISTGCN_TWG_STAMP
    if (P.dbg && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) { P.dbg[4] = __builtin_amdgcn_s_memtime() - clk0; P.dbg[5] = __builtin_amdgcn_s_memrealtime() - real0; P.dbg[0] = tacc[0]; P.dbg[1] = tacc[1]; P.dbg[2] = tacc[2]; P.dbg[3] = tacc[3]; P.dbg[7] = (unsigned long long)ntile; }
    if (P.dbg && blockIdx.x == 0 && blockIdx.y == 0 && tid == 192) { P.dbg[11] = tacc[0]; P.dbg[12] = tacc[1]; P.dbg[13] = tacc[2]; P.dbg[14] = tacc[3]; }
#endif
  } else {
    // =========================================== memory waves ============================================
    const int q = ltid & 7;                                 // this thread's channel vector of a 64-channel row (both tensors)
    const int sub = q >> 2, ql = q & 3;
    int r0 = ltid >> 3;                                     // row of slot 0; slot u is row r0 + 32 u
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      sc[e] = P.pre ? P.pre[i0 + q * 8 + e] : 1.f;
      sh[e] = P.pre ? P.pre[P.Cin + i0 + q * 8 + e] : 0.f;
    }
    const uint32_t floor2 = P.pre_relu ? 0u : 0x80008000u;
    const T* dzg = reinterpret_cast<const T*>(P.dz);
    const T* gg = reinterpret_cast<const T*>(P.g);
    const unsigned zrow = (unsigned)(P.Cout * 2), urow = (unsigned)(P.Cin * 2);          // bytes per tensor row
    const unsigned zvo0 = (unsigned)r0 * zrow + (unsigned)(q * 16);                     // slot 0 inside the tile / the slab
    const unsigned zstep = SWEEP * zrow;
    // slot u of a slab / of a fresh window's first frames is row r0 + 32 u of it: view frame r / V, joint r % V -- its byte
    // offset from the slab's first row in the sequence, and its frame index (8 bits per slot) for the edge test
    unsigned uoff[UX], frs_lo = 0, frs_hi = 0;
#pragma unroll
    for (int u = 0; u < UX; ++u) {
      const int r = r0 + u * SWEEP, f = r / V;
      uoff[u] = (unsigned)(P.fs * f * V + (r - f * V)) * urow + (unsigned)(q * 16);
      if (u < 4) frs_lo |= (unsigned)f << (8 * u); else frs_hi |= (unsigned)f << (8 * (u - 4));
    }
    auto slot_frame = [&](int u) __attribute__((always_inline)) { return (int)(((u < 4 ? frs_lo : frs_hi) >> (8 * (u & 3))) & 255u); };
    const unsigned seq_z = (unsigned)(P.Tz * V) * zrow, seq_u = (unsigned)(P.Tin * V) * urow;   // bytes per sequence
    const unsigned useq_rec = (unsigned)(P.Tin * V - 1) * urow + 128u;                  // the i-block's last byte in a sequence
    // LDS: vector q of row r of a sub-tile at sub-tile + r * 64 + (q & 3) * 16
    const unsigned zl0 = (unsigned)(sub * dz_sub + r0 * RB + ql * 16);
    const unsigned ul0 = (unsigned)(OFF_U + sub * u_sub + r0 * RB + ql * 16);

    // ---- issue: dz tile + the window's last adv frames of tile k -> registers.  No predicates: the descriptors end where
    //      the data ends (invalid tile: zero records). ----
    auto issue = [&](int k, const TPos& c, u32x4 (&RZ)[UZ], u32x4 (&RS)[US]) __attribute__((always_inline)) {
#ifdef TWL_X_NOMEM          /* experiment build: the memory waves only keep the barriers (results wrong) */
      return;
#endif
      const bool valid = k < ntile;
      const int n = valid ? c.n : 0, m0 = (valid ? c.mq : 0) * P.F;
      const int nf = min(P.F, P.Tz - m0);
      const rsrc_t rz = make_rsrc(reinterpret_cast<const unsigned char*>(dzg) + (size_t)n * seq_z + o0 * 2,
                                  valid ? (unsigned)((m0 + nf) * V - 1) * zrow + 128u : 0u);
      const rsrc_t ru = make_rsrc(reinterpret_cast<const unsigned char*>(gg) + (size_t)n * seq_u + i0 * 2, valid ? useq_rec : 0u);
      unsigned zo = zvo0 + (unsigned)(m0 * V) * zrow;
      const unsigned ub = (unsigned)((P.fs * (m0 + P.min_off + keep) + P.fbase) * V) * urow;   // (frames in front of the sequence: wraps far out of range)
#pragma unroll
      for (int u = 0; u < UZ; ++u) { RZ[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rz, zo, 0, 0)); zo += zstep; }
#pragma unroll
      for (int u = 0; u < US; ++u) RS[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(ru, ub + uoff[u], 0, 0));
    };
    // BatchNorm affine + ReLU of one vector; EDGE: rows outside the sequence (conv padding) are zeros AFTER it
    auto transform = [&](u32x4 v, bool inside, auto edge) __attribute__((always_inline)) {
      uint32_t d4[4] = {v[0], v[1], v[2], v[3]};
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        float lo, hi;
        unpk2<T>(d4[d], lo, hi);
        lo = __builtin_fmaf(lo, sc[2 * d], sh[2 * d]);
        hi = __builtin_fmaf(hi, sc[2 * d + 1], sh[2 * d + 1]);
        uint32_t p = max_pk(pk2<T>(lo, hi), floor2);
        if constexpr (decltype(edge)::value) p = inside ? p : 0u;
        d4[d] = p;
      }
      const u32x4 o = {d4[0], d4[1], d4[2], d4[3]};
      return o;
    };
    using yes = std::integral_constant<bool, true>;
    using no = std::integral_constant<bool, false>;
    // ---- commit: registers of tile k -> LDS (dz half k & 1; the new frames behind the window's kept ones); a FRESH window's
    //      first keep frames are loaded and staged here as well (once per region pass: the pipeline drains) ----
    auto commit = [&](int k, const TPos& c, int w, bool fresh, u32x4 (&RZ)[UZ], u32x4 (&RS)[US]) __attribute__((always_inline)) {
#ifdef TWL_X_NOMEM
      return;
#endif
      asm volatile("" : "+v"(r0));          // (opaque per tile: or every slot's row test / address is kept in its own register)
      unsigned char* dzs = smem + ((k & 1) ? OFF_DZ1 : OFF_DZ) + zl0;
#pragma unroll
      for (int u = 0; u < UZ; ++u) *reinterpret_cast<u32x4*>(dzs + u * (SWEEP * RB)) = RZ[u];
      const int m0 = c.mq * P.F;
      const int fin0 = m0 + P.min_off;                      // first view frame of the window
      const int gf_s = P.fs * (fin0 + keep) + P.fbase;      // frame of the sequence the slab starts at
      unsigned char* us = smem + ul0 + (unsigned)((w + keep) * V * RB);
      if (gf_s >= 0 && gf_s + P.fs * ((TR - 1) / V) < P.Tin) {
#pragma unroll
        for (int u = 0; u < US; ++u) *reinterpret_cast<u32x4*>(us + u * (SWEEP * RB)) = transform(RS[u], true, no{});
      } else {
#pragma unroll
        for (int u = 0; u < US; ++u)
          *reinterpret_cast<u32x4*>(us + u * (SWEEP * RB)) = transform(RS[u], (unsigned)(gf_s + P.fs * slot_frame(u)) < (unsigned)P.Tin, yes{});
      }
      if (fresh) {
        const rsrc_t ru = make_rsrc(reinterpret_cast<const unsigned char*>(gg) + (size_t)c.n * seq_u + i0 * 2, useq_rec);
        const int gf_x = P.fs * fin0 + P.fbase;             // frame of the sequence the window starts at; negative at a sequence start
        const unsigned ub = (unsigned)(gf_x * V) * urow;    // (unsigned: rows in front of the sequence are far out of range)
        u32x4 RX[UX];
#pragma unroll
        for (int u = 0; u < UX; ++u) RX[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(ru, ub + uoff[u], 0, 0));
        unsigned char* ux = smem + ul0 + (unsigned)(w * V * RB);
        // (rows keep*V .. 223 of these slots are the slab's first rows once more: the same values to the same places)
#pragma unroll
        for (int u = 0; u < UX; ++u)
          *reinterpret_cast<u32x4*>(ux + u * (SWEEP * RB)) = transform(RX[u], (unsigned)(gf_x + P.fs * slot_frame(u)) < (unsigned)P.Tin, yes{});
      }
    };
    // tiles in flight: k (being contracted), k+1 (being committed), k+2 (being issued)
    u32x4 ZA[UZ], SA[US], ZB[UZ], SB[US];
    int w1 = 0, w2 = 0;
    bool f1 = true, f2 = true;
    const TPos c0 = tpos_first();
    TPos c1 = tpos_next(c0), c2 = c1;                       // positions of tiles k+1 and k+2
    issue(0, c0, ZA, SA);
    w1 = next_window(1, c1, 0, f1);
    issue(1, c1, ZB, SB);
    __builtin_amdgcn_sched_barrier(0);
    if (ntile > 0) commit(0, c0, 0, true, ZA, SA);
    lds_barrier();                                          // tile 0 staged
#ifdef ISTGCN_TWG_STAMP
    tlast = __builtin_amdgcn_s_memtime();
#endif
    int w_k = 0;                                            // window of the tile the compute waves are on
    auto iteration = [&](int k, u32x4 (&Zn)[UZ], u32x4 (&Sn)[US], u32x4 (&Zf)[UZ], u32x4 (&Sf)[US]) __attribute__((always_inline)) {
      // (Zn, Sn): tile k+1, loaded;  (Zf, Sf): free -> tile k+2
      c2 = tpos_next(c1);
      w2 = next_window(k + 2, c2, w1, f2);
      issue(k + 2, c2, Zf, Sf);                             // (past the last tile: zero-record descriptors, same number of loads)
      __builtin_amdgcn_sched_barrier(0);
      WSTAMP(1)
      const bool have = k + 1 < ntile;
      const bool late = have && f1 && w_k * V < front_rows;   // fresh window at the front would overlap window k: after the barrier
      if (have && !late) commit(k + 1, c1, w1, f1, Zn, Sn);
      WSTAMP(0)
      lds_barrier();                                        // tile k contracted
      if (late) {
        commit(k + 1, c1, w1, f1, Zn, Sn);
        lds_barrier();
      }
      w_k = w1; w1 = w2; f1 = f2; c1 = c2;
      WSTAMP(2)
    };
    for (int k = 0; k < ntile; k += 2) {
      iteration(k, ZB, SB, ZA, SA);
      if (k + 1 < ntile) iteration(k + 1, ZA, SA, ZB, SB);
    }
#ifdef ISTGCN_TWG_STAMP
    if (P.dbg && blockIdx.x == 0 && blockIdx.y == 0 && ltid == 0) { P.dbg[8] = tacc[0]; P.dbg[9] = tacc[1]; P.dbg[10] = tacc[2]; }
#endif
  }
  __syncthreads();

  // ---- flush: D tile rows = o (registers), cols = i (lanes): two 128-byte segments per instruction.  Every element of a
  //      workspace slice is written by exactly one wave of the workgroups with that blockIdx.x (workgroups without tiles
  //      write zeros).  The shared middle tap: group 1 hands its partial sums to group 0 through LDS (the dz halves are free). ----
  if constexpr (SHARED) {
    f32x16* xch = reinterpret_cast<f32x16*>(smem + OFF_DZ);               // [i-tile][o-tile][lane]
    if (is_compute && grp_w == 1) {
#pragma unroll
      for (int o = 0; o < 2; ++o) xch[(it_w * 2 + o) * 64 + lane] = acc[o][JW - 1];
    }
    __syncthreads();
    if (is_compute && grp_w == 0) {
#pragma unroll
      for (int o = 0; o < 2; ++o) {
        const f32x16 v = xch[(it_w * 2 + o) * 64 + lane];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[o][JW - 1][r] += v[r];
      }
    }
  }
  if (is_compute) {
    float* dst = P.ws ? P.ws + (size_t)blockIdx.x * P.ws_slice : P.dW;
    float* p0 = dst + (size_t)(o0 + 4 * (lane >> 5)) * P.Cin + i0 + it_w * CB + (lane & 31);
    const size_t tap_stride = (size_t)P.Cout * P.Cin;
#pragma unroll
    for (int j = 0; j < JW; ++j) {
      const int tap = grp_w ? JT - 1 - j : j;
      if (tap < P.ntaps && !(SHARED && j == JW - 1 && grp_w == 1)) {
#pragma unroll
        for (int o = 0; o < 2; ++o)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            // (a workspace slice holds THIS launch's taps, 0 .. ntaps - 1; dW the layer's: tap_dst0 + j * tap_dstep)
            float* p = p0 + (P.ws ? tap : P.tap_dst0 + tap * P.tap_dstep) * tap_stride + (size_t)(o * CB + (r & 3) + 8 * (r >> 2)) * P.Cin;
            if (P.ws) *p = acc[o][j][r];
            else atomicAdd(p, acc[o][j][r]);
          }
      }
    }
  }
}

// Conv-bias gradient for callers that ask for it (eval-mode backward, the unit tests; the training step does not):
// dbias[o] += sum over all rows of dz[row][o].  One 16-byte vector per thread and row, fp32 partial sums, a block reduction
// through LDS, one float atomic per channel and workgroup.  HBM-bound: dz is read once more.
template <typename T>
__global__ __launch_bounds__(256) void dz_colsum_kernel(const T* __restrict__ dz, float* __restrict__ dbias, long long rows, int Cout) {
  __shared__ float red[256][9];
  const int QC = Cout / 8;                                  // vectors per row (8, 16 or 32: divides 256)
  const int q = threadIdx.x % QC, rl = threadIdx.x / QC, rpb = 256 / QC;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  // (four rows in flight per thread: the loop is latency-bound otherwise)
  const long long step = (long long)gridDim.x * rpb;
  for (long long row = (long long)blockIdx.x * rpb + rl; row < rows; row += 4 * step) {
    u32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long long r = row + u * step;
      v[u] = r < rows ? __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(dz + r * Cout + q * 8)) : u32x4{0u, 0u, 0u, 0u};
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        float lo, hi;
        unpk2<T>(v[u][d], lo, hi);
        acc[2 * d] += lo;
        acc[2 * d + 1] += hi;
      }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[threadIdx.x][e] = acc[e];
  __syncthreads();
  if (threadIdx.x < Cout) {
    const int c = threadIdx.x, cq = c / 8, ce = c % 8;
    float sum = 0.f;
    for (int r = 0; r < rpb; ++r) sum += red[r * QC + cq][ce];
    atomicAdd(dbias + c, sum);
  }
}

template <typename T, int JT>
int launch_twl(TwlParams& P, int grid_cap, hipStream_t stream) {
  P.n_iblk = P.Cin / 64;
  const int n_oblk = P.Cout / 64;
  const size_t base = OFF_U;
  // region: as many slides as fit, at least until an overflow re-stage can run beside the window it replaces (n*F >= Fin)
  const int span_rows = (P.Fin - P.F) * P.V;                // tap span in rows; pad positions of a tile read TR + span rows
  int n_adv = 0;
  for (int n = 8; n >= 1; --n) {
    const size_t rows = (size_t)(n * P.F) * P.V + TR + span_rows;
    if (base + 2 * rows * RB <= 158 * 1024) { n_adv = n; break; }
  }
  if (n_adv * P.F * P.V < (P.Fin * P.V > UX * SWEEP ? P.Fin * P.V : UX * SWEEP)) return -1;   // an overflow re-stage must fit beside the window it replaces
  P.capf = P.Fin + n_adv * P.F;
  P.urows = n_adv * P.F * P.V + TR + span_rows;
  // a fresh window's first Fin - F frames go through UX slots whose last rows spill into the slab: inside the region
  if ((P.Fin - P.F) * P.V > UX * SWEEP || UX * SWEEP > P.urows) return -1;
  const size_t lds = base + (size_t)2 * P.urows * RB;
  const int blocks = n_oblk * P.n_iblk;
  auto kfn = twg_lean_kernel<T, JT>;
  static std::atomic<unsigned long long> optin{0};
  if (int ea_ = istgcn_lds_optin((const void*)kfn, optin)) return ea_;
  if (grid_cap < 1) grid_cap = istgcn_resident_blocks((const void*)kfn, NTH, lds);
  int gx = grid_cap / blocks;
  if (gx < 1) gx = 1;
  if (gx > P.total_tiles) gx = P.total_tiles;
  const int n0 = P.ntaps * P.Cout * P.Cin;
#ifndef TWL_WS_MIN
#define TWL_WS_MIN 128
#endif
  if (P.ws && ((long long)gx * n0 > P.ws_slice || gx < TWL_WS_MIN || P.tap_dstep != 1)) P.ws = nullptr;   // too small / atomics are as fast / taps interleaved in dW
  P.ws_slice = n0;
#ifdef ISTGCN_TWG_STAMP
  { const char* e_dbg = getenv("ISTGCN_TWG_DBG_PTR"); P.dbg = e_dbg ? reinterpret_cast<unsigned long long*>(strtoull(e_dbg, nullptr, 0)) : nullptr; }
#endif
  ISTGCN_LAUNCH(kfn, dim3(gx, blocks), dim3(NTH), lds, stream, P);
  ISTGCN_CHECK_LAUNCH();
  if (P.ws) return istgcn_wgrad_reduce(P.ws, P.ws_slice, gx, P.dW + (size_t)P.tap_dst0 * P.Cout * P.Cin, n0, nullptr, 0, stream);
  return ISTGCN_OK;
}

}  // namespace

static bool twl_subset_ok(int V, int Cin, int Cout, int ntaps, const int* tap_off, int Tin, int Tz) {
  int mn = tap_off[0], mx = tap_off[0];
  for (int j = 1; j < ntaps; ++j) { mn = tap_off[j] < mn ? tap_off[j] : mn; mx = tap_off[j] > mx ? tap_off[j] : mx; }
  const int F = TR / V, Fin = F - 1 + (mx - mn) + 1, keep = Fin - F;
  if (F < 1 || Fin < F || (TR - 1) / V > 255) return false;
  if (F * V > US * SWEEP || keep * V > UX * SWEEP) return false;
  {
    // the LDS plan of launch_twl must come out (so that a multi-launch dispatch never fails half way)
    const int span_rows = keep * V;
    int n_adv = 0;
    for (int n = 8; n >= 1; --n)
      if ((size_t)OFF_U + 2 * ((size_t)(n * F) * V + TR + span_rows) * RB <= 158 * 1024) { n_adv = n; break; }
    if (n_adv * F * V < (Fin * V > UX * SWEEP ? Fin * V : UX * SWEEP) || UX * SWEEP > n_adv * F * V + TR + span_rows) return false;
  }
  // byte offsets inside a sequence are 32-bit, "in front of the sequence" must stay out of range after wrapping
  if ((long long)Tin * V * Cin * 2 >= (1ll << 30) || (long long)Tz * V * Cout * 2 >= (1ll << 30)) return false;
  return true;
}

// stride 2: consecutive tap offsets t0, t0 + 1, ...: the taps of one parity read every second frame -- two unit-stride
// problems on the views "frame 2 f + t0" and "frame 2 f + t0 + 1", with taps 0, 1, 2, ... each
static bool twl_consecutive(int ntaps, const int* tap_off) {
  for (int j = 1; j < ntaps; ++j) if (tap_off[j] != tap_off[0] + j) return false;
  return true;
}

// 10..15 taps (the 15-tap fold of the Inception-TCN, net/st_gcn_multi3_fix_3A_mstcn.py:160-180): two launches over the first
// ceil(n / 2) and the remaining taps, each with the window of its own taps
bool twg_lean_ok(int V, int Cin, int Cout, int ntaps, const int* tap_off, int in_mul, int dtype, int Tin, int Tz) {
  static const bool off = [] { const char* e = getenv("ISTGCN_TWG_LEAN"); return e && atoi(e) == 0; }();   // dispatch override, read once
  if (off || dtype == 0 || in_mul < 1 || in_mul > 2 || ntaps < 1 || ntaps > 15) return false;
  if (Cin % 64 || Cout % 64 || V < 2 || V > TR) return false;
  if (in_mul == 2) {
    const int one[1] = {0};
    if (ntaps == 1) return twl_subset_ok(V, Cin, Cout, 1, one, Tin, Tz);    // the 1 x 1 stride-2 residual conv: every second frame
    if (ntaps < 8 || !twl_consecutive(ntaps, tap_off)) return false;     // (8..15 taps: 4..8 per parity)
    const int view[8] = {0, 1, 2, 3, 4, 5, 6, 7};
    return twl_subset_ok(V, Cin, Cout, (ntaps + 1) / 2, view, Tin, Tz);
  }
  if (ntaps <= 9) return twl_subset_ok(V, Cin, Cout, ntaps, tap_off, Tin, Tz);
  const int n1 = (ntaps + 1) / 2;
  return twl_subset_ok(V, Cin, Cout, n1, tap_off, Tin, Tz) && twl_subset_ok(V, Cin, Cout, ntaps - n1, tap_off + n1, Tin, Tz);
}

// one launch: the taps `tap_off` (view frames) of a view (fs, fbase) of g, written to taps tap_dst0 + j * tap_dstep of dW
static int twl_launch_one(const void* dz, const void* g, const float* pre, int pre_relu, float* dW, int NM, int Tin, int Tz, int V,
                          int Cin, int Cout, int ntaps, const int* tap_off, int fs, int fbase, int tap_dst0, int tap_dstep, int dtype,
                          int grid_cap, float* ws, long long ws_floats, hipStream_t stream) {
  TwlParams P{};
  P.dz = dz; P.g = g; P.pre = pre; P.dW = dW; P.pre_relu = pre ? pre_relu : 0;
  P.NM = NM; P.Tin = Tin; P.Tz = Tz; P.V = V; P.Cin = Cin; P.Cout = Cout; P.ntaps = ntaps;
  P.fs = fs; P.fbase = fbase; P.tap_dst0 = tap_dst0; P.tap_dstep = tap_dstep;
  P.ws = ws_floats > 0 ? ws : nullptr; P.ws_slice = ws_floats;
  int mn = tap_off[0], mx = tap_off[0];
  for (int j = 0; j < 16; ++j) {
    P.tap_off[j] = tap_off[j < ntaps ? j : 0];
    mn = P.tap_off[j] < mn ? P.tap_off[j] : mn; mx = P.tap_off[j] > mx ? P.tap_off[j] : mx;
  }
  P.min_off = mn;
  P.F = TR / V;
  P.Fin = P.F - 1 + (mx - mn) + 1;
  P.tiles_per_seq = ceil_div(Tz, P.F);
  P.total_tiles = NM * P.tiles_per_seq;
#define TWL_CASE(J) return dtype == 2 ? launch_twl<_Float16, J>(P, grid_cap, stream) : launch_twl<__bf16, J>(P, grid_cap, stream)
  if (ntaps <= 4) TWL_CASE(4);
  if (ntaps == 5) TWL_CASE(5);
  if (ntaps <= 7) TWL_CASE(7);
  if (ntaps == 8) TWL_CASE(8);
  TWL_CASE(9);
#undef TWL_CASE
}

int twg_lean_launch(const void* dz, const void* g, const float* pre, int pre_relu, float* dW, int NM, int Tin, int Tz, int V,
                    int Cin, int Cout, int ntaps, const int* tap_off, int in_mul, int dtype, int grid_cap, float* ws, long long ws_floats,
                    hipStream_t stream) {
  if (in_mul == 2 && ntaps == 1) {
    const int one[1] = {0};
    return twl_launch_one(dz, g, pre, pre_relu, dW, NM, Tin, Tz, V, Cin, Cout, 1, one, 2, tap_off[0], 0, 1, dtype, grid_cap, ws, ws_floats, stream);
  }
  if (in_mul == 2) {
    const int view[8] = {0, 1, 2, 3, 4, 5, 6, 7};
    const int ne = (ntaps + 1) / 2, no = ntaps / 2;
    const int rc = twl_launch_one(dz, g, pre, pre_relu, dW, NM, Tin, Tz, V, Cin, Cout, ne, view, 2, tap_off[0], 0, 2, dtype, grid_cap, ws,
                                  ws_floats, stream);
    if (rc != ISTGCN_OK) return rc;
    return twl_launch_one(dz, g, pre, pre_relu, dW, NM, Tin, Tz, V, Cin, Cout, no, view, 2, tap_off[0] + 1, 1, 2, dtype, grid_cap, ws,
                          ws_floats, stream);
  }
  if (ntaps > 9) {
    const int n1 = (ntaps + 1) / 2;
    const int rc = twl_launch_one(dz, g, pre, pre_relu, dW, NM, Tin, Tz, V, Cin, Cout, n1, tap_off, 1, 0, 0, 1, dtype, grid_cap, ws, ws_floats, stream);
    if (rc != ISTGCN_OK) return rc;
    return twl_launch_one(dz, g, pre, pre_relu, dW, NM, Tin, Tz, V, Cin, Cout, ntaps - n1, tap_off + n1, 1, 0, n1, 1, dtype, grid_cap, ws,
                          ws_floats, stream);
  }
  return twl_launch_one(dz, g, pre, pre_relu, dW, NM, Tin, Tz, V, Cin, Cout, ntaps, tap_off, 1, 0, 0, 1, dtype, grid_cap, ws, ws_floats, stream);
}

int twg_lean_dbias(const void* dz, float* dbias, int NM, int Tz, int V, int Cout, int dtype, hipStream_t stream) {
  const long long rows = (long long)NM * Tz * V;
  if (Cout % 8 || 256 % (Cout / 8) || Cout > 256) return ISTGCN_EINVAL;
  const int rpb = 256 / (Cout / 8);
  long long g = (rows + rpb - 1) / rpb;
  if (g > 2048) g = 2048;
  if (dtype == 2) ISTGCN_LAUNCH(dz_colsum_kernel<_Float16>, dim3((int)g), dim3(256), 0, stream, (const _Float16*)dz, dbias, rows, Cout);
  else ISTGCN_LAUNCH(dz_colsum_kernel<__bf16>, dim3((int)g), dim3(256), 0, stream, (const __bf16*)dz, dbias, rows, Cout);
  ISTGCN_CHECK_LAUNCH();
  return ISTGCN_OK;
}
