// Temporal convolution of the trunk layers in 16-bit storage: the wave-specialised kernel of round 4 ("lean").
//
//   out[n, out_mul*m + out_off, v, o] = epi( sum_j sum_i Wf[j][o][i] * pre(in[n, in_mul*m + tap_off[j], v, i]) )
//
// -- the (k,1) Conv2d of net/st_gcnold.py:165-175 with the BatchNorm + ReLU in front of it, the 15-tap fold of the
// Inception-TCN (net/st_gcn_multi3_fix_3A_mstcn.py:160-180,212-215), their stride-2 forms, their data gradients (one
// launch per output phase) and the folded inference conv -- the arithmetic of tconv.hip, reorganised around what the
// instruments of rounds 3 and 4 measured:
//
//  * The two waves of a SIMD share its vector-issue port.  Next to a wave issuing back-to-back MFMAs a vector wave gets
//    one plain instruction per 8-13 cycles (5 alone) and one PACKED fp32 instruction (v_pk_fma_f32, v_pk_add_f32) per 22
//    (tools/valu_beside_mfma.hip, profiles/r04_valu_beside_mfma.txt); neither s_setprio nor which half of the workgroup
//    runs which role changes that in the kernel.  So the memory role is written for few, plain vector instructions:
//      - `pre` (BatchNorm affine + ReLU) per dword: shift / and (unpack two bf16), two fma, ONE v_cvt_pk_bf16_f32, one
//        v_pk_max_i16 (ReLU on the packed pair: a negative float is a negative int16), on scalars; this file is compiled
//        with -fno-slp-vectorize (below) -- the SLP vectoriser re-packed the fma / add pairs and `commit` took 5700 cycles
//        per item with its v_pk_fma_f32, 2400 without (profiles/r04_tconv_lean_stamps.txt);
//      - rows outside the sequence are masked only in tiles that touch a sequence edge (uniform branch); tile decode once
//        per tile and stream; per-lane load offsets fixed per tile (one add per load); no predicated LDS store (the
//        staged buffers have a slot for every vector a thread handles);
//      - the epilogue of a tile is SPREAD over the items of the next tile (EPP image rows per thread and item): constant
//        work per item, the data gradient's `aux` rows requested a whole `commit` ahead of their use, stores through a
//        buffer descriptor that ends with the tile (no row predicate);
//      - mode 1 accumulates sum(d) and sum(d * x); centring and 1/sigma of x-hat are applied to the two sums at the end.
//  * The compute role (one wave per SIMD issues every MFMA) has a fully static step structure: the tap count is a template
//    parameter, a step = (tap, k-group), weight ring DA steps deep straight from L2 (DA divides the steps of an item, so
//    the ring runs continuously across items with static slots), activation fragments two steps ahead from LDS.  The
//    MFMAs of an item's LAST step are issued behind the item barrier, after the first fragment reads of the next item:
//    their LDS latency used to be exposed once per item.
//  * Staged rows are 64 bytes (32 channels) with the 16-byte vectors XOR-swizzled by the JOINT index (bits 2-3), which a
//    tap shift leaves unchanged: fragment reads and the memory waves' 16-byte stores are conflict-free without the
//    80-byte row padding of tconv.hip -- and the buffers of a stride-2 forward (27 input frames) or a 15-tap conv (24)
//    fit next to the output image with 32-channel chunks (tconv.hip falls back to 16-channel chunks there).
//
// Shapes: 16-bit storage, C_in % 32 == 0, C_out % 64 == 0, 4 / 5 / 9 / 15 equally spaced taps, 256-row tiles, 64 or 128
// output channels per workgroup; modes 0 (forward + BatchNorm sums), 1 (data gradient: ReLU mask from `aux`, BatchNorm-
// backward sums), 2 (inference: relu(conv + residual)).  tconv_lean_geom() (tconv_geom.hpp) is the one decision the
// geometry query -- i.e. the weight packer -- and the launcher share.  Everything else: tconv.hip.
//
// hipcc-flags: -fno-slp-vectorize
#include "common.hpp"
#include "gcn_rc.hpp"     // rsrc_t / make_rsrc, unpack2
#include "bn_tail.hpp"
#include "tconv_geom.hpp"
#include <cstdlib>

namespace {

using tconv_geo::NROLE;
using tconv_geo::LeanGeom;
constexpr int NTH = 2 * NROLE;
constexpr int EPP = 4;               // image rows per thread and item in the spread epilogue
constexpr int RB = 64;               // bytes per staged row: 32 channels, no padding
constexpr int OS = 136;              // elements per image row: 128 channels + 8 (rows 4 banks apart)

struct TlParams {
  const void* in;
  const void* Wp;
  const float* bias;     // [Cout] or null
  const float* pre;      // [2][Cin] scale, shift or null
  const void* aux;       // mode 1: [NM][Tout][V][Cout]; mode 2: the same or null
  const float* maux;     // mode 1: [4][Cout] scale, shift, mean, rstd; mode 2: [2][Cout] scale, shift or null
  void* out;
  double* stats;         // [stats_rep][2][Cout] or null
  int NM, Tin, Tout, Mlog, V, Cin, Cout;
  int in_mul, out_mul, out_off, pre_relu, stats_rep;
  int tap0, tapd;        // first tap offset, tap spacing (frames)
  int F, tiles_per_seq, total_tiles, nch, MTtot, min_off, Fin;
  unsigned tps_magic, v_magic;
  int off_stat, off_u0, off_u1, off_o;
  BnTail tail;
  unsigned long long* dbg;   // experiment builds (-DISTGCN_TCONV_STAMP): cycle stamps of workgroup 0 (null otherwise)
};

#ifdef TL_X_NOMFMA          /* experiment build: the compute waves issue everything but the MFMAs (results wrong) */
#define TL_MMA(acc, a, b) { asm volatile("" :: "v"(a), "v"(b)); }
#else
#define TL_MMA(acc, a, b) mma_kgroup(acc, a, b)
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

struct TileL {
  int n, m0, nf, rows, in_rows, r_lo, r_hi;
  unsigned base;          // byte offset of the first staged row inside the sequence (wraps when the halo starts in front of it)
  bool valid, edge;
};

// NTAPS: taps (a step = one (tap, k-group) pair: 2 * NTAPS steps per item).  ULV: 16-byte vectors of a staged chunk per
// memory-wave thread = 64-row sweeps of the staged window (8: up to 512 rows; 11: up to 704).
template <typename T, int MT, int MODE, int NTAPS, int ULV>
__global__ __launch_bounds__(NTH, 2) void tconv_lean_kernel(const TlParams P) {
  using E = Elem<T>;
  constexpr int EPL = 8, CC = 32;
  constexpr int TR = 256;
  constexpr int MTW = MT / 2, NTW = 4;                 // a compute wave: MT/2 channel tiles x 4 row tiles (2 x 2 waves)
  constexpr int NIT = 2 * NTAPS;                       // steps per item
  typedef typename E::frag frag_t;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* stat = reinterpret_cast<float*>(smem + P.off_stat);                 // [2][MT*32]
  float* bias_l = stat + 2 * MT * 32;                                        // [MT*32]
  float* pre_l = bias_l + MT * 32;                                           // [2][Cin]
  T* outs = reinterpret_cast<T*>(smem + P.off_o);                            // [TR][OS]

  const int tid = threadIdx.x, lane = tid & 63;
  const bool is_compute = tid < NROLE;
  const int ltid = tid & (NROLE - 1), wave = ltid >> 6;
  const int V = P.V;
  const int mt0 = blockIdx.y * MT;
  const int cbase_blk = mt0 * 32;

  for (int c = tid; c < 2 * MT * 32; c += NTH) stat[c] = 0.f;
  for (int c = tid; c < MT * 32; c += NTH) bias_l[c] = P.bias ? P.bias[cbase_blk + c] : 0.f;
  if constexpr (MODE == 1) {
    // data gradient: there is no `pre` (checked by the launcher); its two LDS rows hold the ReLU-mask coefficients of this
    // workgroup's channels instead (MT * 32 <= C_in by the geometry) -- in registers they were 16 of the memory role's 256
    for (int c = tid; c < MT * 32; c += NTH) { pre_l[c] = P.maux[cbase_blk + c]; pre_l[P.Cin + c] = P.maux[P.Cout + cbase_blk + c]; }
  } else {
    for (int c = tid; c < 2 * P.Cin; c += NTH) {
      const int h = c / P.Cin;
      pre_l[c] = P.pre ? P.pre[c] : (h == 0 ? 1.f : 0.f);
    }
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
    t.nf = min(P.F, P.Mlog - t.m0);
    t.rows = t.nf * V;
    const int fin0 = P.in_mul * t.m0 + P.min_off;                           // first staged input frame (may be < 0)
    t.in_rows = (P.in_mul * (t.nf - 1) + P.Fin - P.in_mul * (P.F - 1)) * V; // frames actually needed
    t.r_lo = fin0 < 0 ? -fin0 * V : 0;
    t.r_hi = min(t.in_rows, (P.Tin - fin0) * V);
    t.base = (unsigned)(fin0 * V * P.Cin * (int)sizeof(T));
    t.edge = t.r_lo > 0 || t.r_hi < t.in_rows;
    return t;
  };
  // frame / joint index of a row (x / V by multiplication: V <= 128, rows < 2^16)
  auto frame_of = [&](int r) __attribute__((always_inline)) { return (int)__umulhi((unsigned)r, P.v_magic); };
  lds_barrier();
#ifdef ISTGCN_TCONV_STAMP
  unsigned long long tacc[6] = {0, 0, 0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memtime();
#endif

  if (is_compute) {
    // =========================================== compute waves ===========================================
#ifndef TL_X_MEMONLY        /* ISA inspection builds: the memory role alone (register need) */
    constexpr int DA = NIT % 6 == 0 ? 6 : NIT % 5 == 0 ? 5 : 4;     // weight ring: DA - 1 steps ahead; divides NIT
    constexpr int DB = 3, PD = 2;                                    // activation ring: two steps ahead
    // the last step's MFMAs can move behind the barrier when its activation slot is not one the next item's first two
    // steps load into, i.e. when the slot numbering does not rotate from item to item
    constexpr bool DEFER = NIT % DB == 0;
    static_assert(NIT % DA == 0 && NIT >= DA, "static ring slots");
    f32x16 acc[MTW][NTW];
    const int wr = wave / 2, wm = wave % 2;                  // row group, channel group of this wave
    const int h = lane >> 5;
    const unsigned astr = (unsigned)(P.MTtot * 64 * EPL);    // elements between the fragments of consecutive steps
    const unsigned alim = (unsigned)(nch * NIT) * astr;
    const T* abase = Wp + ((size_t)(mt0 + wm * MTW) * 64 + lane) * EPL;
    u32x4 a[DA][MTW], b[DB][NTW];
    unsigned ao = 0;
    auto load_as = [&](u32x4 (&dst)[MTW]) __attribute__((always_inline)) {
#pragma unroll
      for (int m = 0; m < MTW; ++m) dst[m] = *reinterpret_cast<const u32x4*>(abase + ao + (unsigned)(m * 64 * EPL));
      const unsigned an = ao + astr;
      ao = an == alim ? 0u : an;
    };
    // per-lane LDS byte offset of each output row's fragment at tap offset 0, one per k-group: row p of the tile = frame
    // p / V, joint p % V, input frame in_mul * frame; the row's vector (2 kg + h) sits at 16 * ((2 kg + h) ^ sw(joint)),
    // sw = bits 2-3 of the joint -- unchanged by a tap shift (whole frames), so the swizzle is a per-lane constant: kg = 1
    // is kg = 0 with bit 5 flipped.  Rows past a short last tile are NOT clamped: they read rows of the staged buffer that
    // nobody wrote for this tile and produce garbage in their own accumulator columns only (masked by the epilogue).
    unsigned brow[NTW];
#pragma unroll
    for (int tt = 0; tt < NTW; ++tt) {
      const int p = wr * (32 * NTW) + tt * 32 + (lane & 31);
      const int f = frame_of(p), v = p - f * V;
      brow[tt] = (unsigned)(((P.in_mul * f) * V + v) * RB + 16 * (h ^ ((v >> 2) & 3)));
#ifdef TL_X_NOCONF          /* experiment build: fragment reads at conflict-free synthetic addresses (results wrong): what the bank conflicts cost */
      brow[tt] = (unsigned)((tt * 32 + (lane & 31)) * RB + 16 * (h ^ (((lane & 31) >> 2) & 3)));
#endif
    }
    unsigned ubase = (unsigned)P.off_u0;
    // (tapd may be negative -- the data gradient's taps are listed flipped --: unsigned arithmetic, the sums are in range)
    const unsigned roff0 = (unsigned)((P.tap0 - P.min_off) * V * RB), rstep = (unsigned)(P.tapd * V * RB);
    // activation fragments of step s (tap s / 2, k-group s % 2)
    auto load_bs = [&](u32x4 (&dst)[NTW], int s) __attribute__((always_inline)) {
      const unsigned so = ubase + roff0 + (unsigned)(s >> 1) * rstep;
#pragma unroll
      for (int tt = 0; tt < NTW; ++tt)
        dst[tt] = *reinterpret_cast<const u32x4*>(smem + (((s & 1) ? brow[tt] ^ 32u : brow[tt]) + so));
    };
    auto acc_init = [&]() __attribute__((always_inline)) {   // accumulators = conv bias (rows of the D tile = output channels)
#pragma unroll
      for (int m = 0; m < MTW; ++m) {
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias_l + (wm * MTW + m) * 32 + 8 * q4 + 4 * h);
#pragma unroll
          for (int tt = 0; tt < NTW; ++tt)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) acc[m][tt][4 * q4 + jj] = b4[jj];
        }
      }
    };
    auto mmas = [&](u32x4 (&aa)[MTW], u32x4 (&bb)[NTW]) __attribute__((always_inline)) {
#pragma unroll
      for (int m = 0; m < MTW; ++m)
#pragma unroll
        for (int tt = 0; tt < NTW; ++tt) TL_MMA(acc[m][tt], __builtin_bit_cast(frag_t, aa[m]), __builtin_bit_cast(frag_t, bb[tt]));
    };
#pragma unroll
    for (int d = 0; d < DA - 1; ++d) load_as(a[d]);         // in flight while the first chunk is being staged
    lds_barrier();                                          // item 0 staged (the memory waves' prologue)
    int ch = 0;
    if (total_items > 0) {
      acc_init();
#pragma unroll
      for (int d = 0; d < PD; ++d) load_bs(b[d], d);
    }
    TSTAMP(5)
    for (int it = 0; it < total_items; ++it) {
      // One step = this step's MFMAs plus the loads of later steps (weights DA - 1 steps ahead, activations PD ahead, none
      // past the item); sched_group_barrier interleaves one MFMA with the loads in its shadow.
#pragma unroll
      for (int s = 0; s < NIT; ++s) {
        load_as(a[(s + DA - 1) % DA]);
        if (s + PD < NIT) load_bs(b[(s + PD) % DB], s + PD);
        if (!(DEFER && s == NIT - 1)) {
          mmas(a[s % DA], b[s % DB]);
#pragma unroll
          for (int i_ = 0; i_ < MTW * NTW; ++i_) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x006, 5, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      TSTAMP(1)                                             // the steps
      ubase = (unsigned)(((it + 1) & 1) ? P.off_u1 : P.off_u0);
      lds_barrier();                                        // item done: this half of the tile buffer may be refilled
      TSTAMP(2)                                             // wait at the item barrier
      const bool tile_end = ++ch == nch;
      if (tile_end) ch = 0;
      // (the row bases are loop invariants; made opaque per item, or the compiler materialises every `base + tap offset`
      //  sum of the item in its own register and spills them around the loop: reloads with vmcnt(0) waits, a drained ring)
#pragma unroll
      for (int tt = 0; tt < NTW; ++tt) asm volatile("" : "+v"(brow[tt]));
      if (it + 1 < total_items) {                           // the next item's first fragment reads ...
#pragma unroll
        for (int d = 0; d < PD; ++d) load_bs(b[d], d);
      }
      if constexpr (DEFER) {
        __builtin_amdgcn_sched_barrier(0);
        mmas(a[(NIT - 1) % DA], b[(NIT - 1) % DB]);         // ... and, in their shadow, the MFMAs of this item's last step
        __builtin_amdgcn_sched_barrier(0);
      }
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
              const int cl = (wm * MTW + m) * 32 + 8 * q4 + 4 * h;
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
#ifdef ISTGCN_TCONV_STAMP
    if (P.dbg && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) { for (int i = 0; i < 6; ++i) P.dbg[i] = tacc[i]; P.dbg[6] = (unsigned long long)total_items; }
#endif
#endif
  } else {
    // =========================================== memory waves ============================================
    constexpr int Q = CC / EPL;                             // 4 channel vectors per staged row
    const int q = ltid & (Q - 1);
    int r0 = ltid >> 2;                                     // (not const: made opaque once per item, see `iteration`)
    constexpr int RS = NROLE / Q;                           // 64 rows per sweep of the 256 threads
    const int winrows = P.Fin * V;
    const unsigned seq_bytes = (unsigned)(P.Tin * V * P.Cin * (int)sizeof(T));
    const size_t seq_elems = (size_t)P.Tin * V * P.Cin;
    // byte offset of this thread's u-th vector inside a staged window: voff0 + u * vstep; rows past the window are out of
    // range for good (0x80000000 + any tile base stays above the 2^30 the launcher guarantees a sequence to be shorter
    // than: no traffic).  Its LDS slot: row (r0 + 64 u), vector q swizzled by the row's joint -- the buffers hold ULV * 64
    // rows, so every vector has a slot and no store is predicated; the swizzled vector positions of all ULV rows are kept
    // as 2-bit fields of ONE register (a register per slot offset cost 2 x ULV registers and the role spilled).
    unsigned vofft[ULV];
    const unsigned voff0 = (unsigned)((r0 * P.Cin + q * EPL) * (int)sizeof(T)), vstep = (unsigned)(RS * P.Cin * (int)sizeof(T));
    unsigned swz = 0;
#pragma unroll
    for (int u = 0; u < ULV; ++u) {
      const int r = r0 + u * RS;
      const int v = r - frame_of(r) * V;
      swz |= (unsigned)(q ^ ((v >> 2) & 3)) << (2 * u);
    }
    const unsigned lbase = (unsigned)(r0 * RB);

    // ---- issue stream: item (ki, chi) -> registers.  One buffer descriptor per sequence: rows behind the sequence read
    //      as zeros without touching memory; rows in FRONT of it get a far offset once per tile (0xC0000000 + base never
    //      wraps below 2^30). ----
    TileL ti = tile_of(0);
    int ki = 0, chi = 0;
    auto set_issue_tile = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int u = 0; u < ULV; ++u) {
        const int r = r0 + u * RS;
        vofft[u] = r < ti.r_lo ? 0xC0000000u : r < winrows ? voff0 + (unsigned)u * vstep : 0x80000000u;
      }
    };
    set_issue_tile();
    auto issue = [&](u32x4 (&R)[ULV]) __attribute__((always_inline)) {
      const rsrc_t rs = make_rsrc(ing + (size_t)ti.n * seq_elems, ti.valid ? seq_bytes : 0u);
      const unsigned base = ti.base + (unsigned)(chi * CC * (int)sizeof(T));
#pragma unroll
      for (int u = 0; u < ULV; ++u) R[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, vofft[u] + base, 0, 0));
      if (++chi == nch) { chi = 0; ++ki; ti = tile_of(ki); set_issue_tile(); }
    };

    // ---- commit stream: registers of item (kc, chc) -> `pre` -> LDS tile ----
    TileL tc = tile_of(0);
    int kc = 0, chc = 0;
    auto commit = [&](u32x4 (&R)[ULV], unsigned ubytes) __attribute__((always_inline)) {
      if (tc.valid) {
        unsigned char* dst = smem + ubytes;
        asm volatile("" : "+v"(swz));        // (opaque per item: the slot offsets are loop invariants the compiler would keep in 2 x ULV registers)
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
          for (int u = 0; u < ULV; ++u) {
            uint32_t w[4] = {R[u][0], R[u][1], R[u][2], R[u][3]};
            if constexpr (decltype(has_pre)::value) {
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
            *reinterpret_cast<u32x4*>(dst + (lbase + (((swz >> (2 * u)) & 3u) << 4)) + u * (RS * RB)) = o;
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
    const int vq = ltid % VPR;
    int prow = ltid / VPR;                                  // (opaque once per item, like r0)
    unsigned img0 = (unsigned)((prow * OS + vq * EPL) * (int)sizeof(T));
    const unsigned imgs = (unsigned)(RSTEP * OS * (int)sizeof(T));
    const unsigned rowb = (unsigned)(P.Cout * (int)sizeof(T));                          // bytes of an output row
    const unsigned colb = (unsigned)(vq * EPL * (int)sizeof(T));
    float s1[EPL], s2[EPL];
#pragma unroll
    for (int jj = 0; jj < EPL; ++jj) { s1[jj] = 0.f; s2[jj] = 0.f; }
    float msc[EPL], msh[EPL];
    if constexpr (MODE == 2) {
      const int cg = cbase_blk + vq * EPL;
#pragma unroll
      for (int jj = 0; jj < EPL; ++jj) {
        msc[jj] = P.maux ? P.maux[cg + jj] : 1.f;
        msh[jj] = P.maux ? P.maux[P.Cout + cg + jj] : 0.f;
      }
    }
    TileL pend = tile_of(0);
    int ppart = NPARTS;                                     // nothing pending
    T* outg = reinterpret_cast<T*>(P.out);
    const T* auxg = reinterpret_cast<const T*>(P.aux);
    const bool has_aux = MODE == 1 || (MODE == 2 && auxg != nullptr);
    // The tile's output rows in HBM: row p of the tile is output frame out_mul * (m0 + p / V) + out_off, joint p % V, i.e.
    // (p + (out_mul - 1) * (p / V) * V) rows behind the tile's first one; the descriptor ends with the tile's last row, so
    // image rows past a short tile land behind it and their stores are dropped (their loads read zeros).
    auto tile_elem0 = [&](const TileL& t) __attribute__((always_inline)) {
      return ((size_t)(t.n * P.Tout + P.out_mul * t.m0 + P.out_off) * V) * P.Cout + cbase_blk;
    };
    auto tile_bytes = [&](const TileL& t) __attribute__((always_inline)) {
      return (unsigned)((t.rows + (P.out_mul - 1) * (t.nf - 1) * V) * (int)rowb) - (unsigned)(cbase_blk * (int)sizeof(T));
    };
    // (9 / 15 taps are forward convolutions or the data gradients of stride-1 ones: dense output rows, out_mul = 1, checked
    //  by the launcher; 4 / 5 taps are the phases of a stride-2 convolution's data gradient: every out_mul-th frame)
    constexpr bool DENSE = NTAPS >= 9;
    auto row_off = [&](int p) __attribute__((always_inline)) {       // byte offset of image row p in the tile's HBM window
      const int rr = DENSE ? p : p + (P.out_mul - 1) * frame_of(p) * V;
      return (unsigned)rr * rowb + colb;
    };
    // modes 1 / 2: the part's `aux` rows, requested at the top of the iteration (a zero-size descriptor when nothing is
    // pending: the same number of loads on every path keeps the compiler's vmcnt bookkeeping exact)
    auto aux_issue = [&](u32x4 (&AV)[EPP]) __attribute__((always_inline)) {
      if constexpr (MODE >= 1) {
        const bool act = ppart < NPARTS && has_aux;
        const rsrc_t rs = make_rsrc((auxg ? auxg : ing) + (act ? tile_elem0(pend) : 0), act ? tile_bytes(pend) : 0u);
#pragma unroll
        for (int e = 0; e < EPP; ++e)       // (all of the offset in the VGPR: that is what the range check sees)
          AV[e] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, row_off(prow + (ppart * EPP + e) * RSTEP), 0, 2 /* nt: read once */));
      }
    };
    auto epi_part = [&](u32x4 (&AV)[EPP]) __attribute__((always_inline)) {
      if (ppart >= NPARTS) return;
      const rsrc_t ro = make_rsrc(outg + tile_elem0(pend), tile_bytes(pend));
      const int i0 = ppart * EPP;
      const unsigned char* img = smem + P.off_o + img0 + (unsigned)i0 * imgs;
      if constexpr (MODE == 1) {
#pragma unroll
        for (int e4 = 0; e4 < EPL; e4 += 4) {
          const f32x4 s4 = *reinterpret_cast<const f32x4*>(pre_l + vq * EPL + e4);
          const f32x4 h4 = *reinterpret_cast<const f32x4*>(pre_l + P.Cin + vq * EPL + e4);
#pragma unroll
          for (int e = 0; e < 4; ++e) { msc[e4 + e] = s4[e]; msh[e4 + e] = h4[e]; }
        }
      }
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
          } else if constexpr (MODE == 2) {                 // inference: relu(conv + scale * residual + shift)
            const uint32_t g[4] = {AV[e][0], AV[e][1], AV[e][2], AV[e][3]};
#pragma unroll
            for (int d = 0; d < 4; ++d) {
              float glo, ghi, zlo, zhi;
              unpack2<T>(g[d], glo, ghi);
              unpack2<T>(w[d], zlo, zhi);
              const float olo = has_aux ? zlo + __builtin_fmaf(glo, msc[2 * d], msh[2 * d]) : zlo;
              const float ohi = has_aux ? zhi + __builtin_fmaf(ghi, msc[2 * d + 1], msh[2 * d + 1]) : zhi;
              w[d] = relu_pk(pk2<T>(olo, ohi));
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
          __builtin_amdgcn_raw_buffer_store_b128(o, ro, row_off(prow + (i0 + e) * RSTEP), 0, 0);     // rows >= pend.rows: dropped
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
    u32x4 RA[ULV], RBf[ULV];
    issue(RA);
    issue(RBf);
    __builtin_amdgcn_sched_barrier(0);
    commit(RA, (unsigned)P.off_u0);
    lds_barrier();                                          // item 0 staged
    int ch = 0, kt = 0;
    auto iteration = [&](int it, u32x4 (&Rn)[ULV], u32x4 (&Rf)[ULV]) __attribute__((always_inline)) {   // Rn: item it+1, Rf: free -> item it+2
      // (r0 opaque per item: everything derived from it -- the ULV row numbers of the edge masks, slot offsets -- is a loop
      //  invariant the compiler would otherwise keep in a register each and spill: reloads with vmcnt(0) waits in the loop)
      asm volatile("" : "+v"(r0));
      asm volatile("" : "+v"(prow));
      asm volatile("" : "+v"(img0));
      u32x4 AV[EPP];
      aux_issue(AV);
      __builtin_amdgcn_sched_barrier(0);
      issue(Rf);                                            // (past the last item: empty descriptor, same number of loads)
      __builtin_amdgcn_sched_barrier(0);
      TSTAMP(0)
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
        // (NPARTS <= nch, checked by the geometry: the previous image has been streamed out completely)
        pend = tile_of(kt++);
        ppart = 0;
      }
    };
    TSTAMP(5)
    for (int it = 0; it < total_items; it += 2) {
      iteration(it, RBf, RA);
      if (it + 1 < total_items) iteration(it + 1, RA, RBf);
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
    if (MODE != 2 && P.stats) {
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

  if (MODE != 2 && P.stats) {
    lds_barrier();
    double* dst = P.stats + (size_t)(blockIdx.x % P.stats_rep) * 2 * P.Cout;
    for (int c = tid; c < MT * 32; c += NTH) {
      atomic_add_f64(dst + cbase_blk + c, (double)stat[c]);
      atomic_add_f64(dst + P.Cout + cbase_blk + c, (double)stat[MT * 32 + c]);
    }
  }
  bn_tail_run(P.tail, gridDim.x * gridDim.y, reinterpret_cast<unsigned*>(smem));
}

template <typename T, int MT, int MODE, int NTAPS, int ULV>
int launch_lean(const TlParams& P, int grid_cap, int gy, size_t lds, hipStream_t stream) {
  auto kfn = tconv_lean_kernel<T, MT, MODE, NTAPS, ULV>;
  static std::atomic<unsigned long long> optin{0};
  if (int ea = istgcn_lds_optin((const void*)kfn, optin)) return ea;
  int gx = (grid_cap > 0 ? grid_cap : istgcn_resident_blocks((const void*)kfn, NTH, lds)) / gy;
  gx = round_up(gx < 1 ? 1 : (gx > P.total_tiles ? P.total_tiles : gx), 8);      // XCD-affine order: multiple of 8
  ISTGCN_LAUNCH(kfn, dim3(gx, gy), dim3(NTH), lds, stream, P);
  ISTGCN_CHECK_LAUNCH();
  return ISTGCN_OK;
}

template <typename T>
int launch_lean_T(const TlParams& P, const LeanGeom& G, int mode, int ntaps, int grid_cap, hipStream_t stream) {
  const size_t lds = (size_t)G.lds;
#define CASE(MTv, MD, NTv, ULv) \
  if (G.MT == MTv && mode == MD && ntaps == NTv && G.UL == ULv) return launch_lean<T, MTv, MD, NTv, ULv>(P, grid_cap, G.gy, lds, stream)
#ifdef TL_X_ONE             /* ISA inspection builds: one instantiation */
#define CASES(MTv) CASE(MTv, 1, 9, 8)
#else
#define CASES(MTv)                                                                                           \
  CASE(MTv, 0, 9, 8); CASE(MTv, 0, 9, 11); CASE(MTv, 0, 15, 11);                                             \
  CASE(MTv, 1, 9, 8); CASE(MTv, 1, 15, 11); CASE(MTv, 1, 5, 8); CASE(MTv, 1, 4, 8);                          \
  CASE(MTv, 2, 9, 8); CASE(MTv, 2, 9, 11); CASE(MTv, 2, 15, 11)
#endif
  CASES(2);
  CASES(4);
#undef CASES
#undef CASE
  return ISTGCN_EINVAL;
}

}  // namespace

// Which (mode, taps, window) combinations have an instantiation above.
bool tconv_lean_serves(int mode, int ntaps, int ul) {
  if (mode == 1) return (ntaps == 9 && ul == 8) || (ntaps == 15 && ul == 11) || ((ntaps == 5 || ntaps == 4) && ul == 8);
  return (ntaps == 9 && (ul == 8 || ul == 11)) || (ntaps == 15 && ul == 11);
}

int tconv_lean_launch(const void* in, const void* Wp, const float* bias, const float* pre, int pre_relu, const void* aux,
                      const float* maux, void* out, double* stats, int stats_rep, int mode, int NM, int Tin, int Tout,
                      int Mlog, int V, int Cin, int Cout, int ntaps, const int* tap_off, int in_mul, int out_mul, int out_off,
                      int dtype, int grid_cap, const tconv_geo::LeanGeom& G, const BnTail& tail, hipStream_t stream) {
  if (!tconv_lean_serves(mode, ntaps, G.UL)) return ISTGCN_EINVAL;
  if (ntaps >= 9 && out_mul != 1) return ISTGCN_EINVAL;      // (the phases of an 18- / 30-tap stride-2 convolution: nobody's layer)
  if (mode == 1 && pre) return ISTGCN_EINVAL;                // (a data gradient's input is a gradient: no BatchNorm + ReLU in front)
  if ((long long)Tin * V * Cin * 2 >= (1ll << 30)) return ISTGCN_EINVAL;        // (a sequence behind one 32-bit descriptor, far offsets above it)
  TlParams P{};
  P.in = in; P.Wp = Wp; P.bias = bias; P.pre = pre; P.aux = aux; P.maux = maux; P.out = out; P.stats = stats;
  P.NM = NM; P.Tin = Tin; P.Tout = Tout; P.Mlog = Mlog; P.V = V; P.Cin = Cin; P.Cout = Cout;
  P.in_mul = in_mul; P.out_mul = out_mul; P.out_off = out_off; P.pre_relu = pre_relu; P.stats_rep = stats_rep < 1 ? 1 : stats_rep;
  P.tap0 = tap_off[0]; P.tapd = ntaps > 1 ? tap_off[1] - tap_off[0] : 0;
  P.F = G.F; P.nch = G.nch; P.MTtot = G.MTtot; P.min_off = G.min_off; P.Fin = G.Fin;
  P.off_stat = G.off_stat; P.off_u0 = G.off_u0; P.off_u1 = G.off_u1; P.off_o = G.off_o;
  P.tiles_per_seq = ceil_div(Mlog, G.F);
  P.total_tiles = NM * P.tiles_per_seq;
  P.tps_magic = (unsigned)(((1ull << 32) + P.tiles_per_seq - 1) / (unsigned long long)P.tiles_per_seq);
  P.v_magic = V == 1 ? 0u : (unsigned)(((1ull << 32) + V - 1) / (unsigned long long)V);   // x / V == umulhi(x, magic) for x * V < 2^32, V > 1
  P.tail = tail;
#ifdef ISTGCN_TCONV_STAMP
  { const char* e_dbg = getenv("ISTGCN_TCONV_DBG_PTR"); P.dbg = e_dbg ? reinterpret_cast<unsigned long long*>(strtoull(e_dbg, nullptr, 0)) : nullptr; }
#endif
  if (dtype == 2) return launch_lean_T<_Float16>(P, G, mode, ntaps, grid_cap, stream);
  return launch_lean_T<__bf16>(P, G, mode, ntaps, grid_cap, stream);
}
