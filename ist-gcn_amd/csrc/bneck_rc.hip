// Bottleneck temporal unit of the "1x1" models, 16-bit storage, register-chained (round 3).
//
//   u  = relu(bn1(g))                                   tcn_start        net/st_gcn_mstcn_1x1.py:174-177, 258
//   q  = Ws u + bs            C  -> w = int(sqrt(C))    conv_1x1_start   :178-184, 259
//   yb = sum_j Wt_j q(t+j-7) + bt   w -> w, 15 taps     tcn_1/2/3 x mstcn_importance, pre-summed by the host (:185-205, 260-263)
//   z  = We yb + be           w  -> C                   conv_1x1_end     :206-212, 264      (+ BatchNorm sums of tcn_end, :213)
//
// The chain is linear between the two BatchNorms, the narrow tensors are 1/4 ... 1/16 of the wide ones, and every pass
// is a pure stream (2 w FLOP per byte): the round-1/2 temporal-conv kernels ran its six launches per block at 1.8-2.9
// TB/s.  Two kernels replace them, both ways:
//
//   bneck_in   wide -> narrow, a FLAT stream over all positions (no frame structure): the 32 x 16-byte row vectors of a
//              tile are loaded STRAIGHT from HBM into the B operand of  D[n][p] = W[n][:] . x[p][:]  (channels are the k
//              axis and contiguous in memory), the BatchNorm affine + ReLU is applied to those registers, the accumulator
//              tile has the position on the lane and 4 consecutive narrow channels per register quad: 8-byte stores that
//              tile the narrow rows exactly.  No LDS in the loop.  Forward: q = Ws relu(bn1(g)) + bs; backward: dyb = We^T dz.
//   bneck_out  narrow -> (taps) -> narrow -> wide: a wave walks the frames of a sequence segment with the last 15 narrow
//              frames in REGISTERS (a narrow frame row is 16 / 32 bytes = the B operand of one k-step as it lies in memory):
//                D2[n'][p]  = sum_j Wt_j[n'][:] . ring_j[:][p]         15 MFMAs, A = tap fragments from LDS
//                D3[p][o]   = D2^T[p][:] . We^T[:][o]                  the accumulator tile of the first product, converted
//                                                                      pairwise in registers, IS the A operand of the second
//                                                                      (its k order adopted by the packed We fragments)
//              D3 has the output channel on the lane: BatchNorm sums per lane, then the per-wave LDS image of gcn_rc.hip
//              turns it into 16-byte row vectors stored as whole 128-byte lines (mode 1: masked by the ReLU of the
//              producer's BatchNorm recomputed from `aux`, with the two BatchNorm-backward sums).  The narrow intermediate
//              (yb forward, dq backward) is stored on the way: the weight gradients need it.
//              Forward: yb, z from q;  backward: dq, d1 from dyb (taps transposed, one launch per stride phase).
// Frames / positions outside a sequence are outside a buffer descriptor: loads return zeros (= the Conv2d zero padding),
// stores are dropped; no predicated memory operation anywhere (gcn_rc.hpp).
#include "gcn_rc.hpp"
#include "dropout.hpp"
#include "bn_tail.hpp"
#include <type_traits>

namespace {

constexpr unsigned OOB = 0x7ffffff0u;
// cache policy of the loads that read a WIDE tensor exactly once (raw buffer `aux` operand: 2 = nt, streaming): they do not
// displace what the neighbouring kernels re-read -- measured WORSE here (config 5: 46.3 -> 49.1 ms/step, config 3 bf16 28.6 -> 29.9):
// these tensors were written by the previous kernel and still sit in the Infinity Cache, which a streaming read forgoes.  Default
// policy (0) therefore; ISTGCN_X_BNECK_LDAUX overrides (experiments).
#ifndef ISTGCN_X_BNECK_LDAUX
#define ISTGCN_X_BNECK_LDAUX 0
#endif
constexpr int LDW = ISTGCN_X_BNECK_LDAUX;

// ======================================================================================================================
// bneck_in
// ======================================================================================================================
struct BinParams {
  const void* x; void* y; const float* W; const float* bias; const float* pre;
  long long w_rs, w_cs;          // W(n, c) = W[n * w_rs + c * w_cs]
  long long rows;                // positions
  int C, Wn, Wp, pre_relu;
  int ntiles;
};

template <typename T, int S>
__global__ __launch_bounds__(256, S >= 16 ? 2 : 4) void bneck_in_kernel(const BinParams P) {
  using E = Elem<T>;
  typedef typename E::frag frag_t;
  constexpr int C = 16 * S;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u32x4* wl = reinterpret_cast<u32x4*>(smem);                          // [S][64] A fragments
  float* pre_l = reinterpret_cast<float*>(smem + (size_t)S * 64 * 16); // [2][C]
  float* wsc = pre_l + 2 * C;                                          // [16][C] staging copy of W

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 31, h = lane >> 5;

  for (int i = tid; i < 16 * C; i += 256) {
    const int n = i / C, cc = i - n * C;
    wsc[i] = n < P.Wn ? P.W[n * P.w_rs + cc * P.w_cs] : 0.f;
  }
  for (int i = tid; i < 2 * C; i += 256) pre_l[i] = P.pre ? P.pre[i] : (i < C ? 1.f : 0.f);
  __syncthreads();
  for (int s = wave; s < S; s += 4) {
    frag_t f;
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = E::from_f(c < 16 ? wsc[c * C + 16 * s + 8 * h + j] : 0.f);
    wl[s * 64 + lane] = __builtin_bit_cast(u32x4, f);
  }
  // bias of the accumulator tile: register i of lane half h is narrow channel (i & 3) + 8 (i >> 2) + 4 h
  float bv[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int n = (i & 3) + 8 * (i >> 2) + 4 * h;
    bv[i] = (P.bias && n < P.Wn) ? P.bias[n] : 0.f;
  }
  __syncthreads();

  const rsrc_t rx = make_rsrc(P.x, (unsigned)(P.rows * C * 2));
  const rsrc_t ry = make_rsrc(P.y, (unsigned)(P.rows * P.Wp * 2));
  const unsigned xl = (unsigned)(c * C + 8 * h) * 2u;                  // this lane's vector of k-step 0 in a tile
  const unsigned yl = (unsigned)(c * P.Wp + 4 * h) * 2u;
  const bool pre = P.pre != nullptr, relu = P.pre_relu != 0, wide16 = P.Wp == 16;
  const int nw = gridDim.x * 4;

  auto loadx = [&](int tile, u32x4 (&xf)[S]) __attribute__((always_inline)) {
    const unsigned base = (unsigned)tile * (unsigned)(32 * C * 2) + xl;   // (beyond the last row: outside the descriptor)
#pragma unroll
    for (int s = 0; s < S; ++s) xf[s] = __builtin_amdgcn_raw_buffer_load_b128(rx, base + 32u * s, 0, LDW);
  };
  auto work = [&](int tile, u32x4 (&xf)[S]) __attribute__((always_inline)) {
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = i < 8 ? bv[i] : 0.f;
#pragma unroll
    for (int s = 0; s < S; ++s) {
      u32x4 v = xf[s];
      if (pre) {
        const f32x4 sc0 = *reinterpret_cast<const f32x4*>(pre_l + 16 * s + 8 * h), sc1 = *reinterpret_cast<const f32x4*>(pre_l + 16 * s + 8 * h + 4);
        const f32x4 sh0 = *reinterpret_cast<const f32x4*>(pre_l + C + 16 * s + 8 * h), sh1 = *reinterpret_cast<const f32x4*>(pre_l + C + 16 * s + 8 * h + 4);
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          float lo, hi;
          unpack2<T>(v[d], lo, hi);
          const float s0 = d < 2 ? sc0[2 * d] : sc1[2 * d - 4], s1 = d < 2 ? sc0[2 * d + 1] : sc1[2 * d - 3];
          const float t0 = d < 2 ? sh0[2 * d] : sh1[2 * d - 4], t1 = d < 2 ? sh0[2 * d + 1] : sh1[2 * d - 3];
          lo = fmaf(lo, s0, t0);
          hi = fmaf(hi, s1, t1);
          if (relu) { lo = fmaxf(lo, 0.f); hi = fmaxf(hi, 0.f); }
          v[d] = pack2<T>(lo, hi);
        }
      }
      mma_kgroup(acc, __builtin_bit_cast(frag_t, wl[s * 64 + lane]), __builtin_bit_cast(frag_t, v));
    }
    const unsigned yb = (unsigned)tile * (unsigned)(32 * P.Wp * 2) + yl;
    __builtin_amdgcn_raw_buffer_store_b64(u32x2{pack2<T>(acc[0], acc[1]), pack2<T>(acc[2], acc[3])}, ry, yb, 0, 0);
    // (8-wide rows: the second quad lies past the descriptor for nobody -- the offset itself is sent out of range)
    __builtin_amdgcn_raw_buffer_store_b64(u32x2{pack2<T>(acc[4], acc[5]), pack2<T>(acc[6], acc[7])}, ry, wide16 ? yb + 16u : OOB, 0, 0);
  };

  int tile = blockIdx.x * 4 + wave;
  if (tile < P.ntiles) {
    u32x4 xa[S], xb[S];
    loadx(tile, xa);
    for (;;) {
      const int t2 = tile + nw;
      const bool more = t2 < P.ntiles;
      loadx(more ? t2 : tile, xb);
      __builtin_amdgcn_sched_barrier(0);
      work(tile, xa);
      if (!more) break;
      const int t3 = t2 + nw;
      const bool more2 = t3 < P.ntiles;
      loadx(more2 ? t3 : t2, xa);
      __builtin_amdgcn_sched_barrier(0);
      work(t2, xb);
      if (!more2) break;
      tile = t3;
    }
  }
}

template <typename T, int S>
int bin_launch(BinParams P, int grid_cap, hipStream_t stream) {
  auto kfn = bneck_in_kernel<T, S>;
  constexpr int C = 16 * S;
  const size_t lds = (size_t)S * 64 * 16 + (size_t)2 * C * 4 + (size_t)16 * C * 4;
  static std::atomic<unsigned long long> optin{0};
  if (int ea = istgcn_lds_optin((const void*)kfn, optin)) return ea;
  int g = grid_cap > 0 ? grid_cap : istgcn_resident_blocks((const void*)kfn, 256, lds);
  if (g > (P.ntiles + 3) / 4) g = (P.ntiles + 3) / 4;
  if (g < 1) g = 1;
  ISTGCN_LAUNCH(kfn, dim3(g), dim3(256), lds, stream, P);
  ISTGCN_CHECK_LAUNCH();
  return ISTGCN_OK;
}

// ======================================================================================================================
// bneck_out
// ======================================================================================================================
constexpr int NTAP = 15;        // ring length: tap slots of a launch (fewer taps: zero fragments in the tail slots)

struct BoutParams {
  const void* q; void* yb; void* z; const void* aux; const float* maux;
  const float* Wt; long long wt_ts, wt_rs, wt_cs;      // Wt(j, n', n) = Wt[tap_sel[j] * wt_ts + n' * wt_rs + n * wt_cs]
  const float* bt;
  const float* We; long long we_rs, we_cs;             // We(o, n') = We[o * we_rs + n' * we_cs]
  const float* be;
  double* stats;
  int stats_rep;
  int NM, Tin, Tout, Mlog, V, C, Wn, Wp, ntaps, off0, out_mul, out_off;
  int tap_sel[NTAP];
  int seg, nseg_seq, nseg;       // logical frames per segment, segments per sequence, segments in total
  BnTail tail;                   // "last workgroup finalises" the BatchNorm behind the sums (bn_tail.hpp), when armed
};

// NPAIR = C / 64, IM = frames the input advances per output frame (1, or 2: the stride-2 forward conv), MODE 0: BatchNorm
// sums of the wide output; MODE 1: data gradient through the producer's BatchNorm + ReLU (mask from `aux`, backward sums).
template <typename T, int NPAIR, int IM, int MODE>
__global__ __launch_bounds__(RC_NTH, 2) void bneck_out_kernel(const BoutParams P) {
  using E = Elem<T>;
  typedef typename E::frag frag_t;
  constexpr int C = 64 * NPAIR;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u32x4* wtl = reinterpret_cast<u32x4*>(smem);                                  // [NTAP][64] A fragments of the taps
  u32x4* wel = wtl + NTAP * 64;                                                 // [2 NPAIR][64] B fragments of the expansion
  uint32_t* img_all = reinterpret_cast<uint32_t*>(wel + 2 * NPAIR * 64);        // 8 per-wave images (setup: scratch)
  float* stat = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(img_all) + 8 * IMG_BYTES);   // [2][C]
  float* mx = stat + 2 * C;                                                     // MODE 1: [4][C] scale, shift, mean, rstd
  float* bel = mx + 4 * C;                                                      // [C] bias of the expansion
  u32x4* idl = reinterpret_cast<u32x4*>(bel + C);                               // [2][64] identity fragments (MODE 1)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 31, h = lane >> 5;
  const int V = P.V, Wn = P.Wn, Wp = P.Wp;

  // ---- setup: the (tiny) weights through an LDS scratch copy, then the per-lane fragments ----
  {
    float* wsc = reinterpret_cast<float*>(img_all);                             // [NTAP][16][16] taps, then [C][16] expansion
    for (int i = tid; i < NTAP * 256; i += RC_NTH) {
      const int j = i >> 8, r = (i >> 4) & 15, cc = i & 15;
      wsc[i] = (j < P.ntaps && r < Wn && cc < Wn) ? P.Wt[P.tap_sel[j] * P.wt_ts + r * P.wt_rs + cc * P.wt_cs] : 0.f;
    }
    float* esc = wsc + NTAP * 256;
    for (int i = tid; i < C * 16; i += RC_NTH) {
      const int o = i >> 4, n = i & 15;
      esc[i] = n < Wn ? P.We[o * P.we_rs + n * P.we_cs] : 0.f;
    }
    for (int i = tid; i < 2 * C; i += RC_NTH) stat[i] = 0.f;
    if (MODE == 1)
      for (int i = tid; i < 4 * C; i += RC_NTH) mx[i] = P.maux[i];
    __syncthreads();
    // taps: A operand, lane (n' = c, h), element e = Wt_j[n'][8h + e]
    for (int j = wave8; j < NTAP; j += 8) {
      frag_t f;
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] = E::from_f(c < 16 ? wsc[j * 256 + c * 16 + 8 * h + e] : 0.f);
      wtl[j * 64 + lane] = __builtin_bit_cast(u32x4, f);
    }
    // expansion: B operand of D3[p][o], lane (o = 32 t + c, h), element e = We[o][n'] with n' in the CHAINED k order of
    // the converted accumulator tile: n' = 8 (e >> 2) + 4 h + (e & 3)
    for (int t = wave8; t < 2 * NPAIR; t += 8) {
      frag_t f;
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] = E::from_f(esc[(32 * t + c) * 16 + 8 * (e >> 2) + 4 * h + (e & 3)]);
      wel[t * 64 + lane] = __builtin_bit_cast(u32x4, f);
    }
  }
  float btv[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int n = (i & 3) + 8 * (i >> 2) + 4 * h;
    btv[i] = (P.bt && n < Wn) ? P.bt[n] : 0.f;
  }
  for (int i = tid; i < C; i += RC_NTH) bel[i] = P.be ? P.be[i] : 0.f;
  if (wave8 < 2) {
    frag_t f;
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = E::from_f(16 * wave8 + 8 * h + e == c ? 1.f : 0.f);
    idl[wave8 * 64 + lane] = __builtin_bit_cast(u32x4, f);
  }
  __syncthreads();                                                              // the scratch becomes the waves' images

  const T* qg = reinterpret_cast<const T*>(P.q);
  T* ybg = reinterpret_cast<T*>(P.yb);
  T* zg = reinterpret_cast<T*>(P.z);
  const T* auxg = reinterpret_cast<const T*>(P.aux);
  uint32_t* img = img_all + wave8 * (IMG_BYTES / 4);
  const uint32_t* imgr = img + (lane >> 3) * IMG_RS + 8 * (lane & 7);     // this lane's slot of a copied-out pair-row
  const unsigned qfrm_b = (unsigned)(V * Wp) * 2u, zfrm_b = (unsigned)(V * C) * 2u;
  // narrow row vector of this lane in a frame: position c, channels 8h .. 8h+7 (8-wide rows: half 1 reads zeros)
  const unsigned ql = (c < V && 8 * h < Wp) ? (unsigned)(c * Wp + 8 * h) * 2u : OOB;
  const unsigned ybl = c < V ? (unsigned)(c * Wp + 4 * h) * 2u : OOB;
  const unsigned ybl2 = (c < V && Wp == 16) ? ybl + 16u : OOB;
  const int rp = lane >> 3, chunk = lane & 7;
  const unsigned zrow_b = (unsigned)C * 2u;
  // rows of a D3 register: p = (i & 3) + 8 (i >> 2) + 4 h; rows >= V are pad rows (their A-operand rows are zeros, but the
  // biases are not): forced to zero so that the sums see exactly the stored frame.  A GHOST frame (the walk is padded to
  // whole rounds of three steps so that the loop body is one fixed sequence of memory operations) has no rows at all.

  float s1[2 * NPAIR], s2[2 * NPAIR];                       // per-lane sums of its channel in each 32-channel tile
#pragma unroll
  for (int t = 0; t < 2 * NPAIR; ++t) { s1[t] = 0.f; s2[t] = 0.f; }
  // MODE 1 needs `aux` (the producer's BatchNorm input at the output positions) in the layout of D3 -- channel on the
  // lane, rows in the registers -- but memory has it row-major.  Its 16-byte row vectors ARE the A operand of a product
  // over channels, so two MFMAs against identity fragments deliver the tile transposed, exactly (x * 1.0 in fp32): the
  // mask and both sums then run per lane like the forward sums, and the masked tile takes the same image path.
  // B operand of that product: lane (o = c, h), k-step s, element e = [16 s + 8 h + e == c]
  // (kept in LDS next to the other fragments: `idl`)
  const unsigned auxl = c < V ? (unsigned)(c * C + 8 * h) * 2u : OOB;     // + (32 t + 16 s) * 2

  // ---- one output frame: ring -> D2 (+ store) -> D3 per channel tile -> epilogue per 64-channel pair ----
  auto frame = [&](int n, int m, bool live, const u32x4 (&R)[NTAP]) __attribute__((always_inline)) {
    const int tf = live ? P.out_mul * m + P.out_off : 0;
    const size_t fo = (size_t)n * P.Tout + tf;
    const int vlim = live ? V : 0;                           // rows that exist
    const unsigned nb_q = live ? qfrm_b : 0u, nb_z = live ? zfrm_b : 0u;   // a ghost frame's descriptors are empty
    // (the fragment reads below are loop-invariant LDS loads: behind an opaque lane index they stay inside the loop
    //  instead of being hoisted into ~100 registers)
    int ln = lane;
    asm volatile("" : "+v"(ln));
    // (likewise the per-lane store / aux offsets: hoisted, the 4 + 4 offsets per channel pair live in registers forever)
    unsigned zl = (unsigned)(2 * rp) * zrow_b + (unsigned)(8 * chunk) * 2u, al = auxl;
    asm volatile("" : "+v"(zl), "+v"(al));
    const rsrc_t rz = make_rsrc(zg + fo * V * C, nb_z);
    const rsrc_t ra = make_rsrc(MODE == 1 ? auxg + fo * V * C : zg + fo * V * C, nb_z);
    u32x4 ax[2], axn[2];                                     // aux row vectors of the current / next channel tile
    if constexpr (MODE == 1) {
      ax[0] = __builtin_amdgcn_raw_buffer_load_b128(ra, al, 0, LDW);
      ax[1] = __builtin_amdgcn_raw_buffer_load_b128(ra, al + 32u, 0, LDW);
    }
    f32x16 D2;
#pragma unroll
    for (int i = 0; i < 16; ++i) D2[i] = i < 8 ? btv[i] : 0.f;
    {
      // tap fragments in groups of three, one group ahead (left to itself the scheduler issues all 15 reads first:
      // 60 registers next to the ring's 60)
      constexpr int TG = 3;
      u32x4 wa[TG], wb[TG];
#pragma unroll
      for (int i = 0; i < TG; ++i) wa[i] = wtl[i * 64 + ln];
#pragma unroll
      for (int g = 0; g < NTAP / TG; ++g) {
        __builtin_amdgcn_sched_barrier(0);
        if (g + 1 < NTAP / TG) {
#pragma unroll
          for (int i = 0; i < TG; ++i) wb[i] = wtl[((g + 1) * TG + i) * 64 + ln];
        }
#pragma unroll
        for (int i = 0; i < TG; ++i) mma_kgroup(D2, __builtin_bit_cast(frag_t, wa[i]), __builtin_bit_cast(frag_t, R[g * TG + i]));
#pragma unroll
        for (int i = 0; i < TG; ++i) wa[i] = wb[i];
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    u32x4 a2;                                                // D2 rounded: the stored narrow frame AND the next A operand
#pragma unroll
    for (int d = 0; d < 4; ++d) a2[d] = pack2<T>(D2[2 * d], D2[2 * d + 1]);
    {
      const rsrc_t ryb = make_rsrc(ybg + fo * V * Wp, nb_q);
      __builtin_amdgcn_raw_buffer_store_b64(u32x2{a2[0], a2[1]}, ryb, ybl, 0, 0);
      __builtin_amdgcn_raw_buffer_store_b64(u32x2{a2[2], a2[3]}, ryb, ybl2, 0, 0);
    }
#pragma unroll
    for (int pr2 = 0; pr2 < NPAIR; ++pr2) {
#pragma unroll
      for (int jt = 0; jt < 2; ++jt) {
        const int t = 2 * pr2 + jt;
        f32x16 D3, Xa;
        __builtin_amdgcn_sched_barrier(0);                   // one tile at a time: overlapping tiles is what spills
        if constexpr (MODE == 1) {
          if (t + 1 < 2 * NPAIR) {
            axn[0] = __builtin_amdgcn_raw_buffer_load_b128(ra, al + (unsigned)(32 * (t + 1)) * 2u, 0, LDW);
            axn[1] = __builtin_amdgcn_raw_buffer_load_b128(ra, al + (unsigned)(32 * (t + 1) + 16) * 2u, 0, LDW);
          }
#pragma unroll
          for (int i = 0; i < 16; ++i) Xa[i] = 0.f;
          mma_kgroup(Xa, __builtin_bit_cast(frag_t, ax[0]), __builtin_bit_cast(frag_t, idl[ln]));
          mma_kgroup(Xa, __builtin_bit_cast(frag_t, ax[1]), __builtin_bit_cast(frag_t, idl[64 + ln]));
        }
        const float bet = bel[32 * t + (ln & 31)];
#pragma unroll
        for (int i = 0; i < 16; ++i) D3[i] = bet;
        mma_kgroup(D3, __builtin_bit_cast(frag_t, a2), __builtin_bit_cast(frag_t, wel[t * 64 + ln]));
        float msc = 0.f, msh = 0.f, mmu = 0.f, mrs = 0.f;
        if constexpr (MODE == 1) {
          const int ch = 32 * t + (ln & 31);
          msc = mx[ch]; msh = mx[C + ch]; mmu = mx[2 * C + ch]; mrs = mx[3 * C + ch];
        }
#pragma unroll
        for (int qd = 0; qd < 8; ++qd) {
          const int r0 = ((2 * qd) & 3) + 8 * ((2 * qd) >> 2) + 4 * h;          // rows of registers 2 qd and 2 qd + 1
          float v0 = r0 < vlim ? D3[2 * qd] : 0.f, v1 = r0 + 1 < vlim ? D3[2 * qd + 1] : 0.f;
          if constexpr (MODE == 1) {
            v0 = fmaf(Xa[2 * qd], msc, msh) > 0.f ? v0 : 0.f;
            v1 = fmaf(Xa[2 * qd + 1], msc, msh) > 0.f ? v1 : 0.f;
          }
          const uint32_t pk = pack2<T>(v0, v1);
          float lo, hi;
          unpack2<T>(pk, lo, hi);
          s1[t] += lo + hi;
          if constexpr (MODE == 1) {
            s2[t] = fmaf(lo, (Xa[2 * qd] - mmu) * mrs, s2[t]);
            s2[t] = fmaf(hi, (Xa[2 * qd + 1] - mmu) * mrs, s2[t]);
          } else {
            s2[t] = fmaf(lo, lo, s2[t]);
            s2[t] = fmaf(hi, hi, s2[t]);
          }
          const int p = (qd & 1) + 4 * (qd >> 1) + 2 * h;
          img[p * IMG_RS + 32 * jt + c] = pk;
        }
        // the sums are used at the end of the kernel only: left alone, their updates sink behind the whole frame and the
        // eight packed registers of EVERY tile stay alive until then (64 registers at 256 channels)
        asm volatile("" : "+v"(s1[t]), "+v"(s2[t]));
        if constexpr (MODE == 1) { ax[0] = axn[0]; ax[1] = axn[1]; }
      }
      // image -> HBM: 8 lanes cover one 128-byte row piece of this pair's 64 channels
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mm = 0; mm < 2; ++mm) {
        const int pr = rp + 8 * mm;
        (void)pr;
        const u32x4 u0 = *reinterpret_cast<const u32x4*>(imgr + 8 * mm * IMG_RS);
        const u32x4 u1 = *reinterpret_cast<const u32x4*>(imgr + 8 * mm * IMG_RS + 4);
        u32x4 ev, od;
        ev[0] = __builtin_amdgcn_perm(u0[1], u0[0], 0x05040100u); od[0] = __builtin_amdgcn_perm(u0[1], u0[0], 0x07060302u);
        ev[1] = __builtin_amdgcn_perm(u0[3], u0[2], 0x05040100u); od[1] = __builtin_amdgcn_perm(u0[3], u0[2], 0x07060302u);
        ev[2] = __builtin_amdgcn_perm(u1[1], u1[0], 0x05040100u); od[2] = __builtin_amdgcn_perm(u1[1], u1[0], 0x07060302u);
        ev[3] = __builtin_amdgcn_perm(u1[3], u1[2], 0x05040100u); od[3] = __builtin_amdgcn_perm(u1[3], u1[2], 0x07060302u);
        // rows 2 pr and 2 pr + 1; rows >= V lie outside the frame's descriptor: dropped by the bounds check
        __builtin_amdgcn_raw_buffer_store_b128(ev, rz, zl + (unsigned)(64 * pr2) * 2u + (unsigned)(16 * mm) * zrow_b, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(od, rz, zl + (unsigned)(64 * pr2) * 2u + (unsigned)(16 * mm + 1) * zrow_b, 0, 0);
      }
    }
  };

  // ---- the walk: segments of P.seg logical frames; ring entry e of output frame m is input frame IM m + off0 + e.
  //      The IM newest frames of a step are requested TWO steps before the step that first touches them (three
  //      landing sets, the loop unrolled by three), so a load has two frames of work to land. ----
  auto loadq = [&](const rsrc_t& rq, int f, u32x4& dst) __attribute__((always_inline)) {
    // frames outside [0, Tin): the unsigned offset lies beyond the sequence's descriptor -> zeros
    dst = __builtin_amdgcn_raw_buffer_load_b128(rq, (unsigned)f * qfrm_b + ql, 0, 0);
  };
  const int nwv = gridDim.x * 8;
  for (int sg = blockIdx.x * 8 + wave8; sg < P.nseg; sg += nwv) {
    const int n = sg / P.nseg_seq;
    const int m_lo = (sg - n * P.nseg_seq) * P.seg, m_hi = min(P.Mlog, m_lo + P.seg);
    const rsrc_t rq = make_rsrc(qg + (size_t)n * P.Tin * V * Wp, (unsigned)(P.Tin * V * Wp) * 2u);
    u32x4 R[NTAP], L0[IM], L1[IM], L2[IM];
    const int f0 = IM * m_lo + P.off0;
#pragma unroll
    for (int e = 0; e < NTAP; ++e) loadq(rq, f0 + e, R[e]);
    // tails of steps m_lo + 1 (L0) and m_lo + 2 (L1): frames f0 + IM k + NTAP - IM + i
#pragma unroll
    for (int i = 0; i < IM; ++i) { loadq(rq, f0 + IM + NTAP - IM + i, L0[i]); loadq(rq, f0 + 2 * IM + NTAP - IM + i, L1[i]); }
    auto step = [&](int m, u32x4 (&Lnew)[IM], const u32x4 (&Lold)[IM]) __attribute__((always_inline)) {
      // (m >= m_hi: a ghost step -- same loads, same stores into empty descriptors, nothing added to the sums)
      // Lnew <- tails of step m + 3;  frame m;  then the ring advances by IM frames and takes the tails of step m + 1 (Lold)
      const int fm = IM * m + P.off0;
#pragma unroll
      for (int i = 0; i < IM; ++i) loadq(rq, fm + 3 * IM + NTAP - IM + i, Lnew[i]);
      __builtin_amdgcn_sched_barrier(0);
      frame(n, m, m < m_hi, R);
#pragma unroll
      for (int e = 0; e < NTAP - IM; ++e) R[e] = R[e + IM];
#pragma unroll
      for (int i = 0; i < IM; ++i) R[NTAP - IM + i] = Lold[i];
    };
    for (int m = m_lo; m < m_hi; m += 3) {
      step(m, L2, L0);
      step(m + 1, L0, L1);
      step(m + 2, L1, L2);
    }
  }

  // ---- sums: lanes -> LDS -> fp64 atomics ----
  if (P.stats) {
#pragma unroll
    for (int t = 0; t < 2 * NPAIR; ++t) {
      const float a = s1[t] + __shfl_xor(s1[t], 32), b = s2[t] + __shfl_xor(s2[t], 32);
      if (h == 0) {
        atomicAdd(&stat[32 * t + c], a);
        atomicAdd(&stat[C + 32 * t + c], b);
      }
    }
    __syncthreads();
    double* dst = P.stats + (size_t)(blockIdx.x % P.stats_rep) * 2 * C;
    for (int i = tid; i < C; i += RC_NTH) {
      atomic_add_f64(dst + i, (double)stat[i]);
      atomic_add_f64(dst + C + i, (double)stat[C + i]);
    }
  }
  bn_tail_run(P.tail, gridDim.x, reinterpret_cast<unsigned*>(smem));
}

template <typename T, int NPAIR, int IM, int MODE>
int bout_launch(BoutParams P, int grid_cap, hipStream_t stream) {
  auto kfn = bneck_out_kernel<T, NPAIR, IM, MODE>;
  constexpr int C = 64 * NPAIR;
  static_assert((size_t)(NTAP * 256 + C * 16) * 4 <= (size_t)8 * IMG_BYTES, "the setup scratch lives in the image region");
  const size_t lds = (size_t)(NTAP + 2 * NPAIR) * 64 * 16 + (size_t)8 * IMG_BYTES + (size_t)(2 + 4 + 1) * C * 4 + (size_t)2 * 64 * 16;
  static std::atomic<unsigned long long> optin{0};
  if (int ea = istgcn_lds_optin((const void*)kfn, optin)) return ea;
  int g = grid_cap > 0 ? grid_cap : istgcn_resident_blocks((const void*)kfn, RC_NTH, lds);
  // segments: about one per wave, never shorter than 12 frames (the ring's warm-up is 15 + 2 IM narrow loads)
  const long long frames = (long long)P.NM * P.Mlog;
  long long seg = (frames + (long long)g * 8 - 1) / ((long long)g * 8);
  if (seg < 12) seg = 12;
  if (seg > P.Mlog) seg = P.Mlog;
  P.nseg_seq = (int)((P.Mlog + seg - 1) / seg);
  P.seg = (int)((P.Mlog + P.nseg_seq - 1) / P.nseg_seq);         // equal segments within a sequence
  P.nseg = P.nseg_seq * P.NM;
  if (g > (P.nseg + 7) / 8) g = (P.nseg + 7) / 8;
  if (g < 1) g = 1;
  ISTGCN_LAUNCH(kfn, dim3(g), dim3(RC_NTH), lds, stream, P);
  ISTGCN_CHECK_LAUNCH();
  return ISTGCN_OK;
}

template <typename T, int NPAIR>
int bout_im(const BoutParams& P, int in_mul, int mode, int grid_cap, hipStream_t stream) {
  if (in_mul == 1) return mode ? bout_launch<T, NPAIR, 1, 1>(P, grid_cap, stream) : bout_launch<T, NPAIR, 1, 0>(P, grid_cap, stream);
  if (in_mul == 2 && mode == 0) return bout_launch<T, NPAIR, 2, 0>(P, grid_cap, stream);
  return ISTGCN_EINVAL;
}

template <typename T>
int bout_T(const BoutParams& P, int in_mul, int mode, int grid_cap, hipStream_t stream) {
  switch (P.C) {
    case 64: return bout_im<T, 1>(P, in_mul, mode, grid_cap, stream);
    case 128: return bout_im<T, 2>(P, in_mul, mode, grid_cap, stream);
    case 256: return bout_im<T, 4>(P, in_mul, mode, grid_cap, stream);
  }
  return ISTGCN_EINVAL;
}

// ======================================================================================================================
// bneck_wgrad: weight gradients of the two 1x1 convolutions (wide x narrow), a FLAT stream over positions
//   dW(n, c) = sum_p nrw[p][n] * pre(wide[p][c]),   db = sum_p wide[p][:]  or  sum_p nrw[p][:]
// Both operands of a contraction over POSITIONS must carry positions on the k axis, but memory has them row-major
// (channels contiguous).  A row vector is, however, the A operand of a product over channels: one MFMA against an identity
// fragment delivers the 32-row tile TRANSPOSED in the accumulator layout (channel on the lane, rows in the registers;
// exact: x * 1.0 in fp32), and that tile converted pairwise is an operand with k = rows -- for BOTH sides, so the
// contraction chains from two accumulator tiles without a single LDS access or transposed load:
//   Tn[p][n] = nrw_tile . I          (1 MFMA)      -> A operand (lane = n)
//   Tw[p][c] = wide_tile . I_s       (2 per tile)  -> B operand (lane = c);  column sums of Tw / Tn = the bias gradients
//   dW^T... D[n][c] += Tn^T . Tw     (2 k-steps per 32-channel tile)
// ======================================================================================================================
struct BwgParams {
  const void* wide; const void* nrw; const float* pre; float* dW; float* db; float* ws;
  long long ws_slice, rows;
  int C, Wp, pre_relu, wide_is_out, db_wide, ntiles;
};

template <typename T, int S>
__global__ __launch_bounds__(256, 2) void bneck_wgrad_kernel(const BwgParams P) {
  using E = Elem<T>;
  typedef typename E::frag frag_t;
  constexpr int C = 16 * S, NT = C / 32;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u32x4* idl = reinterpret_cast<u32x4*>(smem);                          // [2][64] identity fragments
  float* pre_l = reinterpret_cast<float*>(smem + 2 * 64 * 16);          // [2][C]
  float* red = pre_l + 2 * C;                                           // [NT][16][32] dW partial sums, then [C] + [16] bias sums
  float* redb = red + NT * 16 * 32;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 31, h = lane >> 5;
  if (wave < 2) {
    frag_t f;
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = E::from_f(16 * wave + 8 * h + e == c ? 1.f : 0.f);
    idl[wave * 64 + lane] = __builtin_bit_cast(u32x4, f);
  }
  for (int i = tid; i < 2 * C; i += 256) pre_l[i] = P.pre ? P.pre[i] : (i < C ? 1.f : 0.f);
  for (int i = tid; i < NT * 16 * 32 + C + 16; i += 256) red[i] = 0.f;
  __syncthreads();

  const rsrc_t rw = make_rsrc(P.wide, (unsigned)(P.rows * C * 2));
  const rsrc_t rn = make_rsrc(P.nrw, (unsigned)(P.rows * P.Wp * 2));
  const unsigned wl_ = (unsigned)(c * C + 8 * h) * 2u;
  const unsigned nl_ = 8 * h < P.Wp ? (unsigned)(c * P.Wp + 8 * h) * 2u : OOB;
  const bool pre = P.pre != nullptr, relu = P.pre_relu != 0;
  const int nw = gridDim.x * 4;

  f32x16 acc[NT];
  float dbw[NT], dbn = 0.f;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    dbw[t] = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
  }

  // 256 channels: 128 accumulator registers leave no room for two whole-row register sets -- the row is taken in four
  // quarters of four k-steps (register sets A / B alternate over the quarters; the narrow tile is loaded with each)
  constexpr int SH = S > 8 ? 4 : S, NH = S / SH;
  auto loadt = [&](int tile, int half, u32x4 (&xf)[SH], u32x4& nf) __attribute__((always_inline)) {
    const unsigned base = (unsigned)tile * (unsigned)(32 * C * 2) + wl_ + (unsigned)(half * SH * 32);
#pragma unroll
    for (int s = 0; s < SH; ++s) xf[s] = __builtin_amdgcn_raw_buffer_load_b128(rw, base + 32u * s, 0, LDW);
    nf = __builtin_amdgcn_raw_buffer_load_b128(rn, nl_ == OOB ? OOB : (unsigned)tile * (unsigned)(32 * P.Wp * 2) + nl_, 0, 0);
  };
  auto work = [&](auto half_c, u32x4 (&xf)[SH], const u32x4& nf) __attribute__((always_inline)) {
    constexpr int HALF = decltype(half_c)::value;
    int ln = lane;
    asm volatile("" : "+v"(ln));                             // (keeps the identity-fragment reads inside the loop)
    f32x16 Tn;
#pragma unroll
    for (int i = 0; i < 16; ++i) Tn[i] = 0.f;
    mma_kgroup(Tn, __builtin_bit_cast(frag_t, nf), __builtin_bit_cast(frag_t, idl[ln]));
    u32x4 an[2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int d = 0; d < 4; ++d) an[s2][d] = pack2<T>(Tn[8 * s2 + 2 * d], Tn[8 * s2 + 2 * d + 1]);
    if constexpr (HALF == 0) {
      float a = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) a += Tn[i];
      dbn += a;
    }
#pragma unroll
    for (int tl = 0; tl < SH / 2; ++tl) {
      constexpr int T0 = HALF * (SH / 2);
      __builtin_amdgcn_sched_barrier(0);                     // one channel tile at a time (overlapped tiles spill)
      f32x16 Tw;
#pragma unroll
      for (int i = 0; i < 16; ++i) Tw[i] = 0.f;
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const int sl = 2 * tl + s2, s = HALF * SH + sl;
        u32x4 v = xf[sl];
        if (pre) {
          const f32x4 sc0 = *reinterpret_cast<const f32x4*>(pre_l + 16 * s + 8 * h), sc1 = *reinterpret_cast<const f32x4*>(pre_l + 16 * s + 8 * h + 4);
          const f32x4 sh0 = *reinterpret_cast<const f32x4*>(pre_l + C + 16 * s + 8 * h), sh1 = *reinterpret_cast<const f32x4*>(pre_l + C + 16 * s + 8 * h + 4);
#pragma unroll
          for (int d = 0; d < 4; ++d) {
            float lo, hi;
            unpack2<T>(v[d], lo, hi);
            const float s0 = d < 2 ? sc0[2 * d] : sc1[2 * d - 4], s1 = d < 2 ? sc0[2 * d + 1] : sc1[2 * d - 3];
            const float t0 = d < 2 ? sh0[2 * d] : sh1[2 * d - 4], t1 = d < 2 ? sh0[2 * d + 1] : sh1[2 * d - 3];
            lo = fmaf(lo, s0, t0);
            hi = fmaf(hi, s1, t1);
            if (relu) { lo = fmaxf(lo, 0.f); hi = fmaxf(hi, 0.f); }
            v[d] = pack2<T>(lo, hi);
          }
        }
        mma_kgroup(Tw, __builtin_bit_cast(frag_t, v), __builtin_bit_cast(frag_t, idl[s2 * 64 + ln]));
      }
      u32x4 bw[2];
      float a = 0.f;
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int d = 0; d < 4; ++d) bw[s2][d] = pack2<T>(Tw[8 * s2 + 2 * d], Tw[8 * s2 + 2 * d + 1]);
#pragma unroll
      for (int i = 0; i < 16; ++i) a += Tw[i];
      dbw[T0 + tl] += a;
      asm volatile("" : "+v"(dbw[T0 + tl]));                 // (the sum is needed at the end only: pinned here, else Tw lives on)
      mma_kgroup(acc[T0 + tl], __builtin_bit_cast(frag_t, an[0]), __builtin_bit_cast(frag_t, bw[0]));
      mma_kgroup(acc[T0 + tl], __builtin_bit_cast(frag_t, an[1]), __builtin_bit_cast(frag_t, bw[1]));
    }
  };
  int tile = blockIdx.x * 4 + wave;
  if (tile < P.ntiles) {
    u32x4 xa[SH], xb[SH], na, nb;
    loadt(tile, 0, xa, na);
    if constexpr (NH == 1) {
      using H0 = std::integral_constant<int, 0>;
      for (;;) {
        const int t2 = tile + nw;
        const bool more = t2 < P.ntiles;
        loadt(more ? t2 : tile, 0, xb, nb);
        __builtin_amdgcn_sched_barrier(0);
        work(H0{}, xa, na);
        if (!more) break;
        const int t3 = t2 + nw;
        const bool more2 = t3 < P.ntiles;
        loadt(more2 ? t3 : t2, 0, xa, na);
        __builtin_amdgcn_sched_barrier(0);
        work(H0{}, xb, nb);
        if (!more2) break;
        tile = t3;
      }
    } else {
      static_assert(NH == 1 || NH == 4, "quarters");
      for (;;) {
        const int t2 = tile + nw;
        const bool more = t2 < P.ntiles;
        loadt(tile, 1, xb, nb);
        __builtin_amdgcn_sched_barrier(0);
        work(std::integral_constant<int, 0>{}, xa, na);
        loadt(tile, NH > 2 ? 2 : 0, xa, na);
        __builtin_amdgcn_sched_barrier(0);
        work(std::integral_constant<int, (NH > 1 ? 1 : 0)>{}, xb, nb);
        loadt(tile, NH > 3 ? 3 : 0, xb, nb);
        __builtin_amdgcn_sched_barrier(0);
        work(std::integral_constant<int, (NH > 2 ? 2 : 0)>{}, xa, na);
        loadt(more ? t2 : tile, 0, xa, na);
        __builtin_amdgcn_sched_barrier(0);
        work(std::integral_constant<int, (NH > 3 ? 3 : 0)>{}, xb, nb);
        if (!more) break;
        tile = t2;
      }
    }
  }

  // ---- the four waves' sums -> LDS (one wave at a time: each lane owns its slots) -> this workgroup's slice ----
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {                         // registers 0..7: narrow channels (i & 3) + 8 (i >> 2) + 4 h < 16
          const int n = (i & 3) + 8 * (i >> 2) + 4 * h;
          red[(t * 16 + n) * 32 + c] += acc[t][i];
        }
        const float a = dbw[t] + __shfl_xor(dbw[t], 32);
        if (h == 0) redb[32 * t + c] += a;
      }
      const float a = dbn + __shfl_xor(dbn, 32);
      if (h == 0 && c < 16) redb[C + c] += a;
    }
    __syncthreads();
  }
  const int n0 = C * P.Wp, n1 = P.db_wide ? C : P.Wp;
  float* sl = P.ws + (size_t)blockIdx.x * P.ws_slice;
  for (int i = tid; i < n0; i += 256) {
    int n, o;
    if (P.wide_is_out) { o = i / P.Wp; n = i - o * P.Wp; } else { n = i / C; o = i - n * C; }
    sl[i] = red[((o >> 5) * 16 + n) * 32 + (o & 31)];
  }
  for (int i = tid; i < n1; i += 256) sl[n0 + i] = P.db_wide ? redb[i] : redb[C + i];
}

extern "C" int istgcn_wgrad_reduce(const float* ws, long long slice, int nsl, float* d0, int n0, float* d1, int n1, void* stream);

template <typename T, int S>
int bwg_launch(BwgParams P, int grid_cap, long long ws_floats, hipStream_t stream) {
  auto kfn = bneck_wgrad_kernel<T, S>;
  constexpr int C = 16 * S;
  const size_t lds = (size_t)2 * 64 * 16 + (size_t)2 * C * 4 + (size_t)((C / 32) * 16 * 32 + C + 16) * 4;
  static std::atomic<unsigned long long> optin{0};
  if (int ea = istgcn_lds_optin((const void*)kfn, optin)) return ea;
  int g = grid_cap > 0 ? grid_cap : istgcn_resident_blocks((const void*)kfn, 256, lds);
  if (g > (P.ntiles + 3) / 4) g = (P.ntiles + 3) / 4;
  if (g < 1) g = 1;
  const int n0 = C * P.Wp, n1 = P.db_wide ? C : P.Wp;
  P.ws_slice = n0 + n1;
  while (g > 1 && (long long)g * P.ws_slice > ws_floats) g >>= 1;
  if ((long long)g * P.ws_slice > ws_floats) return ISTGCN_EINVAL;
  ISTGCN_LAUNCH(kfn, dim3(g), dim3(256), lds, stream, P);
  ISTGCN_CHECK_LAUNCH();
  return istgcn_wgrad_reduce(P.ws, P.ws_slice, g, P.dW, n0, P.db, P.db ? n1 : 0, stream);
}

// ======================================================================================================================
// bneck_wgrad_taps: weight gradient of the narrow temporal conv,  dWt[j][n'][n] += sum_{seq, m, v} dy[m][v][n'] q[IM m + off0 + j][v][n]
// (+ its bias gradient sum dy).  The frame walk of bneck_out with the ring holding TRANSPOSED frames: a narrow frame
// (<= 32 rows x <= 16 channels) through two 16x16x32 MFMAs against an identity fragment comes out with the channel on
// the lane and its 32 positions in eight registers -- converted pairwise, one operand register quad with k = positions,
// the same size as the raw row vector.  Each new q frame is transposed ONCE when it enters the ring; an output frame then
// costs one MFMA per tap (K = 32 = the whole frame) into a 4-register accumulator tile per tap.
// ======================================================================================================================
typedef float f32x4v __attribute__((ext_vector_type(4)));
__device__ static inline f32x4v mma16(const bf16x8& a, const bf16x8& b, const f32x4v& c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
__device__ static inline f32x4v mma16(const f16x8& a, const f16x8& b, const f32x4v& c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }

struct BwtParams {
  const void* dy; const void* q; float* dW; float* db; float* ws;
  long long ws_slice;
  int NM, Tin, Tz, V, Wp, ntaps, off0;
  int seg, nseg_seq, nseg;
};

template <typename T, int IM>
__global__ __launch_bounds__(RC_NTH, 2) void bneck_wgrad_taps_kernel(const BwtParams P) {
  using E = Elem<T>;
  typedef typename E::frag frag_t;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* red = reinterpret_cast<float*>(smem);               // [NTAP][16][16] + [16]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l16 = lane & 15, g = lane >> 4;
  const int V = P.V, Wp = P.Wp;
  for (int i = tid; i < NTAP * 256 + 16; i += RC_NTH) red[i] = 0.f;
  // identity fragment (B operand of the transposing product): column n = l16, k = 8 g + e
  u32x4 idf;
  {
    frag_t f;
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = E::from_f(8 * g + e == l16 ? 1.f : 0.f);
    idf = __builtin_bit_cast(u32x4, f);
  }
  __syncthreads();

  const T* dyg = reinterpret_cast<const T*>(P.dy);
  const T* qg = reinterpret_cast<const T*>(P.q);
  const unsigned frm_b = (unsigned)(V * Wp) * 2u;
  // raw row vectors of a frame: rows l16 (lo) and 16 + l16 (hi), channels 8 g .. 8 g + 7
  const unsigned lo_off = (l16 < V && 8 * g < Wp) ? (unsigned)(l16 * Wp + 8 * g) * 2u : OOB;
  const unsigned hi_off = (16 + l16 < V && 8 * g < Wp) ? (unsigned)((16 + l16) * Wp + 8 * g) * 2u : OOB;

  // frame (two raw vectors) -> transposed operand: k slot e of lane group g = position 4 g + (e & 3) + 16 (e >> 2)
  auto transpose = [&](const u32x4& lo, const u32x4& hi, float& colsum) __attribute__((always_inline)) {
    f32x4v z = {0.f, 0.f, 0.f, 0.f};
    const f32x4v dl = mma16(__builtin_bit_cast(frag_t, lo), __builtin_bit_cast(frag_t, idf), z);
    const f32x4v dh = mma16(__builtin_bit_cast(frag_t, hi), __builtin_bit_cast(frag_t, idf), z);
    colsum += (dl[0] + dl[1]) + (dl[2] + dl[3]) + (dh[0] + dh[1]) + (dh[2] + dh[3]);
    return u32x4{pack2<T>(dl[0], dl[1]), pack2<T>(dl[2], dl[3]), pack2<T>(dh[0], dh[1]), pack2<T>(dh[2], dh[3])};
  };

  f32x4v acc[NTAP];
#pragma unroll
  for (int j = 0; j < NTAP; ++j) acc[j] = f32x4v{0.f, 0.f, 0.f, 0.f};
  float dbs = 0.f, dummy = 0.f;

  const int nwv = gridDim.x * 8;
  for (int sg = blockIdx.x * 8 + wave8; sg < P.nseg; sg += nwv) {
    const int n = sg / P.nseg_seq;
    const int m_lo = (sg - n * P.nseg_seq) * P.seg, m_hi = min(P.Tz, m_lo + P.seg);
    const rsrc_t rq = make_rsrc(qg + (size_t)n * P.Tin * V * Wp, (unsigned)(P.Tin * V * Wp) * 2u);
    const T* dyn = dyg + (size_t)n * P.Tz * V * Wp;
    auto loadq = [&](int f, u32x4 (&dst)[2]) __attribute__((always_inline)) {
      dst[0] = __builtin_amdgcn_raw_buffer_load_b128(rq, lo_off == OOB ? OOB : (unsigned)f * frm_b + lo_off, 0, 0);
      dst[1] = __builtin_amdgcn_raw_buffer_load_b128(rq, hi_off == OOB ? OOB : (unsigned)f * frm_b + hi_off, 0, 0);
    };
    auto loady = [&](int m, u32x4 (&dst)[2]) __attribute__((always_inline)) {
      // (frames m >= m_hi: ghost steps -- an empty descriptor, zeros)
      const rsrc_t ry = make_rsrc(dyn + (size_t)(m < m_hi ? m : 0) * V * Wp, m < m_hi ? frm_b : 0u);
      dst[0] = __builtin_amdgcn_raw_buffer_load_b128(ry, lo_off, 0, 0);
      dst[1] = __builtin_amdgcn_raw_buffer_load_b128(ry, hi_off, 0, 0);
    };
    u32x4 R[NTAP];
    u32x4 L0[IM][2], L1[IM][2], L2[IM][2], Y0[2], Y1[2], Y2[2];
    const int f0 = IM * m_lo + P.off0;
    {
      u32x4 tmp[2];
#pragma unroll
      for (int e = 0; e < NTAP; ++e) { loadq(f0 + e, tmp); R[e] = transpose(tmp[0], tmp[1], dummy); }
    }
#pragma unroll
    for (int i = 0; i < IM; ++i) { loadq(f0 + IM + NTAP - IM + i, L0[i]); loadq(f0 + 2 * IM + NTAP - IM + i, L1[i]); }
    loady(m_lo, Y0);
    loady(m_lo + 1, Y1);
    auto step = [&](int m, u32x4 (&Lnew)[IM][2], const u32x4 (&Lold)[IM][2], u32x4 (&Ynew)[2], const u32x4 (&Ycur)[2]) __attribute__((always_inline)) {
      const int fm = IM * m + P.off0;
#pragma unroll
      for (int i = 0; i < IM; ++i) loadq(fm + 3 * IM + NTAP - IM + i, Lnew[i]);
      loady(m + 2, Ynew);
      __builtin_amdgcn_sched_barrier(0);
      const u32x4 ad = transpose(Ycur[0], Ycur[1], dbs);
#pragma unroll
      for (int j = 0; j < NTAP; ++j) acc[j] = mma16(__builtin_bit_cast(frag_t, ad), __builtin_bit_cast(frag_t, R[j]), acc[j]);
#pragma unroll
      for (int e = 0; e < NTAP - IM; ++e) R[e] = R[e + IM];
#pragma unroll
      for (int i = 0; i < IM; ++i) R[NTAP - IM + i] = transpose(Lold[i][0], Lold[i][1], dummy);
    };
    for (int m = m_lo; m < m_hi; m += 3) {
      step(m, L2, L0, Y2, Y0);
      step(m + 1, L0, L1, Y0, Y1);
      step(m + 2, L1, L2, Y1, Y2);
    }
  }

  // ---- the eight waves' sums -> LDS (one wave at a time) -> this workgroup's slice ----
  {
    // column sums of the transposed dy tiles: lane (n' = l16, g) summed its rows; the four lane groups together = all rows
    dbs += __shfl_xor(dbs, 16);
    dbs += __shfl_xor(dbs, 32);
  }
  for (int w = 0; w < 8; ++w) {
    if (wave8 == w) {
#pragma unroll
      for (int j = 0; j < NTAP; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) red[(j * 16 + 4 * g + i) * 16 + l16] += acc[j][i];     // D: column n = l16, row n' = 4 g + i
      if (g == 0) red[NTAP * 256 + l16] += dbs;
    }
    __syncthreads();
  }
  const int n0 = P.ntaps * Wp * Wp;
  float* sl = P.ws + (size_t)blockIdx.x * P.ws_slice;
  for (int i = tid; i < n0; i += RC_NTH) {
    const int j = i / (Wp * Wp), r = (i / Wp) % Wp, cc = i % Wp;
    sl[i] = red[(j * 16 + r) * 16 + cc];
  }
  for (int i = tid; i < Wp; i += RC_NTH) sl[n0 + i] = red[NTAP * 256 + i];
}

template <typename T, int IM>
int bwt_launch(BwtParams P, int grid_cap, long long ws_floats, hipStream_t stream) {
  auto kfn = bneck_wgrad_taps_kernel<T, IM>;
  const size_t lds = (size_t)(NTAP * 256 + 16) * 4;
  static std::atomic<unsigned long long> optin{0};
  if (int ea = istgcn_lds_optin((const void*)kfn, optin)) return ea;
  int g = grid_cap > 0 ? grid_cap : istgcn_resident_blocks((const void*)kfn, RC_NTH, lds);
  const long long frames = (long long)P.NM * P.Tz;
  long long seg = (frames + (long long)g * 8 - 1) / ((long long)g * 8);
  if (seg < 12) seg = 12;
  if (seg > P.Tz) seg = P.Tz;
  P.nseg_seq = (int)((P.Tz + seg - 1) / seg);
  P.seg = (int)((P.Tz + P.nseg_seq - 1) / P.nseg_seq);
  P.nseg = P.nseg_seq * P.NM;
  if (g > (P.nseg + 7) / 8) g = (P.nseg + 7) / 8;
  if (g < 1) g = 1;
  const int n0 = P.ntaps * P.Wp * P.Wp, n1 = P.Wp;
  P.ws_slice = n0 + n1;
  if ((long long)g * P.ws_slice > ws_floats) return ISTGCN_EINVAL;
  ISTGCN_LAUNCH(kfn, dim3(g), dim3(RC_NTH), lds, stream, P);
  ISTGCN_CHECK_LAUNCH();
  return istgcn_wgrad_reduce(P.ws, P.ws_slice, g, P.dW, n0, P.db, P.db ? n1 : 0, stream);
}

// ======================================================================================================================
// bneck_bwd_in: the backward pass INTO the chain in one stream (64 / 128 channels):
//   dz  = a[c] * dropmask * dres + b[c] * z + c[c]      the elementwise half of tcn_end's BatchNorm backward (what
//                                                       istgcn_affine2 would write as a tensor), formed in registers from
//                                                       the two row vectors, rounded to the storage type like the tensor was
//   dyb = We^T dz                                       (bneck_in's product: dz vectors as the B operand)
//   dWe += dz^T yb,  dbe += sum dz                      (bneck_wgrad's products: dz vectors transposed by identity MFMAs)
// Replaces affine2 (3 wide passes) + bneck_in (1) + bneck_wgrad (1) by 2 wide passes.  The dropout mask is regenerated
// with the forward's Philox stream (dropout.hpp) per 8-channel vector.  Rows are taken in passes of four k-steps
// (64 channels) with two register sets alternating; the narrow accumulator of dyb lives across the passes of a tile.
// ======================================================================================================================
struct BbiParams {
  const void* dres; const void* z; const float* abc; const void* nrw; const float* W; void* dyb; float* dW; float* db; float* ws;
  long long w_rs, w_cs, ws_slice, rows;
  DropCfg D;
  int C, Wn, Wp, ntiles;
};

template <typename T, int S>
__global__ __launch_bounds__(256, 2) void bneck_bwd_in_kernel(const BbiParams P) {
  using E = Elem<T>;
  typedef typename E::frag frag_t;
  constexpr int C = 16 * S, NT = C / 32, SH = 4, NH = S / SH;
  static_assert(NH == 1 || NH == 2, "64 or 128 channels");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u32x4* idl = reinterpret_cast<u32x4*>(smem);                          // [2][64] identity fragments
  u32x4* wl = idl + 2 * 64;                                             // [S][64] A fragments of W (dyb = W dz)
  float* abc_l = reinterpret_cast<float*>(wl + S * 64);                 // [3][C]
  float* red = abc_l + 3 * C;                                           // [NT][16][32] + [C] + [16]; setup: [16][C] copy of W
  float* redb = red + NT * 16 * 32;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 31, h = lane >> 5;
  uint32_t dk0, dk1;
  drop_key(P.D, dk0, dk1);
  const bool drop_on = P.D.on != 0;
  if (wave < 2) {
    frag_t f;
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = E::from_f(16 * wave + 8 * h + e == c ? 1.f : 0.f);
    idl[wave * 64 + lane] = __builtin_bit_cast(u32x4, f);
  }
  for (int i = tid; i < 3 * C; i += 256) abc_l[i] = P.abc[i];
  for (int i = tid; i < 16 * C; i += 256) {
    const int n = i / C, cc = i - n * C;
    red[i] = n < P.Wn ? P.W[n * P.w_rs + cc * P.w_cs] : 0.f;
  }
  __syncthreads();
  for (int s = wave; s < S; s += 4) {
    frag_t f;
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = E::from_f(c < 16 ? red[c * C + 16 * s + 8 * h + j] : 0.f);
    wl[s * 64 + lane] = __builtin_bit_cast(u32x4, f);
  }
  __syncthreads();
  for (int i = tid; i < NT * 16 * 32 + C + 16; i += 256) red[i] = 0.f;
  __syncthreads();

  const rsrc_t rd = make_rsrc(P.dres, (unsigned)(P.rows * C * 2));
  const rsrc_t rz = make_rsrc(P.z, (unsigned)(P.rows * C * 2));
  const rsrc_t rn = make_rsrc(P.nrw, (unsigned)(P.rows * P.Wp * 2));
  const rsrc_t ry = make_rsrc(P.dyb, (unsigned)(P.rows * P.Wp * 2));
  const unsigned wl_ = (unsigned)(c * C + 8 * h) * 2u;
  const unsigned nl_ = 8 * h < P.Wp ? (unsigned)(c * P.Wp + 8 * h) * 2u : OOB;
  const unsigned yl = (unsigned)(c * P.Wp + 4 * h) * 2u;
  const bool wide16 = P.Wp == 16;
  const int nw = gridDim.x * 4;

  f32x16 acc[NT], accy;
  float dbw[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    dbw[t] = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
  }

  auto loadt = [&](int tile, int q, u32x4 (&df)[SH], u32x4 (&zf)[SH], u32x4& nf) __attribute__((always_inline)) {
    const unsigned base = (unsigned)tile * (unsigned)(32 * C * 2) + wl_ + (unsigned)(q * SH * 32);
#pragma unroll
    for (int s = 0; s < SH; ++s) {
      df[s] = __builtin_amdgcn_raw_buffer_load_b128(rd, base + 32u * s, 0, LDW);
      zf[s] = __builtin_amdgcn_raw_buffer_load_b128(rz, base + 32u * s, 0, LDW);
    }
    nf = __builtin_amdgcn_raw_buffer_load_b128(rn, nl_ == OOB ? OOB : (unsigned)tile * (unsigned)(32 * P.Wp * 2) + nl_, 0, 0);
  };
  auto work = [&](auto q_c, int tile, u32x4 (&df)[SH], u32x4 (&zf)[SH], const u32x4& nf) __attribute__((always_inline)) {
    constexpr int Q = decltype(q_c)::value;
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const long long row = (long long)tile * 32 + c;
    const bool rowok = row < P.rows;                          // (rows past the tensor: dz must be zero, not c[c])
    f32x16 Tn;
#pragma unroll
    for (int i = 0; i < 16; ++i) Tn[i] = 0.f;
    mma_kgroup(Tn, __builtin_bit_cast(frag_t, nf), __builtin_bit_cast(frag_t, idl[ln]));
    u32x4 an[2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int d = 0; d < 4; ++d) an[s2][d] = pack2<T>(Tn[8 * s2 + 2 * d], Tn[8 * s2 + 2 * d + 1]);
    if constexpr (Q == 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) accy[i] = 0.f;
    }
#pragma unroll
    for (int tl = 0; tl < SH / 2; ++tl) {
      constexpr int T0 = Q * (SH / 2);
      __builtin_amdgcn_sched_barrier(0);
      f32x16 Tw;
#pragma unroll
      for (int i = 0; i < 16; ++i) Tw[i] = 0.f;
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const int sl = 2 * tl + s2, s = Q * SH + sl;
        const int ch0 = 16 * s + 8 * h;
        float m[8];
        if (drop_on) drop_scales<8>(m, (size_t)(row * C + ch0), P.D.thr, P.D.inv_keep, dk0, dk1);
        u32x4 v;
#pragma unroll
        for (int d4 = 0; d4 < 2; ++d4) {
          const f32x4 ca = *reinterpret_cast<const f32x4*>(abc_l + ch0 + 4 * d4), cb = *reinterpret_cast<const f32x4*>(abc_l + C + ch0 + 4 * d4);
          const f32x4 cc = *reinterpret_cast<const f32x4*>(abc_l + 2 * C + ch0 + 4 * d4);
#pragma unroll
          for (int d2 = 0; d2 < 2; ++d2) {
            const int d = 2 * d4 + d2;
            float g0, g1, z0, z1;
            unpack2<T>(df[sl][d], g0, g1);
            unpack2<T>(zf[sl][d], z0, z1);
            if (drop_on) { g0 *= m[2 * d]; g1 *= m[2 * d + 1]; }
            // (the same operation order as affine2_kernel: (d * mask) * a, then + x * b + c)
            float r0 = g0 * ca[2 * d2] + (z0 * cb[2 * d2] + cc[2 * d2]);
            float r1 = g1 * ca[2 * d2 + 1] + (z1 * cb[2 * d2 + 1] + cc[2 * d2 + 1]);
            v[d] = rowok ? pack2<T>(r0, r1) : 0u;
          }
        }
        mma_kgroup(accy, __builtin_bit_cast(frag_t, wl[s * 64 + ln]), __builtin_bit_cast(frag_t, v));
        mma_kgroup(Tw, __builtin_bit_cast(frag_t, v), __builtin_bit_cast(frag_t, idl[s2 * 64 + ln]));
      }
      u32x4 bw[2];
      float a = 0.f;
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int d = 0; d < 4; ++d) bw[s2][d] = pack2<T>(Tw[8 * s2 + 2 * d], Tw[8 * s2 + 2 * d + 1]);
#pragma unroll
      for (int i = 0; i < 16; ++i) a += Tw[i];
      dbw[T0 + tl] += a;
      asm volatile("" : "+v"(dbw[T0 + tl]));
      mma_kgroup(acc[T0 + tl], __builtin_bit_cast(frag_t, an[0]), __builtin_bit_cast(frag_t, bw[0]));
      mma_kgroup(acc[T0 + tl], __builtin_bit_cast(frag_t, an[1]), __builtin_bit_cast(frag_t, bw[1]));
    }
    if constexpr (Q == NH - 1) {
      const unsigned yb = (unsigned)tile * (unsigned)(32 * P.Wp * 2) + yl;
      __builtin_amdgcn_raw_buffer_store_b64(u32x2{pack2<T>(accy[0], accy[1]), pack2<T>(accy[2], accy[3])}, ry, yb, 0, 0);
      __builtin_amdgcn_raw_buffer_store_b64(u32x2{pack2<T>(accy[4], accy[5]), pack2<T>(accy[6], accy[7])}, ry, wide16 ? yb + 16u : OOB, 0, 0);
    }
  };

  int tile = blockIdx.x * 4 + wave;
  if (tile < P.ntiles) {
    u32x4 da[SH], za[SH], db_[SH], zb[SH], na, nb;
    loadt(tile, 0, da, za, na);
    for (;;) {
      const int t2 = tile + nw;
      const bool more = t2 < P.ntiles;
      if constexpr (NH == 1) {
        loadt(more ? t2 : tile, 0, db_, zb, nb);
        __builtin_amdgcn_sched_barrier(0);
        work(std::integral_constant<int, 0>{}, tile, da, za, na);
        if (!more) break;
        const int t3 = t2 + nw;
        const bool more2 = t3 < P.ntiles;
        loadt(more2 ? t3 : t2, 0, da, za, na);
        __builtin_amdgcn_sched_barrier(0);
        work(std::integral_constant<int, 0>{}, t2, db_, zb, nb);
        if (!more2) break;
        tile = t3;
      } else {
        loadt(tile, 1, db_, zb, nb);
        __builtin_amdgcn_sched_barrier(0);
        work(std::integral_constant<int, 0>{}, tile, da, za, na);
        loadt(more ? t2 : tile, 0, da, za, na);
        __builtin_amdgcn_sched_barrier(0);
        work(std::integral_constant<int, NH - 1>{}, tile, db_, zb, nb);
        if (!more) break;
        tile = t2;
      }
    }
  }

  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int n = (i & 3) + 8 * (i >> 2) + 4 * h;
          red[(t * 16 + n) * 32 + c] += acc[t][i];
        }
        const float a = dbw[t] + __shfl_xor(dbw[t], 32);
        if (h == 0) redb[32 * t + c] += a;
      }
    }
    __syncthreads();
  }
  const int n0 = C * P.Wp;
  float* sl = P.ws + (size_t)blockIdx.x * P.ws_slice;
  for (int i = tid; i < n0; i += 256) {
    const int o = i / P.Wp, n = i - o * P.Wp;
    sl[i] = red[((o >> 5) * 16 + n) * 32 + (o & 31)];
  }
  for (int i = tid; i < C; i += 256) sl[n0 + i] = redb[i];
}

template <typename T, int S>
int bbi_launch(BbiParams P, int grid_cap, long long ws_floats, hipStream_t stream) {
  auto kfn = bneck_bwd_in_kernel<T, S>;
  constexpr int C = 16 * S;
  size_t red_b = (size_t)((C / 32) * 16 * 32 + C + 16) * 4;
  if (red_b < (size_t)16 * C * 4) red_b = (size_t)16 * C * 4;
  const size_t lds = (size_t)(2 + S) * 64 * 16 + (size_t)3 * C * 4 + red_b;
  static std::atomic<unsigned long long> optin{0};
  if (int ea = istgcn_lds_optin((const void*)kfn, optin)) return ea;
  int g = grid_cap > 0 ? grid_cap : istgcn_resident_blocks((const void*)kfn, 256, lds);
  if (g > (P.ntiles + 3) / 4) g = (P.ntiles + 3) / 4;
  if (g < 1) g = 1;
  const int n0 = C * P.Wp, n1 = C;
  P.ws_slice = n0 + n1;
  while (g > 1 && (long long)g * P.ws_slice > ws_floats) g >>= 1;
  if ((long long)g * P.ws_slice > ws_floats) return ISTGCN_EINVAL;
  ISTGCN_LAUNCH(kfn, dim3(g), dim3(256), lds, stream, P);
  ISTGCN_CHECK_LAUNCH();
  return istgcn_wgrad_reduce(P.ws, P.ws_slice, g, P.dW, n0, P.db, P.db ? n1 : 0, stream);
}

}  // namespace

// Shapes the two kernels serve (istgcn.h): 16-bit storage, V <= 32, wide side 64 / 128 / 256 channels, narrow side
// 1 .. 16 channels stored 8 or 16 wide.
extern "C" int istgcn_bneck_ok(int V, int C, int Wn, int Wp, int dtype) {
  return (dtype == 1 || dtype == 2) && V >= 1 && V <= 32 && (C == 64 || C == 128 || C == 256) && Wn >= 1 && Wn <= Wp &&
         (Wp == 8 || Wp == 16);
}

extern "C" int istgcn_bneck_in(const void* x, const float* W, long long w_rs, long long w_cs, const float* bias,
                               const float* pre, int pre_relu, void* y, long long rows, int C, int Wn, int Wp, int dtype,
                               int grid_cap, void* stream) {
  if (!x || !W || !y || rows < 0) return ISTGCN_EINVAL;
  if (!istgcn_bneck_ok(1, C, Wn, Wp, dtype)) return ISTGCN_EINVAL;
  if (rows * C * 2 >= (1ll << 32) - (1 << 20)) return ISTGCN_EINVAL;  // one buffer descriptor spans the tensor
  if (rows == 0) return ISTGCN_OK;
  BinParams P{};
  P.x = x; P.y = y; P.W = W; P.bias = bias; P.pre = pre; P.w_rs = w_rs; P.w_cs = w_cs; P.rows = rows;
  P.C = C; P.Wn = Wn; P.Wp = Wp; P.pre_relu = pre_relu;
  P.ntiles = (int)((rows + 31) / 32);
  hipStream_t st = (hipStream_t)stream;
#define GO(TT)                                              \
  switch (C) {                                              \
    case 64: return bin_launch<TT, 4>(P, grid_cap, st);     \
    case 128: return bin_launch<TT, 8>(P, grid_cap, st);    \
    case 256: return bin_launch<TT, 16>(P, grid_cap, st);   \
  }
  if (dtype == 1) { GO(__bf16) } else { GO(_Float16) }
#undef GO
  return ISTGCN_EINVAL;
}

extern "C" int istgcn_bneck_out(const void* q, const float* Wt, long long wt_ts, long long wt_rs, long long wt_cs,
                                const int* tap_sel, int ntaps, int off0, const float* bt, void* yb, const float* We,
                                long long we_rs, long long we_cs, const float* be, void* z, const void* aux,
                                const float* maux, double* stats, int stats_rep, int mode, int NM, int Tin, int Tout,
                                int Mlog, int V, int C, int Wn, int Wp, int in_mul, int out_mul, int out_off, int dtype,
                                int grid_cap, void* stream) {
  if (!q || !Wt || !tap_sel || !yb || !We || !z) return ISTGCN_EINVAL;
  if (!istgcn_bneck_ok(V, C, Wn, Wp, dtype)) return ISTGCN_EINVAL;
  if (ntaps < 1 || ntaps > NTAP || NM < 0 || Mlog < 0 || Tin < 1 || Tout < 1 || out_mul < 1 || out_off < 0) return ISTGCN_EINVAL;
  if (mode < 0 || mode > 1 || (mode == 1 && (!aux || !maux)) || (stats && stats_rep < 1)) return ISTGCN_EINVAL;
  if (in_mul != 1 && !(in_mul == 2 && mode == 0)) return ISTGCN_EINVAL;
  if (Mlog > 0 && (Mlog - 1) * out_mul + out_off >= Tout) return ISTGCN_EINVAL;
  if ((long long)Tin * V * Wp * 2 >= (1ll << 30)) return ISTGCN_EINVAL;   // one descriptor per sequence of the narrow tensor
  if (NM == 0 || Mlog == 0) return ISTGCN_OK;
  BoutParams P{};
  P.q = q; P.yb = yb; P.z = z; P.aux = aux; P.maux = maux; P.Wt = Wt; P.wt_ts = wt_ts; P.wt_rs = wt_rs; P.wt_cs = wt_cs;
  P.bt = bt; P.We = We; P.we_rs = we_rs; P.we_cs = we_cs; P.be = be; P.stats = stats; P.stats_rep = stats_rep < 1 ? 1 : stats_rep;
  P.NM = NM; P.Tin = Tin; P.Tout = Tout; P.Mlog = Mlog; P.V = V; P.C = C; P.Wn = Wn; P.Wp = Wp; P.ntaps = ntaps; P.off0 = off0;
  P.out_mul = out_mul; P.out_off = out_off;
  for (int j = 0; j < NTAP; ++j) P.tap_sel[j] = j < ntaps ? tap_sel[j] : 0;
  if (stats) istgcn_bn_tail_take(stats, &P.tail);
  if (dtype == 1) return bout_T<__bf16>(P, in_mul, mode, grid_cap, (hipStream_t)stream);
  return bout_T<_Float16>(P, in_mul, mode, grid_cap, (hipStream_t)stream);
}

extern "C" int istgcn_bneck_wgrad(const void* wide, const void* nrw, const float* pre, int pre_relu, float* dW, float* db,
                                  int wide_is_out, int db_wide, long long rows, int C, int Wp, int dtype, int grid_cap,
                                  float* ws, long long ws_floats, void* stream) {
  if (!wide || !nrw || !dW || !ws || rows < 0) return ISTGCN_EINVAL;
  if (!istgcn_bneck_ok(1, C, Wp, Wp, dtype)) return ISTGCN_EINVAL;
  if (rows * C * 2 >= (1ll << 32) - (1 << 20)) return ISTGCN_EINVAL;
  if (rows == 0) return ISTGCN_OK;
  BwgParams P{};
  P.wide = wide; P.nrw = nrw; P.pre = pre; P.dW = dW; P.db = db; P.ws = ws; P.rows = rows; P.C = C; P.Wp = Wp;
  P.pre_relu = pre_relu; P.wide_is_out = wide_is_out; P.db_wide = db_wide;
  P.ntiles = (int)((rows + 31) / 32);
  hipStream_t st = (hipStream_t)stream;
#define GO(TT)                                                          \
  switch (C) {                                                          \
    case 64: return bwg_launch<TT, 4>(P, grid_cap, ws_floats, st);      \
    case 128: return bwg_launch<TT, 8>(P, grid_cap, ws_floats, st);     \
    case 256: return bwg_launch<TT, 16>(P, grid_cap, ws_floats, st);    \
  }
  if (dtype == 1) { GO(__bf16) } else { GO(_Float16) }
#undef GO
  return ISTGCN_EINVAL;
}

extern "C" int istgcn_bneck_wgrad_taps(const void* dy, const void* q, float* dW, float* db, int NM, int Tin, int Tz, int V,
                                       int Wp, int ntaps, int off0, int in_mul, int dtype, int grid_cap, float* ws,
                                       long long ws_floats, void* stream) {
  if (!dy || !q || !dW || !ws) return ISTGCN_EINVAL;
  if (!istgcn_bneck_ok(V, 64, Wp, Wp, dtype)) return ISTGCN_EINVAL;
  if (ntaps < 1 || ntaps > NTAP || NM < 0 || Tin < 1 || Tz < 0 || (in_mul != 1 && in_mul != 2)) return ISTGCN_EINVAL;
  if ((long long)Tin * V * Wp * 2 >= (1ll << 30) || (long long)Tz * V * Wp * 2 >= (1ll << 30)) return ISTGCN_EINVAL;
  if (NM == 0 || Tz == 0) return ISTGCN_OK;
  BwtParams P{};
  P.dy = dy; P.q = q; P.dW = dW; P.db = db; P.ws = ws; P.NM = NM; P.Tin = Tin; P.Tz = Tz; P.V = V; P.Wp = Wp;
  P.ntaps = ntaps; P.off0 = off0;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == 1) return in_mul == 1 ? bwt_launch<__bf16, 1>(P, grid_cap, ws_floats, st) : bwt_launch<__bf16, 2>(P, grid_cap, ws_floats, st);
  return in_mul == 1 ? bwt_launch<_Float16, 1>(P, grid_cap, ws_floats, st) : bwt_launch<_Float16, 2>(P, grid_cap, ws_floats, st);
}

extern "C" int istgcn_bneck_bwd_in_ok(int C, int Wn, int Wp, int dtype) {
  return istgcn_bneck_ok(1, C, Wn, Wp, dtype) && (C == 64 || C == 128);
}

extern "C" int istgcn_bneck_bwd_in(const void* dres, const void* z, const float* abc, float p_drop, unsigned long long seed,
                                   const unsigned long long* seed_epoch, const void* yb, const float* W, long long w_rs,
                                   long long w_cs, void* dyb, float* dW, float* db, long long rows, int C, int Wn, int Wp,
                                   int dtype, int grid_cap, float* ws, long long ws_floats, void* stream) {
  if (!dres || !z || !abc || !yb || !W || !dyb || !dW || !ws || rows < 0 || p_drop < 0.f || p_drop > 1.f) return ISTGCN_EINVAL;
  if (!istgcn_bneck_bwd_in_ok(C, Wn, Wp, dtype)) return ISTGCN_EINVAL;
  if (rows * C * 2 >= (1ll << 32) - (1 << 20)) return ISTGCN_EINVAL;
  if (rows == 0) return ISTGCN_OK;
  BbiParams P{};
  P.dres = dres; P.z = z; P.abc = abc; P.nrw = yb; P.W = W; P.w_rs = w_rs; P.w_cs = w_cs; P.dyb = dyb; P.dW = dW; P.db = db;
  P.ws = ws; P.rows = rows; P.C = C; P.Wn = Wn; P.Wp = Wp; P.D = make_drop(p_drop, seed, seed_epoch);
  P.ntiles = (int)((rows + 31) / 32);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == 1) return C == 64 ? bbi_launch<__bf16, 4>(P, grid_cap, ws_floats, st) : bbi_launch<__bf16, 8>(P, grid_cap, ws_floats, st);
  return C == 64 ? bbi_launch<_Float16, 4>(P, grid_cap, ws_floats, st) : bbi_launch<_Float16, 8>(P, grid_cap, ws_floats, st);
}
