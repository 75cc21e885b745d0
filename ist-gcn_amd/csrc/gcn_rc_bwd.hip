// Graph-convolution unit, data gradient + adjacency gradient, register-chained (16-bit storage; scheme: gcn_rc.hip).
//
//   dxa_k[t,w,i] = sum_c W_k[c][i] dy[t,w,c]
//   dx[t,v,i]    = sum_k sum_w A_k[v][w] dxa_k[t,w,i]  (+ addend)
//   dA_k[v][w]  += sum_{t,i} x[t,v,i] dxa_k[t,w,i]                                  on the pattern
// = autograd of net/utils/tgcn.py:79-86 (and of the folded variants tgcn_multi3_fix_3A.py:86-89, inceptionv2_gcn.py:69-80).
//
// Round 2 kept the K dxa images in LDS between the contraction, the transposed aggregation and the dA product (that
// segment ran at exactly the LDS rate).  Here a wave owns (frame, ONE 32-channel tile of dx) and chains
// through registers:
//   H  = dy W_k      (rows w, lane = channel)   mfma(A = dy rows straight from HBM, B = W fragment from LDS)
//   H' = (dy W_k)^T  (rows = channel, lane = w) mfma(A = the SAME W fragment, B = the SAME dy registers): operands swapped
//   dx tile += A_k . H     A operand = A_k in the chained k order (per-lane constants), B = H converted in registers
//   dA_k^T  += H'^T . x    A operand = H' converted in registers (rows w, k = channel), B = x rows straight from HBM
// The second contraction costs MFMA time the HBM-bound kernel has to spare, and buys the transposed tile without an LDS
// image.  The W fragments are packed with the channel of MFMA row / column r permuted (bits 2 and 3 of r swapped) so that
// H' converted pairwise carries the channels of a k-step in memory order: its partner, x, is then a plain 16-byte load.
// dA_k^T lives in 16 accumulator registers per partition for the whole walk and is reduced once per workgroup.
//
// Round 5: where a wave computes BOTH chains, H' is no longer a second contraction (SO MFMAs per partition) but the
// exact transpose of the converted H: the register pairs of H that feed `dx += A_k . H` as a B operand, read as an A operand
// (rows = channel, k = joints in the chained order), times an identity fragment in the same k order -- two MFMAs, x * 1.0
// summed with zeros in fp32 and converted back: bit-exact 16-bit values.  Per (frame, tile, partition) SO + 6 MFMAs instead
// of 2 SO + 4 (256 output channels: 22 instead of 36), the second accumulator chain starts from registers instead of LDS
// fragments, and the 256-channel layers with the adjacency gradient fit one wave (no split roles, no round-2 kernel).
// Measured (bf16, NM = 128, per call, second contraction -> identity transpose; tools/ab_gbwd.sh): 64 -> 64 118 -> 113 us,
// 64 -> 128 152-167 -> 139-142, 128 -> 128 156 -> 145, 128 -> 256 216 (round-2 kernel) -> 190-197, 256 -> 256 215-220
// (round-2 kernel) -> 202: less than the MFMA count promises -- the walk is bound by the weight fragments' LDS reads
// (one 1 KB fragment per contraction MFMA) and the chains' latency, not by the matrix pipe.
#include "gcn_rc.hpp"
#include <type_traits>

extern "C" int istgcn_gcn_bwd_rc_layout(int Cin, int Cout, int K, int dtype);

// cache policy (raw buffer `aux`: 2 = nt, streaming) of the loads that read a tensor exactly once: x (the block input, from
// the forward pass -- cold) and the residual addend.  Experiment builds override.
#ifndef RCB_X_AUX
#define RCB_X_AUX 0
#endif
#ifndef RCB_ADD_AUX
#define RCB_ADD_AUX 0
#endif

namespace {

struct RcBwdParams {
  const void* dy; const void* x; const float* A; const float* pat; const void* Wq; const void* addend; void* dx; float* dA;
  const unsigned char* addm;     // ADD == 2: the ReLU byte mask of the addend (addend := addend * [bit]; one byte per 8 channels)
  int NM, T, V, Cin, Cout;
  int nfw, step_n, step_t, gy;
};

constexpr int BIMG_RS = 36;                   // dwords per pair-row of a wave's 32-channel output image
constexpr int BIMG_BYTES = 16 * BIMG_RS * 4;

__device__ static inline int perm23(int c) { return (c & ~12) | ((c & 4) << 1) | ((c & 8) >> 1); }

// SO = Cout / 16 (k-steps of the contraction over output channels), NCT = 32-channel tiles of dx per workgroup slice,
// DA = with the adjacency gradient; ADD = 0: no addend, 1: dx += addend, 2: dx += addend * [ReLU mask bit] -- the st_gcn block's
// identity-residual gradient dout * [out > 0] taken from dout and the forward's byte mask (net/st_gcnold.py:181-182,201-203)
// instead of a dres tensor written for it.
// (HIP: the second launch bound is WAVES PER SIMD -- 2 = the 8 waves of one workgroup per CU, 256 registers each)
template <typename T, int SO, int K, int NCT, bool DA, int ADD, int CN>
__global__ __launch_bounds__(RC_NTH, 2) void gcn_rc_bwd_kernel(const RcBwdParams P) {
  // CN != 0: the models' first layer (CN = 3 input channels, net/st_gcnold.py:44): one zero-padded 32-channel tile per frame,
  // x rows read with 16-bit loads.  The adjacency gradient always; dx (P.dx != NULL: the gradient of data_bn's weight and
  // bias flows through it, st_gcnold.py:74-80) as 16-bit stores of the three valid columns, else no dx chain at all.
  static_assert(CN == 0 || (NCT == 1 && DA && ADD == 0 && CN <= 4), "narrow input: dA only");
  using E = Elem<T>;
  typedef typename E::frag frag_t;
  constexpr int COUT = 16 * SO;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int WVEC = NCT * K * SO * 64;                    // 16-byte vectors of the weight slice
  u32x4* wl = reinterpret_cast<u32x4*>(smem);
  uint32_t* img_all = reinterpret_cast<uint32_t*>(smem + (size_t)WVEC * 16);
  u32x4* adl = reinterpret_cast<u32x4*>(smem + (size_t)WVEC * 16 + 8 * BIMG_BYTES);      // [K][2][64] fragments of A_k
  // after the walk the whole buffer is reused for the dA reduction: [8 waves][K][32][32] floats

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  // Every wave runs both chains for its (frame, tile).  At >= 128 output channels there are no registers for a second dy
  // set: dy is reloaded in place right after its last use (PF2 = false); at 256 x likewise.  (Round 3's split-role form for
  // 256 channels -- dx waves / dA waves -- measured slower than the round-2 kernel and went with it in round 5.)
  constexpr int NWR = 8;
  const int wr = wave8;
  const int itl = wr % NCT, fwl = wr / NCT;
  const int b = blockIdx.x;
  const int slice = (b >> 3) % P.gy;
  const int grp = (b / (8 * P.gy)) * 8 + (b & 7);
  const int V = P.V, Cin = P.Cin;
  const int c = lane & 31, h = lane >> 5;
  const int it = slice * NCT + itl;                           // this wave's 32-channel tile of dx / of x

  // ---- setup ----
  {
    const u32x4* wg = reinterpret_cast<const u32x4*>(P.Wq) + (size_t)slice * WVEC;
    constexpr int NWI = (WVEC + RC_NTH - 1) / RC_NTH;
    u32x4 wv[NWI];
#pragma unroll
    for (int i = 0; i < NWI; ++i) wv[i] = wg[min(tid + i * RC_NTH, WVEC - 1)];
#pragma unroll
    for (int i = 0; i < NWI; ++i) if (tid + i * RC_NTH < WVEC) wl[tid + i * RC_NTH] = wv[i];
  }
  // A operand of dx = A_k . H: lane (v, h), k-step s, element j holds A_k[v][w = 16s + 8(j>>2) + 4h + (j&3)]; kept in LDS
  // (6 KB) and read where it is used (as 24 per-lane constants next to the 48 accumulator registers of dA^T it spilled)
  for (int idx = tid; idx < K * 2 * 64; idx += RC_NTH) {
    const int ln = idx & 63, s = (idx >> 6) & 1, k = idx >> 7;
    const int v = ln & 31, hh = ln >> 5;
    frag_t f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int w = 16 * s + 8 * (j >> 2) + 4 * hh + (j & 3);
      f[j] = E::from_f((v < V && w < V) ? P.A[(k * V + min(v, V - 1)) * V + min(w, V - 1)] : 0.f);
    }
    adl[idx] = __builtin_bit_cast(u32x4, f);
  }
  __syncthreads();
  // identity in the chained k order: as a B operand, lane (n, h), k-step s, element j is [joint 16s + 8(j>>2) + 4h + (j&3) == n].
  // Per-lane constants: in 8 registers, or (IDL: 256 output channels, where every register counts) in LDS behind the A fragments
  constexpr bool IDL = SO >= 16;
  u32x4 idf[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    frag_t f;
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = E::from_f((16 * s + 8 * (j >> 2) + 4 * h + (j & 3)) == c ? 1.f : 0.f);
    idf[s] = __builtin_bit_cast(u32x4, f);
    if constexpr (IDL) adl[(2 * K + s) * 64 + lane] = idf[s];          // (each lane reads back its own entry: no barrier needed)
  }

  const T* dyg = reinterpret_cast<const T*>(P.dy);
  const T* xg = reinterpret_cast<const T*>(P.x);
  T* dxg = reinterpret_cast<T*>(P.dx);
  const T* addg = reinterpret_cast<const T*>(P.addend);
  const unsigned dyfrm_b = (unsigned)(V * COUT) * 2u, xfrm_b = (unsigned)(V * Cin) * 2u;
  const u32x4* wlane = wl + (size_t)itl * (K * SO * 64) + lane;            // fragment (k, s) at + (k*SO + s)*64
  const unsigned dyoff = (unsigned)(c * COUT + 8 * h) * 2u;                // this lane's row vector of a dy frame (bytes)
  // ... of an x frame, this wave's channel tile (narrow rows: lane half 0 holds the whole row, half 1 reads zeros)
  const unsigned xoff = CN ? (h == 0 ? (unsigned)(c * CN) * 2u : 0x7ffffff0u) : (unsigned)(c * Cin + 32 * it + 8 * h) * 2u;
  const size_t dy_frm = (size_t)V * COUT, x_frm = (size_t)V * Cin;
  uint32_t* img = img_all + wave8 * (BIMG_BYTES / 4);
  const int pc = perm23(c);                                   // channel (within the tile) of MFMA column c
  const int rp = lane >> 2, chunk = lane & 3;                 // image slot this lane copies out: pair-row, channel vector
  const unsigned orow_b = (unsigned)Cin * 2u, ooff = (unsigned)(32 * it + 8 * chunk) * 2u;

  auto load_dy = [&](int fn, int ft, u32x4 (&f)[SO]) __attribute__((always_inline)) {
    const rsrc_t r = make_rsrc(dyg + ((size_t)fn * P.T + ft) * dy_frm, dyfrm_b);
#pragma unroll
    for (int s = 0; s < SO; ++s) f[s] = __builtin_amdgcn_raw_buffer_load_b128(r, dyoff + 32u * s, 0, 0);
  };
  auto load_x = [&](int fn, int ft, u32x4 (&f)[2]) __attribute__((always_inline)) {
    const rsrc_t r = make_rsrc(xg + ((size_t)fn * P.T + ft) * x_frm, xfrm_b);
    if constexpr (CN) {
      uint32_t e[4] = {0u, 0u, 0u, 0u};
#pragma unroll
      for (int j = 0; j < CN; ++j) e[j] = (uint32_t)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(r, xoff + 2u * j, 0, 0);
      f[0] = u32x4{e[0] | (e[1] << 16), e[2] | (e[3] << 16), 0u, 0u};
      f[1] = u32x4{0u, 0u, 0u, 0u};
    } else {
      f[0] = __builtin_amdgcn_raw_buffer_load_b128(r, xoff, 0, RCB_X_AUX);
      f[1] = __builtin_amdgcn_raw_buffer_load_b128(r, xoff + 32u, 0, RCB_X_AUX);
    }
  };

  f32x16 Z[K];
#pragma unroll
  for (int k = 0; k < K; ++k)
#pragma unroll
    for (int i = 0; i < 16; ++i) Z[k][i] = 0.f;

  // One frame of one tile.  DX / DAF select the chains; with RELOAD the dy registers are refilled with frame (n2, t2)
  // right after their last use.
  auto frame = [&](auto dx_tag, auto da_tag, auto reload_tag, int fn, int ft, u32x4 (&df)[SO], u32x4 (&xf)[2], int n2,
                   int t2) __attribute__((always_inline)) {
    constexpr bool DX = decltype(dx_tag)::value, DAF = decltype(da_tag)::value, RELOAD = decltype(reload_tag)::value;
    constexpr bool TRI = DX && DAF;                           // H' from the converted H (both chains on this wave)
    f32x16 Y;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      f32x16 H, HT;
#pragma unroll
      for (int i = 0; i < 16; ++i) { H[i] = 0.f; HT[i] = 0.f; }
#pragma unroll
      for (int s = 0; s < SO; ++s) {
        const u32x4 wv = wlane[(k * SO + s) * 64];
        if constexpr (DX) mma_kgroup(H, __builtin_bit_cast(frag_t, df[s]), __builtin_bit_cast(frag_t, wv));
        if constexpr (DAF && !TRI) mma_kgroup(HT, __builtin_bit_cast(frag_t, wv), __builtin_bit_cast(frag_t, df[s]));
      }
      if constexpr (RELOAD) {
        if (k == K - 1) load_dy(n2, t2, df);
      }
      if constexpr (DX) {
        u32x4 hb[2];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int q = 0; q < 4; ++q) hb[s][q] = pack2<T>(H[8 * s + 2 * q], H[8 * s + 2 * q + 1]);
        if (k == 0) {
#pragma unroll
          for (int i = 0; i < 16; ++i) Y[i] = 0.f;
        }
        mma_kgroup(Y, __builtin_bit_cast(frag_t, adl[(2 * k) * 64 + lane]), __builtin_bit_cast(frag_t, hb[0]));
        mma_kgroup(Y, __builtin_bit_cast(frag_t, adl[(2 * k + 1) * 64 + lane]), __builtin_bit_cast(frag_t, hb[1]));
        if constexpr (TRI) {
          // H'[i][w] = sum over joints of (converted H as the A operand: rows = channel, k = joint) x identity
          if constexpr (IDL) {
            mma_kgroup(HT, __builtin_bit_cast(frag_t, hb[0]), __builtin_bit_cast(frag_t, adl[(2 * K) * 64 + lane]));
            mma_kgroup(HT, __builtin_bit_cast(frag_t, hb[1]), __builtin_bit_cast(frag_t, adl[(2 * K + 1) * 64 + lane]));
          } else {
            mma_kgroup(HT, __builtin_bit_cast(frag_t, hb[0]), __builtin_bit_cast(frag_t, idf[0]));
            mma_kgroup(HT, __builtin_bit_cast(frag_t, hb[1]), __builtin_bit_cast(frag_t, idf[1]));
          }
        }
      }
      if constexpr (DAF) {
        u32x4 ht[2];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int q = 0; q < 4; ++q) ht[s][q] = pack2<T>(HT[8 * s + 2 * q], HT[8 * s + 2 * q + 1]);
        mma_kgroup(Z[k], __builtin_bit_cast(frag_t, ht[0]), __builtin_bit_cast(frag_t, xf[0]));
        if constexpr (CN == 0) mma_kgroup(Z[k], __builtin_bit_cast(frag_t, ht[1]), __builtin_bit_cast(frag_t, xf[1]));
      }
    }
    if constexpr (RELOAD && TRI && SO >= 16) load_x(n2, t2, xf);      // (one x set: refilled right after its last use, like dy)
    if constexpr (DX && CN != 0) {
      // narrow input (the 3-channel first layer, whose dx is the gradient of data_bn): rows of 2 * CN bytes -- 16-bit stores
      // straight from the accumulator tile (lane = channel perm23(c) = c for c < 4, registers = rows); lanes of the zero
      // padding and rows >= V fall outside the frame's descriptor
      const size_t fo = ((size_t)fn * P.T + ft) * x_frm;
      const rsrc_t ro = make_rsrc(dxg + fo, xfrm_b);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
        const unsigned off = c < CN ? (unsigned)(row * CN + c) * 2u : 0x7ffffff0u;
        const uint32_t pk = pack2<T>(Y[i], 0.f);
        __builtin_amdgcn_raw_buffer_store_b16((unsigned short)(pk & 0xffffu), ro, off, 0, 0);
      }
    } else if constexpr (DX) {
      // dx tile -> pair-row image (column c of the tile is channel perm23(c)) -> 16-byte row vectors -> HBM
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int p = (q & 1) + 4 * (q >> 1) + 2 * h;
        img[p * BIMG_RS + pc] = pack2<T>(Y[2 * q], Y[2 * q + 1]);
      }
      const size_t fo = ((size_t)fn * P.T + ft) * x_frm;
      const rsrc_t ro = make_rsrc(dxg + fo, xfrm_b);
      u32x4 av[2];
      uint32_t mb[2] = {0xffu, 0xffu};
      if constexpr (ADD != 0) {
        const rsrc_t ra = make_rsrc(addg + fo, xfrm_b);
        av[0] = __builtin_amdgcn_raw_buffer_load_b128(ra, ooff + (unsigned)(2 * rp) * orow_b, 0, RCB_ADD_AUX);
        av[1] = __builtin_amdgcn_raw_buffer_load_b128(ra, ooff + (unsigned)(2 * rp + 1) * orow_b, 0, RCB_ADD_AUX);
        if constexpr (ADD == 2) {
          // one mask byte per 16-byte vector: frame f starts at byte f * V * Cin / 8, row w at w * Cin / 8 (rows >= V: outside
          // the descriptor -> 0 -> nothing added to rows whose stores are dropped anyway)
          const rsrc_t rm = make_rsrc(P.addm + (fo >> 3), (unsigned)(V * Cin) >> 3);
          const unsigned mrow = (unsigned)Cin >> 3, moff = (unsigned)(4 * it + chunk);
          mb[0] = (uint32_t)(unsigned char)__builtin_amdgcn_raw_buffer_load_b8(rm, moff + (unsigned)(2 * rp) * mrow, 0, RCB_ADD_AUX);
          mb[1] = (uint32_t)(unsigned char)__builtin_amdgcn_raw_buffer_load_b8(rm, moff + (unsigned)(2 * rp + 1) * mrow, 0, RCB_ADD_AUX);
        }
      }
      const u32x4 u0 = *reinterpret_cast<const u32x4*>(img + rp * BIMG_RS + 8 * chunk);
      const u32x4 u1 = *reinterpret_cast<const u32x4*>(img + rp * BIMG_RS + 8 * chunk + 4);
      u32x4 ev, od;
      ev[0] = __builtin_amdgcn_perm(u0[1], u0[0], 0x05040100u); od[0] = __builtin_amdgcn_perm(u0[1], u0[0], 0x07060302u);
      ev[1] = __builtin_amdgcn_perm(u0[3], u0[2], 0x05040100u); od[1] = __builtin_amdgcn_perm(u0[3], u0[2], 0x07060302u);
      ev[2] = __builtin_amdgcn_perm(u1[1], u1[0], 0x05040100u); od[2] = __builtin_amdgcn_perm(u1[1], u1[0], 0x07060302u);
      ev[3] = __builtin_amdgcn_perm(u1[3], u1[2], 0x05040100u); od[3] = __builtin_amdgcn_perm(u1[3], u1[2], 0x07060302u);
      if constexpr (ADD == 2) {
        // keep the 16-bit halves whose mask bit is set (element j of the vector = bit j)
#pragma unroll
        for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const uint32_t lo = (uint32_t)(-(int32_t)((mb[m2] >> (2 * q)) & 1u)) & 0x0000ffffu;
            const uint32_t hi = (uint32_t)(-(int32_t)((mb[m2] >> (2 * q + 1)) & 1u)) & 0xffff0000u;
            av[m2][q] &= (lo | hi);
          }
      }
      if constexpr (ADD != 0) {
        const frag_t a0 = __builtin_bit_cast(frag_t, av[0]), a1 = __builtin_bit_cast(frag_t, av[1]);
        frag_t o0 = __builtin_bit_cast(frag_t, ev), o1 = __builtin_bit_cast(frag_t, od);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          o0[j] = E::from_f(E::to_f(o0[j]) + E::to_f(a0[j]));
          o1[j] = E::from_f(E::to_f(o1[j]) + E::to_f(a1[j]));
        }
        ev = __builtin_bit_cast(u32x4, o0);
        od = __builtin_bit_cast(u32x4, o1);
      }
      __builtin_amdgcn_raw_buffer_store_b128(ev, ro, ooff + (unsigned)(2 * rp) * orow_b, 0, 0);
      __builtin_amdgcn_raw_buffer_store_b128(od, ro, ooff + (unsigned)(2 * rp + 1) * orow_b, 0, 0);
    }
  };

  // ---- the walk over this worker's frames ----
  const int fw = grp * (NWR / NCT) + fwl;                     // per role: groups * NWR / NCT frame workers
  int n = fw / P.T, t = fw - n * P.T;
  auto next = [&](int& nn, int& tt) __attribute__((always_inline)) {
    nn += P.step_n;
    tt += P.step_t;
    if (tt >= P.T) { tt -= P.T; ++nn; }
  };
  auto walk = [&](auto dx_tag, auto da_tag, auto pf2_tag) __attribute__((always_inline)) {
    constexpr bool DAF = decltype(da_tag)::value, PF = decltype(pf2_tag)::value;
    typedef std::integral_constant<bool, !PF> reload_t;
    if (n >= P.NM) return;
    u32x4 da[SO], xa[2], xb[2];
    load_dy(n, t, da);
    if constexpr (DAF) load_x(n, t, xa);
    if constexpr (PF) {
      u32x4 db[SO];
      for (;;) {
        int n2 = n, t2 = t;
        next(n2, t2);
        const bool more = n2 < P.NM;
        load_dy(more ? n2 : n, more ? t2 : t, db);
        if constexpr (DAF) load_x(more ? n2 : n, more ? t2 : t, xb);
        __builtin_amdgcn_sched_barrier(0);
        frame(dx_tag, da_tag, reload_t{}, n, t, da, xa, n2, t2);
        if (!more) break;
        n = n2; t = t2;
        next(n2, t2);
        const bool more2 = n2 < P.NM;
        load_dy(more2 ? n2 : n, more2 ? t2 : t, da);
        if constexpr (DAF) load_x(more2 ? n2 : n, more2 ? t2 : t, xa);
        __builtin_amdgcn_sched_barrier(0);
        frame(dx_tag, da_tag, reload_t{}, n, t, db, xb, n2, t2);
        if (!more2) break;
        n = n2; t = t2;
      }
    } else if constexpr (DAF && decltype(dx_tag)::value && SO >= 16) {
      // 256 output channels, both chains: ONE set of dy and ONE of x, each refilled by frame() right after its last use
      for (;;) {
        int n2 = n, t2 = t;
        next(n2, t2);
        const bool more = n2 < P.NM;
        frame(dx_tag, da_tag, reload_t{}, n, t, da, xa, more ? n2 : n, more ? t2 : t);
        if (!more) break;
        n = n2; t = t2;
      }
    } else {
      for (;;) {
        int n2 = n, t2 = t;
        next(n2, t2);
        const bool more = n2 < P.NM;
        if constexpr (DAF) load_x(more ? n2 : n, more ? t2 : t, xb);
        __builtin_amdgcn_sched_barrier(0);
        frame(dx_tag, da_tag, reload_t{}, n, t, da, xa, more ? n2 : n, more ? t2 : t);
        if (!more) break;
        n = n2; t = t2;
        next(n2, t2);
        const bool more2 = n2 < P.NM;
        if constexpr (DAF) load_x(more2 ? n2 : n, more2 ? t2 : t, xa);
        __builtin_amdgcn_sched_barrier(0);
        frame(dx_tag, da_tag, reload_t{}, n, t, da, xb, more2 ? n2 : n, more2 ? t2 : t);
        if (!more2) break;
        n = n2; t = t2;
      }
    }
  };
  typedef std::true_type yes;
  typedef std::false_type no;
  if constexpr (CN != 0) {
    if (P.dx) walk(yes{}, yes{}, yes{});                      // (round 5: with dx -- the gradient data_bn needs)
    else walk(no{}, yes{}, yes{});
  }
  else if constexpr (!DA) walk(yes{}, no{}, yes{});
  else walk(yes{}, yes{}, std::integral_constant<bool, (SO <= 4)>{});

  // ---- adjacency gradient: Z[k] (rows w in registers, lane = v) of the waves -> LDS -> one atomic per pattern entry ----
  if constexpr (DA) {
    constexpr int NZ = 8;                                     // waves that hold partial sums
    __syncthreads();                                          // every wave is done with the weights and its image
    float* red = reinterpret_cast<float*>(smem);              // [NZ][K][32][32]
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int w = (i & 3) + 8 * (i >> 2) + 4 * h;
        red[((wr * K + k) * 32 + w) * 32 + c] = Z[k][i];
      }
    __syncthreads();
    const float* pat = P.pat ? P.pat : P.A;
    for (int e = tid; e < K * V * V; e += RC_NTH) {
      if (pat[e] != 0.f) {
        const int k = e / (V * V), r = e - k * V * V, v = r / V, w = r - v * V;
        float sum = 0.f;
#pragma unroll
        for (int wv = 0; wv < NZ; ++wv) sum += red[((wv * K + k) * 32 + w) * 32 + v];
        atomicAdd(P.dA + e, sum);
      }
    }
  }
}

template <typename T, int SO, int K, int NCT, bool DA, int ADD, int CN = 0>
int rc_bwd_launch(RcBwdParams P, int grid_cap, hipStream_t stream) {
  auto kfn = gcn_rc_bwd_kernel<T, SO, K, NCT, DA, ADD, CN>;
  size_t lds = (size_t)NCT * K * SO * 64 * 16 + 8 * BIMG_BYTES + (size_t)(K + 1) * 2 * 64 * 16;      // (+ the identity fragments)
  if (DA && (size_t)8 * K * 32 * 32 * 4 > lds) lds = (size_t)8 * K * 32 * 32 * 4;
  if (lds > 160 * 1024) return ISTGCN_EINVAL;
  static std::atomic<unsigned long long> optin{0};
  if (int ea = istgcn_lds_optin((const void*)kfn, optin)) return ea;
  int res = grid_cap > 0 ? grid_cap : istgcn_resident_blocks((const void*)kfn, RC_NTH, lds);
  P.gy = CN ? 1 : P.Cin / (32 * NCT);
  int G = res / P.gy / 8 * 8;
  if (G < 8) G = 8;
  const long long frames = (long long)P.NM * P.T;
  const int fwpg = 8 / NCT;                     // frame workers per group
  while (G > 8 && (long long)(G - 8) * fwpg >= frames) G -= 8;
  P.nfw = G * fwpg;
  P.step_n = P.nfw / P.T;
  P.step_t = P.nfw % P.T;
  ISTGCN_LAUNCH(kfn, dim3(G * P.gy), dim3(RC_NTH), lds, stream, P);
  ISTGCN_CHECK_LAUNCH();
  return ISTGCN_OK;
}

template <typename T, int SO, int K, int NCT>
int rc_bwd_flags(const RcBwdParams& P, int grid_cap, hipStream_t stream) {
  if constexpr ((size_t)NCT * K * SO * 64 * 16 > 100 * 1024) return ISTGCN_EINVAL;
  else {
    const bool da = P.dA != nullptr;
    const int add = P.addend == nullptr ? 0 : (P.addm ? 2 : 1);
    if (da && add == 2) return rc_bwd_launch<T, SO, K, NCT, true, 2>(P, grid_cap, stream);
    if (da && add == 1) return rc_bwd_launch<T, SO, K, NCT, true, 1>(P, grid_cap, stream);
    if (da) return rc_bwd_launch<T, SO, K, NCT, true, 0>(P, grid_cap, stream);
    if (add == 2) return rc_bwd_launch<T, SO, K, NCT, false, 2>(P, grid_cap, stream);
    if (add == 1) return rc_bwd_launch<T, SO, K, NCT, false, 1>(P, grid_cap, stream);
    return rc_bwd_launch<T, SO, K, NCT, false, 0>(P, grid_cap, stream);
  }
}

template <typename T, int SO, int K>
int rc_bwd_nct(const RcBwdParams& P, int grid_cap, hipStream_t stream) {
  // tiles of dx per workgroup: 4 where the slice's weights (K * Cout * 32 * NCT 16-bit elements) fit LDS, else 2
  constexpr bool four = (size_t)K * 16 * SO * 128 * 2 <= 100 * 1024;
  if constexpr (four) {
#ifdef ISTGCN_EXPERIMENT
    static const bool two = [] { const char* e = getenv("ISTGCN_RC_NCT"); return e && atoi(e) == 2; }();   // 2 = two tiles per slice everywhere
#else
    constexpr bool two = false;
#endif
    if (P.Cin % 128 == 0 && !two) return rc_bwd_flags<T, SO, K, 4>(P, grid_cap, stream);
  }
  return rc_bwd_flags<T, SO, K, 2>(P, grid_cap, stream);
}

template <typename T, int SO>
int rc_bwd_k(const RcBwdParams& P, int K, int grid_cap, hipStream_t stream) {
  switch (K) {
    case 1: return rc_bwd_nct<T, SO, 1>(P, grid_cap, stream);
    case 2: return rc_bwd_nct<T, SO, 2>(P, grid_cap, stream);
    case 3: return rc_bwd_nct<T, SO, 3>(P, grid_cap, stream);
    case 4: return rc_bwd_nct<T, SO, 4>(P, grid_cap, stream);
  }
  return ISTGCN_EINVAL;
}

template <typename T>
int rc_bwd_first_layer(const RcBwdParams& P, int K, int grid_cap, hipStream_t stream) {
  if (P.Cout != 64 || !P.dA || P.addend) return ISTGCN_EINVAL;
  switch (K) {
    case 1: return rc_bwd_launch<T, 4, 1, 1, true, 0, 3>(P, grid_cap, stream);
    case 2: return rc_bwd_launch<T, 4, 2, 1, true, 0, 3>(P, grid_cap, stream);
    case 3: return rc_bwd_launch<T, 4, 3, 1, true, 0, 3>(P, grid_cap, stream);
    case 4: return rc_bwd_launch<T, 4, 4, 1, true, 0, 3>(P, grid_cap, stream);
  }
  return ISTGCN_EINVAL;
}

template <typename T>
int rc_bwd_T(const RcBwdParams& P, int K, int grid_cap, hipStream_t stream) {
  if (P.Cin == 3) return rc_bwd_first_layer<T>(P, K, grid_cap, stream);
  switch (P.Cout) {
    case 64: return rc_bwd_k<T, 4>(P, K, grid_cap, stream);
    case 128: return rc_bwd_k<T, 8>(P, K, grid_cap, stream);
    case 256: return rc_bwd_k<T, 16>(P, K, grid_cap, stream);
  }
  return ISTGCN_EINVAL;
}

}  // namespace

// Do the packed weights of istgcn_gcn_bwd_data carry the register-chained section (istgcn.h)?
extern "C" int istgcn_gcn_bwd_rc_layout(int Cin, int Cout, int K, int dtype) {
  if (dtype != 1 && dtype != 2) return 0;
  if (Cout != 64 && Cout != 128 && Cout != 256) return 0;
  if (Cin == 3 && Cout == 64 && K >= 1 && K <= 4) return 1;          // first layer: adjacency gradient only (dx == NULL)
  if (Cin < 64 || Cin % 64 != 0 || K < 1 || K > 4) return 0;
  if ((size_t)K * Cout * 64 * 2 > 100 * 1024) return 0;
  return 1;
}

extern "C" int istgcn_gcn_bwd_data_rc(const void* dy, const void* x, const float* A, const float* pattern, const void* Wq,
                                      const void* addend, const unsigned char* addend_mask, void* dx, float* dA, int NM, int T,
                                      int V, int Cin, int Cout, int K, int dtype, int grid_cap, void* stream) {
  if (!istgcn_gcn_bwd_rc_layout(Cin, Cout, K, dtype) || V > 32 || (addend_mask && (!addend || Cin % 8 != 0))) return ISTGCN_EINVAL;
  RcBwdParams P{};
  P.dy = dy; P.x = x; P.A = A; P.pat = pattern; P.Wq = Wq; P.addend = addend; P.addm = addend_mask; P.dx = dx; P.dA = dA;
  P.NM = NM; P.T = T; P.V = V; P.Cin = Cin; P.Cout = Cout;
  if (dtype == 1) return rc_bwd_T<__bf16>(P, K, grid_cap, (hipStream_t)stream);
  return rc_bwd_T<_Float16>(P, K, grid_cap, (hipStream_t)stream);
}
