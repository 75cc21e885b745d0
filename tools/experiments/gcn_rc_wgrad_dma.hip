// EXPERIMENT, NOT BUILT INTO THE LIBRARY (round 5).  csrc/gcn_rc_wgrad.hip with a second kernel, gcn_rc_wgrad_dma_kernel: frames
// staged by LDS-DMA (global_load_lds_dwordx4 through inline asm, three buffers, two batches in flight across a raw s_barrier
// with counted s_waitcnt vmcnt), two XA tiles (partition k + 1 aggregated while k matures), S balanced over the waves -- and the
// RCW_ABL ablation switches.  Measured: correct (dW to the bf16 rounding of the old kernel on 24 shapes), and NOT faster --
// 113-121 us against 112-119 at 128 -> 128 channels, 187-202 against 184-201 at 256 -> 256 (profiles/r05_gcn_wgrad_ablation.txt,
// which also says why: the staging floor and the matrix work add up instead of overlapping).  Kept as the starting point
// for anyone who tries LDS-DMA here again, with two things learnt on the way:
//   * after __builtin_amdgcn_global_load_lds hipcc (ROCm 7.2) puts s_waitcnt vmcnt(0) in front of the next
//     ds_read_b64_tr_b16 (and in front of plain LDS reads once the DMA sits under a lane predicate): a DMA that is to stay in
//     flight across LDS reads has to be inline asm (glds16 below), with every wait written by hand;
//   * __builtin_amdgcn_readfirstlane returns int: reassembling a 64-bit pointer from two halves sign-extends the low half
//     unless it is cast to unsigned first -- a wild pointer (memory access fault) whenever bit 31 of the address is set.
// To try it: copy over csrc/gcn_rc_wgrad.hip, tools/build_variant.sh <name> gcn_rc_wgrad.hip [-DRCW_ABL=<bits>], tools/ab_gwg.sh.
// Graph-convolution unit, weight gradient, register-chained (16-bit storage; scheme: gcn_rc.hip).
//
//   xa_k[t,w,i] = sum_v A_k[v][w] x[t,v,i]
//   dW[k][c][i] += sum_{n,t,w} dy[t,w,c] xa_k[t,w,i]            (= Conv2d weight gradient, autograd of net/utils/tgcn.py:79-86)
//   S[w][c]     += sum_{n,t} dy[t,w,c]                           (gradient of the bias term)
//
// Both contractions run over joints, the axis that is NOT contiguous in memory, so both operands come out of LDS
// through ds_read_b64_tr_b16 as in round 2 -- but the aggregated tile xa_k no longer makes an LDS round trip between
// the two MFMA stages (round 2: all eight waves aggregate into swizzled images, barrier, four waves contract; 30 % of
// that kernel's LDS cycles were bank conflicts).  A wave owns one (32 output channels, 32 input channels) pair of the
// workgroup's channel block for ALL K partitions and walks frames (joints padded to 32 rows):
//   XA_k = A_k^T . x_frame      mfma(A = A_k^T per-lane constants, B = x^T fragments from the frame's LDS image)
//                               -> rows w in the 16 registers, lane = input channel
//   dW_k += dy_frame^T . XA_k   mfma(A = dy^T fragments from LDS in the chained k order, B = XA_k converted in registers)
//   S^T  += dy_frame^T . I      one more product against a constant permuted identity (waves of input tile 0)
// Frames are staged by all eight waves: global -> registers (in flight during a whole batch of four frames) -> LDS
// image with 16-byte chunks XOR-swizzled so that the transposed reads are conflict-free; one barrier per batch.
// The K accumulator tiles live in registers for the whole walk; flush = per-workgroup partial sums to the workspace +
// the reduce kernel of tconv_wgrad.hip (or atomics without a workspace).
//
// Round 5 (gcn_rc_wgrad_dma_kernel; the kernel above stays for the 3-channel first layer, whose 6-byte rows are no DMA
// pieces): the ablations of profiles/r05_gcn_wgrad_ablation.txt put a quarter of the kernel into the staging (global ->
// registers -> LDS, one batch ahead -- all the registers leave room for) and the rest into a compute phase that ran at half
// the matrix rate because each partition's chain (2 aggregation MFMAs -> 64 cycles until the result can be read ->
// convert -> 4 contraction MFMAs) was serial within a wave, with no register left for a second XA tile.  Now
//   * frames arrive by LDS-DMA (global_load_lds_dwordx4: a wave instruction writes 1 KB of the image = 8 rows of 128 bytes
//     or 4 rows of 256), the XOR swizzle of the images applied on the SOURCE side (lane -> which 16-byte chunk of its row
//     it fetches); pad rows are never written and stay zero; three buffers, TWO batches in flight across the batch
//     barrier (counted s_waitcnt vmcnt + raw s_barrier: __syncthreads() would drain the DMA);
//   * the 36 registers the staging set and its offsets held are two XA tiles: partition k + 1 is aggregated while partition
//     k's result matures and is converted;
//   * the two waves that share output tiles take one tile's S product each.
// Same frames per wave, same order of accumulation: results are bit-identical to the kernel above.
#include "gcn_rc.hpp"
#include <cstdlib>

extern "C" int istgcn_wgrad_reduce(const float* ws, long long slice, int nsl, float* d0, int n0, float* d1, int n1, void* stream);

// cache policy (raw buffer `aux`: 2 = nt) of the staging loads; experiment builds override
#ifndef RCW_X_AUX
#define RCW_X_AUX 0
#endif
#ifndef RCW_DY_AUX
#define RCW_DY_AUX 0
#endif

namespace {

struct RcWgParams {
  const void* dy; const void* x; const float* A; float* dW; float* S; float* ws;
  long long ws_slice;
  int NM, T, V, Cin, Cout;
  int G, gy, nib;        // groups (grid-stride over batches), channel blocks per group, input-channel blocks
};

constexpr int WG_FBT = 4;                 // frames per batch (all frame groups together)

// CT = 32-channel tiles of dy per workgroup block (2 or 4); the block's input side is always 64 channels (two tiles).
// OT = output tiles a wave owns next to ONE input tile (1 or 2): the aggregated tile XA_k is computed once per wave and
// frame, so with OT = 2 a frame costs 2K + 2 * 2K + 2 MFMAs for two (output, input) pairs instead of 2 * (4K + 1) -- a
// quarter of the matrix work of this MFMA-issue-bound kernel -- at 2 * (K + 1) accumulator tiles per wave.
// Waves: (ct group, input tile) x FG frame groups, FG = 8 / (2 CT / OT).
// CN != 0: the models' 3-channel first layer -- x rows are 2*CN bytes, read with 16-bit loads into channel vector 0 of
// the (otherwise zero) 64-channel image.
template <typename T, int K, int CT, int CN, int OT>
__global__ __launch_bounds__(RC_NTH, 2) void gcn_rc_wgrad_kernel(const RcWgParams P) {
  using E = Elem<T>;
  typedef typename E::frag frag_t;
  constexpr int CTG = CT / OT;                              // wave groups along the output tiles
  constexpr int FG = 8 / (2 * CTG);
  constexpr int XROW = 64, DROW = 32 * CT;                  // elements per image row
  constexpr int XFRM = 32 * XROW, DFRM = 32 * DROW;         // elements per frame image (32 rows)
  constexpr int BUF = WG_FBT * (XFRM + DFRM);               // elements per buffer
  constexpr int QX = 8, QD = 4 * CT;                        // 16-byte chunks per image row
  constexpr int NITX = WG_FBT * 32 * QX / RC_NTH, NITD = WG_FBT * 32 * QD / RC_NTH;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T* lds = reinterpret_cast<T*>(smem);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int itl = wave8 & 1, ctg = (wave8 >> 1) % CTG, fg = (wave8 >> 1) / CTG;
  const int ct = ctg * OT;                                  // first output tile of this wave
  const int b = blockIdx.x;
  const int blk = (b >> 3) % P.gy;                          // channel block of this workgroup
  const int grp = (b / (8 * P.gy)) * 8 + (b & 7);
  const int ib = blk % P.nib, cb = blk / P.nib;             // input-channel block (64 wide), output-channel block (32 CT wide)
  const int V = P.V, Cin = P.Cin, Cout = P.Cout;
  const int c = lane & 31, h = lane >> 5;

  // ---- setup: zero both buffers (pad rows stay zero), adjacency (through an LDS copy: built from global memory the
  //      compiler waits for each of the 48 scalar loads in turn) -> per-lane fragments, and the permuted identity ----
  for (int i = tid; i < 2 * BUF / 8; i += RC_NTH) reinterpret_cast<u32x4*>(smem)[i] = u32x4{0u, 0u, 0u, 0u};
  float* Asc = reinterpret_cast<float*>(smem + (size_t)2 * BUF * 2);      // [K][V][V] behind the buffers
  for (int i = tid; i < K * V * V; i += RC_NTH) Asc[i] = P.A[i];
  __syncthreads();
  // A operand of XA_k = A_k^T x: lane (w = c, h), k-step s, element j holds A_k[v = 16s + 8h + j][w]
  u32x4 At[K][2];
#pragma unroll
  for (int k = 0; k < K; ++k) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      frag_t f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int v = 16 * s + 8 * h + j;
        f[j] = E::from_f((c < V && v < V) ? Asc[(k * V + v) * V + c] : 0.f);
      }
      At[k][s] = __builtin_bit_cast(u32x4, f);
    }
  }
  // B operand of S^T = dy^T I: lane (w' = c, h), k-step s, element j = [w' == 16s + 8(j>>2) + 4h + (j&3)]
  u32x4 Ip[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    frag_t f;
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = E::from_f(c == 16 * s + 8 * (j >> 2) + 4 * h + (j & 3) ? 1.f : 0.f);
    Ip[s] = __builtin_bit_cast(u32x4, f);
  }

  // ---- staging slots of this thread: item = (frame of the batch, row, 16-byte chunk) ----
  unsigned gx_off[NITX], gd_off[NITD];                      // byte offset from the batch's first frame (OOB: row >= V)
  int lx_off[NITX], ld_off[NITD];                           // element offset in a buffer
#pragma unroll
  for (int j = 0; j < NITX; ++j) {
    const int idx = tid + j * RC_NTH;
    const int q = idx % QX, v = (idx / QX) % 32, f = idx / (QX * 32);
    gx_off[j] = v < V ? (unsigned)(((f * V + v) * Cin + ib * 64 + 8 * q) * 2) : 0x7ffffff0u;
    lx_off[j] = f * XFRM + v * XROW + ((q ^ (4 * ((v >> 1) & 1))) * 8);
  }
  if constexpr (CN != 0) {
    // narrow rows: thread idx < 4 * 32 owns (frame idx >> 5, row idx & 31) and fills channel vector 0; the rest load nothing
    const int v = tid & 31, f = tid >> 5;
    gx_off[0] = (tid < WG_FBT * 32 && v < V) ? (unsigned)((f * V + v) * CN * 2) : 0x7ffffff0u;
    lx_off[0] = (tid < WG_FBT * 32) ? f * XFRM + v * XROW + ((0 ^ (4 * ((v >> 1) & 1))) * 8) : -1;
  }
#pragma unroll
  for (int j = 0; j < NITD; ++j) {
    const int idx = tid + j * RC_NTH;
    const int q = idx % QD, v = (idx / QD) % 32, f = idx / (QD * 32);
    gd_off[j] = v < V ? (unsigned)(((f * V + v) * Cout + cb * 32 * CT + 8 * q) * 2) : 0x7ffffff0u;
    const int sw = CT == 2 ? 4 * ((v >> 1) & 1) : 4 * (v & 3);
    ld_off[j] = WG_FBT * XFRM + f * DFRM + v * DROW + ((q ^ sw) * 8);
  }
  const T* xg = reinterpret_cast<const T*>(P.x);
  const T* dyg = reinterpret_cast<const T*>(P.dy);
  const long long F = (long long)P.NM * P.T;                // frames; frame g starts at element g * V * C (no strides)
  const long long NB = (F + WG_FBT - 1) / WG_FBT;
  auto issue = [&](long long bt, u32x4 (&rx)[NITX], u32x4 (&rd)[NITD]) __attribute__((always_inline)) {
    const long long g0 = bt * WG_FBT;
    const int nfr = (int)min((long long)WG_FBT, F - g0);
    const rsrc_t r0 = make_rsrc(xg + g0 * V * Cin, (unsigned)(nfr * V * Cin) * 2u);
    const rsrc_t r1 = make_rsrc(dyg + g0 * V * Cout, (unsigned)(nfr * V * Cout) * 2u);
    if constexpr (CN != 0) {
      uint32_t e[4] = {0u, 0u, 0u, 0u};
#pragma unroll
      for (int j = 0; j < CN; ++j) e[j] = (uint32_t)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(r0, gx_off[0] + 2u * j, 0, 0);
      rx[0] = u32x4{e[0] | (e[1] << 16), e[2] | (e[3] << 16), 0u, 0u};
    } else {
#pragma unroll
      for (int j = 0; j < NITX; ++j) rx[j] = __builtin_amdgcn_raw_buffer_load_b128(r0, gx_off[j], 0, RCW_X_AUX);
    }
#pragma unroll
    for (int j = 0; j < NITD; ++j) rd[j] = __builtin_amdgcn_raw_buffer_load_b128(r1, gd_off[j], 0, RCW_DY_AUX);
  };
  auto commit = [&](int half, u32x4 (&rx)[NITX], u32x4 (&rd)[NITD]) __attribute__((always_inline)) {
    T* bufp = lds + half * BUF;
    if constexpr (CN != 0) {
      if (lx_off[0] >= 0) *reinterpret_cast<u32x4*>(bufp + lx_off[0]) = rx[0];
    } else {
#pragma unroll
      for (int j = 0; j < NITX; ++j) *reinterpret_cast<u32x4*>(bufp + lx_off[j]) = rx[j];
    }
#pragma unroll
    for (int j = 0; j < NITD; ++j) *reinterpret_cast<u32x4*>(bufp + ld_off[j]) = rd[j];
  };

  // ---- lane constants of the transposed reads (T10: lane 4q+p of a 16-lane group addresses row q, columns 4p..4p+3) ----
  const int qq = (lane & 15) >> 2, g1 = (lane >> 4) & 1;
  // x^T (B operand, natural k order): block rows 16s + 8h (+4), columns 32 itl + 16 g1 + 4 (lane&3)
  const int xq = 4 * itl + 2 * g1 + ((lane & 3) >> 1);
  const int xlane = (8 * h + qq) * XROW + ((xq ^ (4 * ((qq >> 1) & 1))) * 8) + 4 * (lane & 1);
  // dy^T (A operand, chained k order): block rows 16s + 4h (+8), columns 32 (ct + o) + 16 g1 + 4 (lane&3)
  const int dsw = CT == 2 ? 4 * ((qq >> 1) & 1) : 4 * qq;
  int dlane[OT];
#pragma unroll
  for (int o = 0; o < OT; ++o) {
    const int dq = 4 * (ct + o) + 2 * g1 + ((lane & 3) >> 1);
    dlane[o] = WG_FBT * XFRM + (4 * h + qq) * DROW + ((dq ^ dsw) * 8) + 4 * (lane & 1);
  }

  f32x16 acc[OT][K], accS[OT];
#pragma unroll
  for (int o = 0; o < OT; ++o) {
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[o][k][i] = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) accS[o][i] = 0.f;
  }

  auto frame = [&](const T* bufp, int f) __attribute__((always_inline)) {
    const T* xb = bufp + f * XFRM + xlane;
    frag_t xT[2], dT[OT][2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      xT[s] = tr_pair<T>(xb + 16 * s * XROW, xb + (16 * s + 4) * XROW);
#pragma unroll
      for (int o = 0; o < OT; ++o) {
        const T* db = bufp + f * DFRM + dlane[o];
        dT[o][s] = tr_pair<T>(db + 16 * s * DROW, db + (16 * s + 8) * DROW);
      }
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
      f32x16 XA;
#pragma unroll
      for (int i = 0; i < 16; ++i) XA[i] = 0.f;
      mma_kgroup(XA, __builtin_bit_cast(frag_t, At[k][0]), xT[0]);
      mma_kgroup(XA, __builtin_bit_cast(frag_t, At[k][1]), xT[1]);
      u32x4 xab[2];
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int q = 0; q < 4; ++q) xab[s][q] = pack2<T>(XA[8 * s + 2 * q], XA[8 * s + 2 * q + 1]);
#pragma unroll
      for (int o = 0; o < OT; ++o) {
        mma_kgroup(acc[o][k], dT[o][0], __builtin_bit_cast(frag_t, xab[0]));
        mma_kgroup(acc[o][k], dT[o][1], __builtin_bit_cast(frag_t, xab[1]));
      }
    }
    if (itl == 0) {
#pragma unroll
      for (int o = 0; o < OT; ++o) {
        mma_kgroup(accS[o], dT[o][0], __builtin_bit_cast(frag_t, Ip[0]));
        mma_kgroup(accS[o], dT[o][1], __builtin_bit_cast(frag_t, Ip[1]));
      }
    }
  };

  // ---- the walk over this group's batches ----
  u32x4 rx[NITX], rd[NITD];
  long long bt = grp;
  if (bt < NB) {
    issue(bt, rx, rd);
    commit(0, rx, rd);
  }
  __syncthreads();
  int half = 0;
  for (; bt < NB; bt += P.G) {
    const long long nxt = bt + P.G;
    issue(nxt < NB ? nxt : bt, rx, rd);                      // unconditional (beyond the end: re-read, never used)
    __builtin_amdgcn_sched_barrier(0);
    const T* bufp = lds + half * BUF;
#pragma unroll
    for (int f = fg; f < WG_FBT; f += FG) frame(bufp, f);
    commit(half ^ 1, rx, rd);
    half ^= 1;
    __syncthreads();
  }

  // ---- flush ----
  float* red = reinterpret_cast<float*>(smem);               // frame groups 1.. hand their sums to group 0 through LDS, one at a time
  const int pw = ctg * 2 + itl;                              // this wave's slot within a frame group
  constexpr int SLOT = OT * (K + 1) * 16 * 64;               // floats per slot
#pragma unroll 1
  for (int gsrc = 1; gsrc < FG; ++gsrc) {
    __syncthreads();
    if (fg == gsrc) {
#pragma unroll
      for (int o = 0; o < OT; ++o) {
#pragma unroll
        for (int k = 0; k < K; ++k)
#pragma unroll
          for (int i = 0; i < 16; ++i) red[pw * SLOT + ((o * (K + 1) + k) * 16 + i) * 64 + lane] = acc[o][k][i];
#pragma unroll
        for (int i = 0; i < 16; ++i) red[pw * SLOT + ((o * (K + 1) + K) * 16 + i) * 64 + lane] = accS[o][i];
      }
    }
    __syncthreads();
    if (fg == 0) {
#pragma unroll
      for (int o = 0; o < OT; ++o) {
#pragma unroll
        for (int k = 0; k < K; ++k)
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[o][k][i] += red[pw * SLOT + ((o * (K + 1) + k) * 16 + i) * 64 + lane];
#pragma unroll
        for (int i = 0; i < 16; ++i) accS[o][i] += red[pw * SLOT + ((o * (K + 1) + K) * 16 + i) * 64 + lane];
      }
    }
  }
  if (fg == 0) {
    const int n0 = K * Cout * Cin;
    const int icol = ib * 64 + 32 * itl + c;
    const bool col_ok = icol < Cin;                           // (the 3-channel first layer: 3 of the 64 columns exist)
#pragma unroll
    for (int o = 0; o < OT; ++o) {
      const int crow = cb * 32 * CT + 32 * (ct + o);
      if (P.ws) {
        float* sl = P.ws + (size_t)grp * P.ws_slice;
        if (col_ok) {
#pragma unroll
          for (int k = 0; k < K; ++k)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
              const int r = (i & 3) + 8 * (i >> 2) + 4 * h;
              sl[((size_t)k * Cout + crow + r) * Cin + icol] = acc[o][k][i];
            }
        }
        if (itl == 0 && ib == 0 && P.S && c < V) {
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int r = (i & 3) + 8 * (i >> 2) + 4 * h;
            sl[n0 + c * Cout + crow + r] = accS[o][i];
          }
        }
      } else {
        if (col_ok) {
#pragma unroll
          for (int k = 0; k < K; ++k)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
              const int r = (i & 3) + 8 * (i >> 2) + 4 * h;
              atomicAdd(P.dW + ((size_t)k * Cout + crow + r) * Cin + icol, acc[o][k][i]);
            }
        }
        if (itl == 0 && ib == 0 && P.S && c < V) {
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int r = (i & 3) + 8 * (i >> 2) + 4 * h;
            atomicAdd(P.S + c * Cout + crow + r, accS[o][i]);
          }
        }
      }
    }
  }
}

// acc + (low / high half of a packed 16-bit pair): one v_dot2 against the constant pair (1, 0) / (0, 1)
template <typename T> __device__ static inline float sdot_lo(uint32_t p, float acc);
template <typename T> __device__ static inline float sdot_hi(uint32_t p, float acc);
template <> __device__ inline float sdot_lo<__bf16>(uint32_t p, float acc) {
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, p), __builtin_bit_cast(bf16x2, 0x00003f80u), acc, false);
}
template <> __device__ inline float sdot_hi<__bf16>(uint32_t p, float acc) {
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, p), __builtin_bit_cast(bf16x2, 0x3f800000u), acc, false);
}
template <> __device__ inline float sdot_lo<_Float16>(uint32_t p, float acc) {
  return __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2, p), __builtin_bit_cast(f16x2, 0x00003c00u), acc, false);
}
template <> __device__ inline float sdot_hi<_Float16>(uint32_t p, float acc) {
  return __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2, p), __builtin_bit_cast(f16x2, 0x3c000000u), acc, false);
}

// s_waitcnt vmcnt(n) with a run-time (wave-uniform) n: the immediate must be a constant
__device__ static inline void wait_vm(int n) {
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
  }
}

// LDS-DMA: lane l fetches 16 bytes from base + voff and the wave's 1 KB lands at LDS byte address `dst` (wave-uniform) +
// 16 l.  Inline asm ON PURPOSE: after the builtin (__builtin_amdgcn_global_load_lds) hipcc puts s_waitcnt vmcnt(0) in front
// of the next LDS read -- it cannot tell the buffer being filled from the one being read -- and the two batches that are
// meant to stay in flight would be drained every batch.  The compiler neither sees nor counts these loads: every wait for
// them is an explicit wait_vm(), and the loop contains no other memory instruction that counts on vmcnt.  M0 (the DMA's
// LDS base) is saved and restored inside the statement.
__device__ static inline void glds16(const void* base, unsigned voff, unsigned dst) {
  unsigned keep;
  // (wave-uniform by construction; the readfirstlanes are there for the register class of the asm operands)
  dst = (unsigned)__builtin_amdgcn_readfirstlane((int)dst);
  const unsigned long long b64 = (unsigned long long)base;
  const unsigned blo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b64);             // (unsigned BEFORE widening:
  const unsigned bhi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(b64 >> 32));     //  the builtin returns int)
  base = (const void*)(((unsigned long long)bhi << 32) | (unsigned long long)blo);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(dst), "s"(base) : "memory");
}

// The weight gradient with LDS-DMA staging and a two-deep XA pipeline (header, "Round 5").  Template parameters and the
// wave roles as in gcn_rc_wgrad_kernel; 64-channel input rows only (CN == 0).
template <typename T, int K, int CT, int OT>
__global__ __launch_bounds__(RC_NTH, 2) void gcn_rc_wgrad_dma_kernel(const RcWgParams P) {
  using E = Elem<T>;
  typedef typename E::frag frag_t;
  constexpr int CTG = CT / OT;
  constexpr int FG = 8 / (2 * CTG);
  constexpr int XROW = 64, DROW = 32 * CT;
  constexpr int XFRM = 32 * XROW, DFRM = 32 * DROW;
  constexpr int BUF = WG_FBT * (XFRM + DFRM);               // elements per buffer
  constexpr int NBUF = 3;
  constexpr int RPX = 8, RPD = 512 / DROW;                  // image rows per 1 KB DMA piece (x: 128-byte rows; dy: 256 / 128)
  constexpr int NSLOT = (WG_FBT * (32 / RPX + 32 / RPD) + 7) / 8;      // pieces per wave and batch, at most
  static_assert(NSLOT <= 6, "wait_vm covers six pieces per wave");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T* lds = reinterpret_cast<T*>(smem);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int itl = wave8 & 1, ctg = (wave8 >> 1) % CTG, fg = (wave8 >> 1) / CTG;
  const int ct = ctg * OT;
  const int b = blockIdx.x;
  const int blk = (b >> 3) % P.gy;
  const int grp = (b / (8 * P.gy)) * 8 + (b & 7);
  const int ib = blk % P.nib, cb = blk / P.nib;
  const int V = P.V, Cin = P.Cin, Cout = P.Cout;
  const int c = lane & 31, h = lane >> 5;

  // ---- setup: zero the three buffers (pad rows are never written afterwards), adjacency -> per-lane fragments ----
  for (int i = tid; i < NBUF * BUF / 8; i += RC_NTH) reinterpret_cast<u32x4*>(smem)[i] = u32x4{0u, 0u, 0u, 0u};
  float* Asc = reinterpret_cast<float*>(smem + (size_t)NBUF * BUF * 2);
  for (int i = tid; i < K * V * V; i += RC_NTH) Asc[i] = P.A[i];
  __syncthreads();
  u32x4 At[K][2];
#pragma unroll
  for (int k = 0; k < K; ++k) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      frag_t f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int v = 16 * s + 8 * h + j;
        f[j] = E::from_f((c < V && v < V) ? Asc[(k * V + v) * V + c] : 0.f);
      }
      At[k][s] = __builtin_bit_cast(u32x4, f);
    }
  }
  u32x4 Ip[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    frag_t f;
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = E::from_f(c == 16 * s + 8 * (j >> 2) + 4 * h + (j & 3) ? 1.f : 0.f);
    Ip[s] = __builtin_bit_cast(u32x4, f);
  }

  // ---- DMA pieces of a batch: [frame][row group] of x, then of dy; only row groups that hold a row < V.  Piece pid =
  //      slot * 8 + wave.  A lane fetches the 16-byte chunk that belongs at ITS place of the (lane-linear) piece: chunk
  //      position p of image row v holds memory chunk p ^ swizzle(v). ----
  const int RX = (V + RPX - 1) / RPX, RD = (V + RPD - 1) / RPD;
  const int NPT = WG_FBT * (RX + RD);
  unsigned goff[NSLOT];                                     // this lane's byte offset from the batch's first frame (x or dy)
  int loff[NSLOT], pfr[NSLOT];                              // piece's byte offset in a buffer, its frame; -1: no piece in this slot
  bool pdy[NSLOT];
  unsigned lane_ok = 0;                                     // bit s: this lane's row of slot s exists (v < V)
  int np_full = 0;
#pragma unroll
  for (int s = 0; s < NSLOT; ++s) {
    const int pid = s * 8 + wave8;
    pfr[s] = -1; loff[s] = 0; goff[s] = 0; pdy[s] = false;
    if (pid < NPT) {
      ++np_full;
      if (pid < WG_FBT * RX) {
        const int f = pid / RX, rg = pid - f * RX;
        const int v = RPX * rg + (lane >> 3), pos = lane & 7;
        const int q = pos ^ (4 * ((v >> 1) & 1));
        goff[s] = (unsigned)(((f * V + v) * Cin + ib * 64 + 8 * q) * 2);
        loff[s] = (f * XFRM + RPX * rg * XROW) * 2;
        pfr[s] = f;
        if (v < V) lane_ok |= 1u << s;
      } else {
        const int qd = pid - WG_FBT * RX;
        const int f = qd / RD, rg = qd - f * RD;
        constexpr int LPR = 64 / RPD;                       // lanes (= 16-byte chunks) per image row
        const int v = RPD * rg + lane / LPR, pos = lane % LPR;
        const int sw = CT == 2 ? 4 * ((v >> 1) & 1) : 4 * (v & 3);
        goff[s] = (unsigned)(((f * V + v) * Cout + cb * 32 * CT + 8 * (pos ^ sw)) * 2);
        loff[s] = (WG_FBT * XFRM + f * DFRM + RPD * rg * DROW) * 2;
        pfr[s] = f;
        pdy[s] = true;
        if (v < V) lane_ok |= 1u << s;
      }
    }
  }
  const unsigned char* xg = reinterpret_cast<const unsigned char*>(P.x);
  const unsigned char* dyg = reinterpret_cast<const unsigned char*>(P.dy);
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const long long F = (long long)P.NM * P.T;
  const long long NB = (F + WG_FBT - 1) / WG_FBT;
  auto issue = [&](long long bt, int buf) __attribute__((always_inline)) {
    const long long g0 = bt * WG_FBT;
    const int nfr = (int)min((long long)WG_FBT, F - g0);
    const unsigned char* xb = xg + (size_t)g0 * V * Cin * 2;
    const unsigned char* db = dyg + (size_t)g0 * V * Cout * 2;
    const unsigned lb = lds_base + (unsigned)buf * (BUF * 2);
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) {
      if (pfr[s] >= 0 && pfr[s] < nfr) {                    // wave-uniform
        if ((lane_ok >> s) & 1u) glds16(pdy[s] ? db : xb, goff[s], lb + (unsigned)loff[s]);
      }
    }
  };

  // ---- lane constants of the transposed reads (as above) ----
  const int qq = (lane & 15) >> 2, g1 = (lane >> 4) & 1;
  const int xq = 4 * itl + 2 * g1 + ((lane & 3) >> 1);
  const int xlane = (8 * h + qq) * XROW + ((xq ^ (4 * ((qq >> 1) & 1))) * 8) + 4 * (lane & 1);
  const int dsw = CT == 2 ? 4 * ((qq >> 1) & 1) : 4 * qq;
  int dlane[OT];
#pragma unroll
  for (int o = 0; o < OT; ++o) {
    const int dq = 4 * (ct + o) + 2 * g1 + ((lane & 3) >> 1);
    dlane[o] = WG_FBT * XFRM + (4 * h + qq) * DROW + ((dq ^ dsw) * 8) + 4 * (lane & 1);
  }

  f32x16 acc[OT][K], accS[OT];
#pragma unroll
  for (int o = 0; o < OT; ++o) {
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[o][k][i] = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) accS[o][i] = 0.f;
  }

#ifndef RCW_ABL
#define RCW_ABL 0
#endif
  frag_t xT[2], dT[OT][2];
  bool first = true;
  auto frame = [&](const T* bufp, int f) __attribute__((always_inline)) {
    const T* xb = bufp + f * XFRM + xlane;
#if (RCW_ABL & 1)
    if (first) {
      first = false;
#endif
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      xT[s] = tr_pair<T>(xb + 16 * s * XROW, xb + (16 * s + 4) * XROW);
#pragma unroll
      for (int o = 0; o < OT; ++o) {
        const T* db = bufp + f * DFRM + dlane[o];
        dT[o][s] = tr_pair<T>(db + 16 * s * DROW, db + (16 * s + 8) * DROW);
      }
    }
#if (RCW_ABL & 1)
    }
#endif
    // two XA tiles: partition k + 1 is aggregated before partition k is converted and contracted
    f32x16 XA[2];
    auto agg = [&](int k) __attribute__((always_inline)) {
      f32x16& X = XA[k & 1];
#pragma unroll
      for (int i = 0; i < 16; ++i) X[i] = 0.f;
#if (RCW_ABL & 4)
      X[0] = __builtin_bit_cast(float, At[k][0][0]); X[9] = (float)xT[1][0];
#else
      mma_kgroup(X, __builtin_bit_cast(frag_t, At[k][0]), xT[0]);
      mma_kgroup(X, __builtin_bit_cast(frag_t, At[k][1]), xT[1]);
#endif
    };
    agg(0);
#pragma unroll
    for (int k = 0; k < K; ++k) {
      if (k + 1 < K) agg(k + 1);
      __builtin_amdgcn_sched_barrier(0);
      const f32x16& X = XA[k & 1];
      u32x4 xab[2];
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
#if (RCW_ABL & 2)
        for (int q = 0; q < 4; ++q) xab[s][q] = __builtin_bit_cast(uint32_t, X[4 * s + q]);
#else
        for (int q = 0; q < 4; ++q) xab[s][q] = pack2<T>(X[8 * s + 2 * q], X[8 * s + 2 * q + 1]);
#endif
#if (RCW_ABL & 16)
      acc[0][k][0] += __builtin_bit_cast(float, xab[0][0]) + __builtin_bit_cast(float, xab[1][3]);
#else
#pragma unroll
      for (int o = 0; o < OT; ++o) {
        mma_kgroup(acc[o][k], dT[o][0], __builtin_bit_cast(frag_t, xab[0]));
        mma_kgroup(acc[o][k], dT[o][1], __builtin_bit_cast(frag_t, xab[1]));
      }
#endif
      __builtin_amdgcn_sched_barrier(0);
    }
#ifdef RCW_SDOT
    {
      // S on the vector ALU: the dy^T fragments ARE the addends (lane = channel, elements = joints 16s + 8(j>>2) + 4h + (j&3));
      // v_dot2 against (1, 0) / (0, 1) adds one half of a packed pair without unpacking it
      const int o = OT == 2 ? itl : 0;
      if (OT == 2 || itl == 0) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const u32x4 dv = __builtin_bit_cast(u32x4, o == 0 ? dT[0][s] : dT[OT - 1][s]);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            accS[0][8 * s + 2 * q] = sdot_lo<T>(dv[q], accS[0][8 * s + 2 * q]);
            accS[0][8 * s + 2 * q + 1] = sdot_hi<T>(dv[q], accS[0][8 * s + 2 * q + 1]);
          }
        }
      }
    }
#elif !(RCW_ABL & 8)
    if constexpr (OT == 2) {
      // S^T += dy^T I: the waves of input tile 0 take output tile 0, those of input tile 1 output tile 1
      if (itl == 0) {
        mma_kgroup(accS[0], dT[0][0], __builtin_bit_cast(frag_t, Ip[0]));
        mma_kgroup(accS[0], dT[0][1], __builtin_bit_cast(frag_t, Ip[1]));
      } else {
        mma_kgroup(accS[1], dT[1][0], __builtin_bit_cast(frag_t, Ip[0]));
        mma_kgroup(accS[1], dT[1][1], __builtin_bit_cast(frag_t, Ip[1]));
      }
    } else if (itl == 0) {
      mma_kgroup(accS[0], dT[0][0], __builtin_bit_cast(frag_t, Ip[0]));
      mma_kgroup(accS[0], dT[0][1], __builtin_bit_cast(frag_t, Ip[1]));
    }
#endif
  };

  // ---- the walk over this group's batches: batch i computes from buffer i % 3 while batches i + 1 and i + 2 land ----
  long long bt = grp;
  if (bt < NB) issue(bt, 0);
  if (bt + P.G < NB) issue(bt + P.G, 1);
  int cur = 0;
  for (; bt < NB; bt += P.G) {
    const long long nxt = bt + P.G;
    // batch `bt` has landed when at most the pieces of batch `nxt` are outstanding -- np_full of them if it is a full batch
    wait_vm((nxt < NB && (nxt + 1) * WG_FBT <= F) ? np_full : 0);
    __builtin_amdgcn_s_barrier();                            // ... for every wave; and every wave is done with buffer cur - 1
    if (nxt + P.G < NB) issue(nxt + P.G, cur == 0 ? 2 : cur - 1);
    __builtin_amdgcn_sched_barrier(0);
    const T* bufp = lds + cur * BUF;
    const int nfr = (int)min((long long)WG_FBT, F - bt * WG_FBT);
#pragma unroll
    for (int f = fg; f < WG_FBT; f += FG)
      if (f < nfr) frame(bufp, f);
    cur = cur == 2 ? 0 : cur + 1;
  }
  wait_vm(0);
  __syncthreads();

  // ---- flush (as above; S: the wave that accumulated the tile writes it) ----
  float* red = reinterpret_cast<float*>(smem);
  const int pw = ctg * 2 + itl;
  constexpr int SLOT = OT * (K + 1) * 16 * 64;
#pragma unroll 1
  for (int gsrc = 1; gsrc < FG; ++gsrc) {
    __syncthreads();
    if (fg == gsrc) {
#pragma unroll
      for (int o = 0; o < OT; ++o) {
#pragma unroll
        for (int k = 0; k < K; ++k)
#pragma unroll
          for (int i = 0; i < 16; ++i) red[pw * SLOT + ((o * (K + 1) + k) * 16 + i) * 64 + lane] = acc[o][k][i];
#pragma unroll
        for (int i = 0; i < 16; ++i) red[pw * SLOT + ((o * (K + 1) + K) * 16 + i) * 64 + lane] = accS[o][i];
      }
    }
    __syncthreads();
    if (fg == 0) {
#pragma unroll
      for (int o = 0; o < OT; ++o) {
#pragma unroll
        for (int k = 0; k < K; ++k)
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[o][k][i] += red[pw * SLOT + ((o * (K + 1) + k) * 16 + i) * 64 + lane];
#pragma unroll
        for (int i = 0; i < 16; ++i) accS[o][i] += red[pw * SLOT + ((o * (K + 1) + K) * 16 + i) * 64 + lane];
      }
    }
  }
  if (fg == 0) {
    const int n0 = K * Cout * Cin;
    const int icol = ib * 64 + 32 * itl + c;
#pragma unroll
    for (int o = 0; o < OT; ++o) {
      const int crow = cb * 32 * CT + 32 * (ct + o);
      const bool mine = (OT == 2 ? itl == o : itl == 0) && ib == 0 && P.S && c < V;       // this wave holds S of tile o
      if (P.ws) {
        float* sl = P.ws + (size_t)grp * P.ws_slice;
#pragma unroll
        for (int k = 0; k < K; ++k)
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int r = (i & 3) + 8 * (i >> 2) + 4 * h;
            sl[((size_t)k * Cout + crow + r) * Cin + icol] = acc[o][k][i];
          }
#ifdef RCW_SDOT
        if ((OT == 2 ? itl == o : itl == 0) && ib == 0 && P.S) {
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int w = 16 * (i >> 3) + 8 * ((i & 7) >> 2) + 4 * h + (i & 3);
            if (w < V) sl[n0 + w * Cout + crow + c] = accS[0][i];
          }
        }
#else
        if (mine) {
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int r = (i & 3) + 8 * (i >> 2) + 4 * h;
            sl[n0 + c * Cout + crow + r] = accS[o][i];
          }
        }
#endif
      } else {
#pragma unroll
        for (int k = 0; k < K; ++k)
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int r = (i & 3) + 8 * (i >> 2) + 4 * h;
            atomicAdd(P.dW + ((size_t)k * Cout + crow + r) * Cin + icol, acc[o][k][i]);
          }
#ifdef RCW_SDOT
        if ((OT == 2 ? itl == o : itl == 0) && ib == 0 && P.S) {
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int w = 16 * (i >> 3) + 8 * ((i & 7) >> 2) + 4 * h + (i & 3);
            if (w < V) atomicAdd(P.S + w * Cout + crow + c, accS[0][i]);
          }
        }
#else
        if (mine) {
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int r = (i & 3) + 8 * (i >> 2) + 4 * h;
            atomicAdd(P.S + c * Cout + crow + r, accS[o][i]);
          }
        }
#endif
      }
    }
  }
}

template <typename T, int K, int CT, int CN = 0, int OT = 1>
int rc_wg_launch(RcWgParams P, int grid_cap, hipStream_t stream) {
  void (*kfn)(const RcWgParams);
  if constexpr (CN == 0) kfn = gcn_rc_wgrad_dma_kernel<T, K, CT, OT>;      // three DMA buffers
  else kfn = gcn_rc_wgrad_kernel<T, K, CT, CN, OT>;
  constexpr int FG = 8 / (2 * (CT / OT));
  size_t lds = (size_t)(CN == 0 ? 3 : 2) * WG_FBT * (32 * 64 + 32 * 32 * CT) * 2 + (size_t)K * 32 * 32 * 4;
  const size_t redb = FG >= 2 ? (size_t)2 * (CT / OT) * OT * (K + 1) * 16 * 64 * 4 : 0;
  if (redb > lds) lds = redb;
  if (lds > 160 * 1024) return ISTGCN_EINVAL;
  static std::atomic<unsigned long long> optin{0};
  if (int ea = istgcn_lds_optin((const void*)kfn, optin)) return ea;
  int res = grid_cap > 0 ? grid_cap : istgcn_resident_blocks((const void*)kfn, RC_NTH, lds);
  P.nib = CN ? 1 : P.Cin / 64;
  P.gy = P.nib * (P.Cout / (32 * CT));
  int G = res / P.gy / 8 * 8;
  if (G < 8) G = 8;
  const long long NB = ((long long)P.NM * P.T + WG_FBT - 1) / WG_FBT;
  while (G > 8 && G - 8 >= NB) G -= 8;
  P.G = G;
  const long long n0 = (long long)K * P.Cout * P.Cin, n1 = P.S ? (long long)P.V * P.Cout : 0;
  const bool use_ws = P.ws && (long long)G * (n0 + n1) <= P.ws_slice && G >= 64;
  if (use_ws) {
    // the slices must start as zeros where no workgroup writes (S rows of input blocks != 0 are never written: only the
    // ib == 0 block writes S, and it writes every entry of its channels) -- every (k, c, i) entry IS written by exactly
    // one workgroup of the group, every S entry by exactly one: no zero fill needed
    P.ws_slice = n0 + n1;
  } else {
    P.ws = nullptr;
  }
  ISTGCN_LAUNCH(kfn, dim3(G * P.gy), dim3(RC_NTH), lds, stream, P);
  ISTGCN_CHECK_LAUNCH();
  if (use_ws) return istgcn_wgrad_reduce(P.ws, n0 + n1, G, P.dW, (int)n0, P.S, (int)n1, stream);
  return ISTGCN_OK;
}

template <typename T, int K>
int rc_wg_ct(const RcWgParams& P, int grid_cap, hipStream_t stream) {
  if (P.Cin == 3) return rc_wg_launch<T, K, 2, 3>(P, grid_cap, stream);
  // two output tiles per wave where the flush image of a frame group fits LDS (K <= 3)
#ifdef ISTGCN_EXPERIMENT
  static const bool one = [] { const char* e = getenv("ISTGCN_GWG_OT"); return e && atoi(e) == 1; }();     // 1: one tile (A/B timing)
#else
  constexpr bool one = false;
#endif
  const bool ot2 = K <= 3 && !one;
  if constexpr (K <= 3) {
    if (ot2) {
      if (P.Cout % 128 == 0) return rc_wg_launch<T, K, 4, 0, 2>(P, grid_cap, stream);
      return rc_wg_launch<T, K, 2, 0, 2>(P, grid_cap, stream);
    }
  }
  if (P.Cout % 128 == 0) return rc_wg_launch<T, K, 4>(P, grid_cap, stream);
  return rc_wg_launch<T, K, 2>(P, grid_cap, stream);
}

template <typename T>
int rc_wg_k(const RcWgParams& P, int K, int grid_cap, hipStream_t stream) {
  switch (K) {
    case 1: return rc_wg_ct<T, 1>(P, grid_cap, stream);
    case 2: return rc_wg_ct<T, 2>(P, grid_cap, stream);
    case 3: return rc_wg_ct<T, 3>(P, grid_cap, stream);
    case 4: return rc_wg_ct<T, 4>(P, grid_cap, stream);
  }
  return ISTGCN_EINVAL;
}

}  // namespace

extern "C" int istgcn_gcn_wgrad_rc_ok(int V, int Cin, int Cout, int K, int dtype) {
  return (dtype == 1 || dtype == 2) && V <= 32 && ((Cin >= 64 && Cin % 64 == 0) || Cin == 3) && Cout >= 64 && Cout % 64 == 0 &&
         K >= 1 && K <= 4;
}

extern "C" int istgcn_gcn_wgrad_rc(const void* dy, const void* x, const float* A, float* dW, float* S, int NM, int T, int V,
                                   int Cin, int Cout, int K, int dtype, int grid_cap, float* ws, long long ws_floats,
                                   void* stream) {
  if (!istgcn_gcn_wgrad_rc_ok(V, Cin, Cout, K, dtype)) return ISTGCN_EINVAL;
  RcWgParams P{};
  P.dy = dy; P.x = x; P.A = A; P.dW = dW; P.S = S; P.ws = ws_floats > 0 ? ws : nullptr; P.ws_slice = ws_floats;
  P.NM = NM; P.T = T; P.V = V; P.Cin = Cin; P.Cout = Cout;
  if (dtype == 1) return rc_wg_k<__bf16>(P, K, grid_cap, (hipStream_t)stream);
  return rc_wg_k<_Float16>(P, K, grid_cap, (hipStream_t)stream);
}
