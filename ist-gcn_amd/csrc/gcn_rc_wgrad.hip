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

template <typename T, int K, int CT, int CN = 0, int OT = 1>
int rc_wg_launch(RcWgParams P, int grid_cap, hipStream_t stream) {
  auto kfn = gcn_rc_wgrad_kernel<T, K, CT, CN, OT>;
  constexpr int FG = 8 / (2 * (CT / OT));
  size_t lds = (size_t)2 * WG_FBT * (32 * 64 + 32 * 32 * CT) * 2 + (size_t)K * 32 * 32 * 4;
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
