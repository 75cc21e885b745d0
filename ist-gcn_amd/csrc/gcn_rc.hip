// Graph-convolution unit, "register-chained" kernels for 16-bit storage (round 3).
//
//   y[t,w,c] = sum_k sum_v A_k[v][w] * ( sum_i W_k[c][i] * x[t,v,i] )  + bterm[w][c]
//
// Replaces (reference file:line): net/utils/tgcn.py:76-89 (1x1 Conv2d :79, einsum :86), net/utils/tgcn_multi3_fix_3A.py:76-92,
// net/utils/inceptionv2_gcn.py:64-89 (variants fold into one A, host side).
//
// Round 2's kernels aggregated first and kept the aggregated tile xa_k = A_k^T x in LDS between the two MFMA stages: the
// SQ counters had 35 % of their LDS cycles in bank conflicts and the matrix pipe 19 % busy.  Here the
// order is GEMM FIRST and the intermediate never leaves the register file:
//   1. H_k[v][c] = sum_i x[v][i] W_k[c][i]     MFMA 32x32x16, A operand = the frame's rows of x (joints padded to 32
//                                              rows) loaded STRAIGHT from HBM as 16-byte vectors (channels are the k axis
//                                              and contiguous in memory: no LDS, no transpose), B operand = W_k from LDS
//   2. Y[w][c] += sum_v A_k[v][w] H_k[v][c]    the 32x32 accumulator tile of step 1 has its column (c) on the lane and its
//                                              rows (v) in the 16 registers, so it IS the B operand of a product that sums
//                                              over v: registers 8s..8s+7 converted pairwise to 16 bit are k-step s, in the
//                                              k order 16s + 8(j>>2) + 4h + (j&3); the A operand (A_k^T in that order) is a
//                                              per-lane constant for the whole launch
//   3. Y -> 16 bit, BatchNorm sums, a 4 KB per-wave LDS image turns (lane = channel, registers = joints) into 16-byte
//      row vectors, stored as whole 128-byte lines.
// Every wave is autonomous: it owns whole frames (grid-stride over all NM*T frames) x one 64-channel pair, prefetches its
// next frame's operand registers while it computes, and never meets a barrier inside the loop.  The weights of the
// workgroup's channel slice (<= 98 KB) are copied to LDS once.
#include "gcn_rc.hpp"
#include "bn_tail.hpp"

extern "C" int istgcn_gcn_rc_layout(int Cin, int Cout, int K, int dtype);

namespace {

struct RcFwdParams {
  const void* x; const void* Wq; const float* A; const float* bterm; const void* addend; void* y; double* stats;
  int NM, Tin, Tout, Tlog, V, Cout, in_t_stride, out_t_stride, stats_rep;
  int nfw;               // frame workers in the grid
  int step_n, step_t;    // nfw = step_n * Tlog + step_t: a worker's (sequence, frame) cursor advances without dividing
  int gy;                // output-channel slices: workgroups that share frames (placed on one XCD: they re-read x from its L2)
  BnTail tail;           // "last workgroup finalises" the BatchNorm that follows (bn_tail.hpp), when the caller armed it
};

// S = Cin / 16 (k-steps of the channel contraction), NCP = 64-channel pairs per workgroup slice (waves of a workgroup:
// 8 / NCP frame workers x NCP pairs), PF2 = the next frame's operand registers are a second set (else the loads reuse
// the set right after the frame's last contraction MFMA and the epilogue covers their latency).
template <typename T, int S, int K, int NCP, bool PF2, bool ADD, int CN>
__global__ __launch_bounds__(RC_NTH, 2) void gcn_rc_fwd_kernel(const RcFwdParams P) {
  using E = Elem<T>;
  typedef typename E::frag frag_t;
  constexpr int CIN = 16 * S;                                // contraction extent (k-steps x 16)
  constexpr int CROW = CN ? CN : CIN;                        // channels of a row in memory (CN: the 3-channel first layer)
  static_assert(CN == 0 || (S == 1 && CN <= 4), "narrow input: one k-step, at most four 16-bit loads per lane");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int WVEC = NCP * 2 * K * S * 64;                 // 16-byte vectors of the weight slice
  u32x4* wl = reinterpret_cast<u32x4*>(smem);
  uint32_t* img_all = reinterpret_cast<uint32_t*>(smem + (size_t)WVEC * 16);
  float* stat = reinterpret_cast<float*>(smem + (size_t)WVEC * 16 + 8 * IMG_BYTES);      // [2][64 * NCP]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cp = wave8 % NCP, fwl = wave8 / NCP;
  const int b = blockIdx.x;
  const int slice = (b >> 3) % P.gy;
  const int grp = (b / (8 * P.gy)) * 8 + (b & 7);
  const int V = P.V;
  const int c = lane & 31, h = lane >> 5;
  const int cbase = (slice * NCP + cp) * 64;                 // first output channel of this wave

  // ---- setup: weight slice -> LDS (all loads of the copy in flight at once), adjacency and bias-term rows staged
  //      through the image region with coalesced loads, then the per-lane constants built from LDS.  (Built straight
  //      from global memory the compiler predicates each of the ~80 scalar loads and waits for them one by one.) ----
  {
    const u32x4* wg = reinterpret_cast<const u32x4*>(P.Wq) + (size_t)slice * WVEC;
    constexpr int NWI = (WVEC + RC_NTH - 1) / RC_NTH;
    u32x4 wv[NWI];
#pragma unroll
    for (int i = 0; i < NWI; ++i) wv[i] = wg[min(tid + i * RC_NTH, WVEC - 1)];
#pragma unroll
    for (int i = 0; i < NWI; ++i) if (tid + i * RC_NTH < WVEC) wl[tid + i * RC_NTH] = wv[i];
    for (int i = tid; i < 2 * 64 * NCP; i += RC_NTH) stat[i] = 0.f;
  }
  float* Asc = reinterpret_cast<float*>(img_all);            // [K][V][V]
  float* Bsc = Asc + K * 32 * 32;                            // [V][64 * NCP]: this workgroup's channels of bterm
  for (int i = tid; i < K * V * V; i += RC_NTH) Asc[i] = P.A[i];
  if (P.bterm) {
    for (int i = tid; i < V * 64 * NCP; i += RC_NTH) {
      const int w = i / (64 * NCP), cc = i - w * (64 * NCP);
      Bsc[i] = P.bterm[w * P.Cout + slice * NCP * 64 + cc];
    }
  }
  __syncthreads();
  u32x4 At[K][2];
#pragma unroll
  for (int k = 0; k < K; ++k) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      frag_t f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int v = 16 * s + 8 * (j >> 2) + 4 * h + (j & 3);
        f[j] = E::from_f((c < V && v < V) ? Asc[(k * V + v) * V + c] : 0.f);
      }
      At[k][s] = __builtin_bit_cast(u32x4, f);
    }
  }
  f32x16 bt[2];
#pragma unroll
  for (int jt = 0; jt < 2; ++jt) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int w = (i & 3) + 8 * (i >> 2) + 4 * h;
      bt[jt][i] = (P.bterm && w < V) ? Bsc[w * (64 * NCP) + cp * 64 + 32 * jt + c] : 0.f;
    }
  }
  __syncthreads();                                           // the scratch becomes the waves' output images

  const T* xg = reinterpret_cast<const T*>(P.x);
  T* yg = reinterpret_cast<T*>(P.y);
  const T* addg = reinterpret_cast<const T*>(P.addend);
  const unsigned xfrm_b = (unsigned)(V * CROW) * 2u, yfrm_b = (unsigned)(V * P.Cout) * 2u;  // bytes of one frame
  const u32x4* wlane = wl + (size_t)cp * (2 * K * S * 64) + lane;         // fragment (jt, k, s) at + ((jt*K + k)*S + s)*64
  uint32_t* img = img_all + wave8 * (IMG_BYTES / 4);
  // byte offset of this lane's row vector in a frame (narrow rows: lane half 0 holds the whole row, half 1 reads past
  // the descriptor = zeros)
  const unsigned xoff = CN ? (h == 0 ? (unsigned)(c * CN) * 2u : 0x7ffffff0u) : (unsigned)(c * CIN + 8 * h) * 2u;
  const size_t in_seq = (size_t)P.Tin * V * CROW, in_frm = (size_t)P.in_t_stride * V * CROW;
  const size_t out_seq = (size_t)P.Tout * V * P.Cout, out_frm = (size_t)P.out_t_stride * V * P.Cout;

  auto loadx = [&](int n, int t, u32x4 (&xf)[S]) __attribute__((always_inline)) {
    const rsrc_t r = make_rsrc(xg + (size_t)n * in_seq + (size_t)t * in_frm, xfrm_b);     // wave-uniform
    if constexpr (CN) {
      // 2 * CN-byte rows: element-wise 16-bit loads (only 2-byte aligned), packed into the one k-step's fragment
      uint32_t e[4] = {0u, 0u, 0u, 0u};
#pragma unroll
      for (int j = 0; j < CN; ++j) e[j] = (uint32_t)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(r, xoff + 2u * j, 0, 0);
      xf[0] = u32x4{e[0] | (e[1] << 16), e[2] | (e[3] << 16), 0u, 0u};
    } else {
#pragma unroll
      for (int s = 0; s < S; ++s) xf[s] = __builtin_amdgcn_raw_buffer_load_b128(r, xoff + 32u * s, 0, 0);
    }
  };

  float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
  // image slots this lane copies out: pair-row (lane>>3) + 8m, channel vector lane&7
  const int rp = lane >> 3, chunk = lane & 7;
  const unsigned yrow_b = (unsigned)P.Cout * 2u, yoff = (unsigned)(cbase + 8 * chunk) * 2u;

  // one frame: contraction + aggregation + epilogue.  `xf` holds the frame; with !PF2 the loads of (n2, t2) are issued
  // into `xf` itself once its last reader has been issued.
  auto frame = [&](int n, int t, u32x4 (&xf)[S], int n2, int t2) __attribute__((always_inline)) {
    f32x16 Y[2];
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) {
#pragma unroll
      for (int k = 0; k < K; ++k) {
        f32x16 H;
#pragma unroll
        for (int i = 0; i < 16; ++i) H[i] = 0.f;
#pragma unroll
        for (int s = 0; s < S; ++s) {
          const u32x4 wv = wlane[((jt * K + k) * S + s) * 64];
          mma_kgroup(H, __builtin_bit_cast(frag_t, xf[s]), __builtin_bit_cast(frag_t, wv));
        }
        if constexpr (!PF2) {
          if (jt == 1 && k == K - 1) loadx(n2, t2, xf);
        }
        u32x4 hb[2];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int q = 0; q < 4; ++q) hb[s][q] = pack2<T>(H[8 * s + 2 * q], H[8 * s + 2 * q + 1]);
        if (k == 0) {
          Y[jt] = bt[jt];
        }
        mma_kgroup(Y[jt], __builtin_bit_cast(frag_t, At[k][0]), __builtin_bit_cast(frag_t, hb[0]));
        mma_kgroup(Y[jt], __builtin_bit_cast(frag_t, At[k][1]), __builtin_bit_cast(frag_t, hb[1]));
      }
      // epilogue of this channel tile: round, BatchNorm sums, pair-row image (rows w >= V are exact zeros: A^T rows and
      // the bias-term rows there are zero)
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const uint32_t pk = pack2<T>(Y[jt][2 * q], Y[jt][2 * q + 1]);
        float lo, hi;
        unpack2<T>(pk, lo, hi);
        s1[jt] += lo + hi;
        s2[jt] = fmaf(lo, lo, s2[jt]);
        s2[jt] = fmaf(hi, hi, s2[jt]);
        const int p = (q & 1) + 4 * (q >> 1) + 2 * h;
        img[p * IMG_RS + 32 * jt + c] = pk;
      }
    }
    // image -> HBM: 8 lanes cover one 128-byte row of this wave's 64 channels
    const size_t yfo = (size_t)n * out_seq + (size_t)t * out_frm;
    const rsrc_t ry = make_rsrc(yg + yfo, yfrm_b);
    const rsrc_t ra = make_rsrc(ADD ? addg + yfo : yg + yfo, yfrm_b);
    u32x4 av[2][2];
    if constexpr (ADD) {
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        av[m][0] = __builtin_amdgcn_raw_buffer_load_b128(ra, yoff + (unsigned)(2 * (rp + 8 * m)) * yrow_b, 0, 0);
        av[m][1] = __builtin_amdgcn_raw_buffer_load_b128(ra, yoff + (unsigned)(2 * (rp + 8 * m) + 1) * yrow_b, 0, 0);
      }
    }
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const int pr = rp + 8 * m;
      const u32x4 u0 = *reinterpret_cast<const u32x4*>(img + pr * IMG_RS + 8 * chunk);
      const u32x4 u1 = *reinterpret_cast<const u32x4*>(img + pr * IMG_RS + 8 * chunk + 4);
      u32x4 ev, od;
      ev[0] = __builtin_amdgcn_perm(u0[1], u0[0], 0x05040100u); od[0] = __builtin_amdgcn_perm(u0[1], u0[0], 0x07060302u);
      ev[1] = __builtin_amdgcn_perm(u0[3], u0[2], 0x05040100u); od[1] = __builtin_amdgcn_perm(u0[3], u0[2], 0x07060302u);
      ev[2] = __builtin_amdgcn_perm(u1[1], u1[0], 0x05040100u); od[2] = __builtin_amdgcn_perm(u1[1], u1[0], 0x07060302u);
      ev[3] = __builtin_amdgcn_perm(u1[3], u1[2], 0x05040100u); od[3] = __builtin_amdgcn_perm(u1[3], u1[2], 0x07060302u);
      if constexpr (ADD) {
        const frag_t a0 = __builtin_bit_cast(frag_t, av[m][0]), a1 = __builtin_bit_cast(frag_t, av[m][1]);
        frag_t o0 = __builtin_bit_cast(frag_t, ev), o1 = __builtin_bit_cast(frag_t, od);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          o0[j] = E::from_f(E::to_f(o0[j]) + E::to_f(a0[j]));
          o1[j] = E::from_f(E::to_f(o1[j]) + E::to_f(a1[j]));
        }
        ev = __builtin_bit_cast(u32x4, o0);
        od = __builtin_bit_cast(u32x4, o1);
      }
      // rows 2 pr and 2 pr + 1; rows >= V lie outside the frame's descriptor: dropped by the bounds check
      __builtin_amdgcn_raw_buffer_store_b128(ev, ry, yoff + (unsigned)(2 * pr) * yrow_b, 0, 0);
      __builtin_amdgcn_raw_buffer_store_b128(od, ry, yoff + (unsigned)(2 * pr + 1) * yrow_b, 0, 0);
    }
  };

  // ---- the walk over this worker's frames ----
  const int fw = grp * (8 / NCP) + fwl;
  int n = fw / P.Tlog, t = fw - n * P.Tlog;
  auto next = [&](int& nn, int& tt) __attribute__((always_inline)) {
    nn += P.step_n;
    tt += P.step_t;
    if (tt >= P.Tlog) { tt -= P.Tlog; ++nn; }
  };
  if (n < P.NM) {
    u32x4 xa[S];
    loadx(n, t, xa);
    if constexpr (PF2) {
      u32x4 xb[S];
      for (;;) {
        int n2 = n, t2 = t;
        next(n2, t2);
        const bool more = n2 < P.NM;
        loadx(more ? n2 : n, more ? t2 : t, xb);
        __builtin_amdgcn_sched_barrier(0);             // the prefetch is issued BEFORE the frame's work, not sunk into it
        frame(n, t, xa, n2, t2);
        if (!more) break;
        n = n2; t = t2;
        next(n2, t2);
        const bool more2 = n2 < P.NM;
        loadx(more2 ? n2 : n, more2 ? t2 : t, xa);
        __builtin_amdgcn_sched_barrier(0);
        frame(n, t, xb, n2, t2);
        if (!more2) break;
        n = n2; t = t2;
      }
    } else {
      for (;;) {
        int n2 = n, t2 = t;
        next(n2, t2);
        const bool more = n2 < P.NM;
        frame(n, t, xa, more ? n2 : n, more ? t2 : t);
        if (!more) break;
        n = n2; t = t2;
      }
    }
  }

  if (P.stats) {
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) {
      const float a = s1[jt] + __shfl_xor(s1[jt], 32), q = s2[jt] + __shfl_xor(s2[jt], 32);
      if (h == 0) {
        atomicAdd(&stat[cp * 64 + 32 * jt + c], a);
        atomicAdd(&stat[64 * NCP + cp * 64 + 32 * jt + c], q);
      }
    }
    __syncthreads();
    double* dst = P.stats + (size_t)(b % P.stats_rep) * 2 * P.Cout + slice * NCP * 64;
    for (int i = tid; i < 64 * NCP; i += RC_NTH) {
      atomic_add_f64(dst + i, (double)stat[i]);
      atomic_add_f64(dst + P.Cout + i, (double)stat[64 * NCP + i]);
    }
  }
  bn_tail_run(P.tail, gridDim.x, reinterpret_cast<unsigned*>(smem));
}

template <typename T, int S, int K, int NCP, bool PF2, bool ADD, int CN>
int rc_fwd_launch2(RcFwdParams P, int grid_cap, hipStream_t stream) {
  auto kfn = gcn_rc_fwd_kernel<T, S, K, NCP, PF2, ADD, CN>;
  const size_t lds = (size_t)NCP * 2 * K * S * 64 * 16 + 8 * IMG_BYTES + 2 * 64 * NCP * 4;
  if (lds > 160 * 1024) return ISTGCN_EINVAL;
  static std::atomic<unsigned long long> optin{0};
  if (int ea = istgcn_lds_optin((const void*)kfn, optin)) return ea;
  int res = grid_cap > 0 ? grid_cap : istgcn_resident_blocks((const void*)kfn, RC_NTH, lds);
  P.gy = P.Cout / (64 * NCP);
  int G = res / P.gy / 8 * 8;                    // groups: a multiple of 8 so that the slices of a group share an XCD
  if (G < 8) G = 8;
  const long long frames = (long long)P.NM * P.Tlog;
  const int fwpg = 8 / NCP;
  while (G > 8 && (long long)(G - 8) * fwpg >= frames) G -= 8;
  P.nfw = G * fwpg;
  P.step_n = P.nfw / P.Tlog;
  P.step_t = P.nfw % P.Tlog;
  ISTGCN_LAUNCH(kfn, dim3(G * P.gy), dim3(RC_NTH), lds, stream, P);
  ISTGCN_CHECK_LAUNCH();
  return ISTGCN_OK;
}

template <typename T, int S, int K, int NCP, bool PF2, int CN>
int rc_fwd_launch(const RcFwdParams& P, int grid_cap, hipStream_t stream) {
  if constexpr ((size_t)K * 16 * S * 64 * NCP * 2 > 100 * 1024) return ISTGCN_EINVAL;
  else {
    if constexpr (CN == 0) {
      if (P.addend) return rc_fwd_launch2<T, S, K, NCP, PF2, true, 0>(P, grid_cap, stream);
    } else if (P.addend) return ISTGCN_EINVAL;
    return rc_fwd_launch2<T, S, K, NCP, PF2, false, CN>(P, grid_cap, stream);
  }
}

template <typename T, int S, int K, int CN = 0>
int rc_fwd_ncp(const RcFwdParams& P, int grid_cap, hipStream_t stream) {
  // channel pairs per workgroup: the slice's weights (K * Cin * 64 * NCP 16-bit elements) must fit LDS next to the images
  constexpr bool two = (size_t)K * 16 * S * 128 * 2 <= 100 * 1024;
  if constexpr (two) {
    if (P.Cout % 128 == 0) return rc_fwd_launch<T, S, K, 2, (S <= 8), CN>(P, grid_cap, stream);
  }
  return rc_fwd_launch<T, S, K, 1, (S <= 8), CN>(P, grid_cap, stream);
}

template <typename T, int S, int CN = 0>
int rc_fwd_k(const RcFwdParams& P, int K, int grid_cap, hipStream_t stream) {
  switch (K) {
    case 1: return rc_fwd_ncp<T, S, 1, CN>(P, grid_cap, stream);
    case 2: return rc_fwd_ncp<T, S, 2, CN>(P, grid_cap, stream);
    case 3: return rc_fwd_ncp<T, S, 3, CN>(P, grid_cap, stream);
    case 4: return rc_fwd_ncp<T, S, 4, CN>(P, grid_cap, stream);
  }
  return ISTGCN_EINVAL;
}

template <typename T>
int rc_fwd_T(const RcFwdParams& P, int Cin, int K, int grid_cap, hipStream_t stream) {
  switch (Cin) {
    case 3: return rc_fwd_k<T, 1, 3>(P, K, grid_cap, stream);     // the models' first layer (net/st_gcnold.py:44: in_channels = 3)
    case 64: return rc_fwd_k<T, 4>(P, K, grid_cap, stream);
    case 128: return rc_fwd_k<T, 8>(P, K, grid_cap, stream);
    case 256: return rc_fwd_k<T, 16>(P, K, grid_cap, stream);
  }
  return ISTGCN_EINVAL;
}

}  // namespace

// Does the packed graph-conv weight buffer of (Cin, Cout, K, dtype) carry the register-chained layout behind the
// round-2 one?  (istgcn.h)
extern "C" int istgcn_gcn_rc_layout(int Cin, int Cout, int K, int dtype) {
  if (dtype == 0)      // float32 (gcn_rc_f32.hip): 64-channel chunks, all K partitions' tiles in registers
    return (Cin == 64 || Cin == 128 || Cin == 256) && Cout >= 64 && Cout % 64 == 0 && K >= 1 && K <= 3;
  if (dtype != 1 && dtype != 2) return 0;
  if (Cin != 3 && Cin != 64 && Cin != 128 && Cin != 256) return 0;
  if (Cout < 64 || Cout % 64 != 0 || K < 1 || K > 4) return 0;
  if ((size_t)K * Cin * 64 * 2 > 100 * 1024) return 0;
  return 1;
}

// Internal entry (called by istgcn_gcn_fwd's dispatch): Wq = the register-chained section of the packed weights.
extern "C" int istgcn_gcn_fwd_rc(const void* x, const float* A, const void* Wq, const float* bterm, const void* addend,
                                 void* y, double* stats, int stats_rep, int NM, int Tin, int Tout, int Tlog, int V,
                                 int Cin, int Cout, int K, int in_t_stride, int out_t_stride, int dtype, int grid_cap,
                                 void* stream) {
  if (!istgcn_gcn_rc_layout(Cin, Cout, K, dtype) || V > 32 || (stats && addend) || (Cin == 3 && addend)) return ISTGCN_EINVAL;
  RcFwdParams P{};
  P.x = x; P.Wq = Wq; P.A = A; P.bterm = bterm; P.addend = addend; P.y = y; P.stats = stats;
  P.NM = NM; P.Tin = Tin; P.Tout = Tout; P.Tlog = Tlog; P.V = V; P.Cout = Cout;
  P.in_t_stride = in_t_stride; P.out_t_stride = out_t_stride; P.stats_rep = stats_rep < 1 ? 1 : stats_rep;
  if (stats) istgcn_bn_tail_take(stats, &P.tail);
  if (dtype == 1) return rc_fwd_T<__bf16>(P, Cin, K, grid_cap, (hipStream_t)stream);
  return rc_fwd_T<_Float16>(P, Cin, K, grid_cap, (hipStream_t)stream);
}
