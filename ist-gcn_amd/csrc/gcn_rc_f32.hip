// Graph-convolution unit, forward, register-chained, FLOAT32 storage (the parity mode and the reference's own arithmetic
// for BASELINE configs 1 / 3 / 4).  Scheme of gcn_rc.hip with the fp32 matrix instruction v_mfma_f32_32x32x2_f32
// (exact fp32 products, fp32 accumulation, 64 FLOP/clk/SIMD):
//
//   H_k[v][c] = sum_i x[v][i] W_k[c][i]      A operand: lane (v, h) holds x[v][64q + 32h + s] for step s of chunk q -- 32
//                                            consecutive floats of its row, loaded STRAIGHT from HBM; B operand from LDS
//   Y[w][c]  += sum_v A_k[v][w] H_k[v][c]    the accumulator tile of step 1 is the B operand as it stands: register i of
//                                            lane half h is row (i&3) + 8(i>>2) + 4h, exactly the (k = h) pair one
//                                            32x32x2 step contracts -- no conversion, no data movement; A operand =
//                                            A_k^T[w][that row], 16 constants per partition and lane (kept in LDS)
// = net/utils/tgcn.py:79 (1x1 Conv2d) then :86 (einsum), variants folded into one A by the host.
//
// Round 2's fp32 path aggregated on the VALU from sparse lists (gcn_fwd_small.hip: 532 us for the 64 -> 64 layer at
// NM = 128, T = 300 -- VERDICT r2 #7); dense aggregation on the matrix cores is 96 of this kernel's 288 instructions per
// frame and 64-channel pair, at the fp32 MFMA rate.  All K*NT H tiles of a frame are accumulated before the aggregation so
// that every chunk of x is loaded exactly once (wide layers: 64-channel chunks, double-buffered); the workgroup's weight
// slice (NT 32-channel tiles x all input channels, <= 98 KB) sits in LDS; each wave owns whole frames; no barrier in the loop.
// The output tile has the channel on the lane: 32 consecutive floats per row and register -> stored as it stands.
#include "gcn_rc.hpp"

extern "C" int istgcn_gcn_rc_layout(int Cin, int Cout, int K, int dtype);
extern "C" int istgcn_wgrad_reduce(const float* ws, long long slice, int nsl, float* d0, int n0, float* d1, int n1, void* stream);

namespace {

struct RcF32Params {
  const float* x; const float* Wq; const float* A; const float* bterm; float* y; double* stats;
  int NM, Tin, Tout, Tlog, V, Cin, Cout, in_t_stride, out_t_stride, stats_rep;
  int nfw, step_n, step_t, gy;
};

// NQ = Cin / 64 (chunks of the channel contraction), NT = 32-channel output tiles per wave (and per workgroup slice)
template <int NQ, int K, int NT>
__global__ __launch_bounds__(RC_NTH, 2) void gcn_rc_f32_fwd_kernel(const RcF32Params P) {
  constexpr int CIN = 64 * NQ;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int WVEC = NT * K * NQ * 8 * 64;                  // 16-byte vectors of the weight slice: [jt][k][q][s4][lane]
  f32x4* wl = reinterpret_cast<f32x4*>(smem);
  f32x4* atl = wl + WVEC;                                     // [K][4][64]: A_k^T constants, 4 steps per vector
  f32x4* btl = atl + K * 4 * 64;                              // [NT][4][64]: bias-term rows of this slice in accumulator order
  float* stat = reinterpret_cast<float*>(btl + NT * 4 * 64);  // [2][32 * NT]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x;
  const int slice = (b >> 3) % P.gy;
  const int grp = (b / (8 * P.gy)) * 8 + (b & 7);
  const int V = P.V;
  const int c = lane & 31, h = lane >> 5;
  const int cbase = slice * NT * 32;

  // ---- setup ----
  {
    const f32x4* wg = reinterpret_cast<const f32x4*>(P.Wq) + (size_t)slice * WVEC;
    constexpr int NWI = WVEC / RC_NTH;                        // WVEC is a multiple of 512
    for (int i0 = 0; i0 < NWI; i0 += 8) {
      f32x4 wv[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) wv[i] = wg[tid + min(i0 + i, NWI - 1) * RC_NTH];
#pragma unroll
      for (int i = 0; i < 8; ++i) if (i0 + i < NWI) wl[tid + (i0 + i) * RC_NTH] = wv[i];
    }
    // A operand of the aggregation: lane (w, hh), step i holds A_k[v = (i&3) + 8(i>>2) + 4hh][w]
    for (int idx = tid; idx < K * 4 * 64; idx += RC_NTH) {
      const int ln = idx & 63, g4 = (idx >> 6) & 3, k = idx >> 8;
      const int w = ln & 31, hh = ln >> 5;
      f32x4 a;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int v = e + 8 * g4 + 4 * hh;                    // step i = 4 g4 + e
        a[e] = (w < V && v < V) ? P.A[(k * V + min(v, V - 1)) * V + min(w, V - 1)] : 0.f;
      }
      atl[idx] = a;
    }
    for (int idx = tid; idx < NT * 4 * 64; idx += RC_NTH) {
      const int ln = idx & 63, g4 = (idx >> 6) & 3, jt = idx >> 8;
      const int cc = ln & 31, hh = ln >> 5;
      f32x4 a;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int w = e + 8 * g4 + 4 * hh;
        a[e] = (P.bterm && w < V) ? P.bterm[min(w, V - 1) * P.Cout + cbase + 32 * jt + cc] : 0.f;
      }
      btl[idx] = a;
    }
    for (int i = tid; i < 2 * 32 * NT; i += RC_NTH) stat[i] = 0.f;
  }
  __syncthreads();

  const unsigned xfrm_b = (unsigned)(V * CIN) * 4u, yfrm_b = (unsigned)(V * P.Cout) * 4u;
  const unsigned xoff = (unsigned)(c * CIN + 32 * h) * 4u;    // this lane's 32 floats of chunk 0 of its row
  const size_t in_seq = (size_t)P.Tin * V * CIN, in_frm = (size_t)P.in_t_stride * V * CIN;
  const size_t out_seq = (size_t)P.Tout * V * P.Cout, out_frm = (size_t)P.out_t_stride * V * P.Cout;

  auto loadq = [&](int n, int t, int q, f32x4 (&xf)[8]) __attribute__((always_inline)) {
    const rsrc_t r = make_rsrc(P.x + (size_t)n * in_seq + (size_t)t * in_frm, xfrm_b);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, xoff + (unsigned)q * 256u + 16u * j, 0, 0);
      xf[j] = __builtin_bit_cast(f32x4, v);
    }
  };

  float s1[NT], s2[NT];
#pragma unroll
  for (int jt = 0; jt < NT; ++jt) { s1[jt] = 0.f; s2[jt] = 0.f; }
  const f32x4* wlane = wl + lane;

  // one frame; `xc` holds chunk 0 on entry and chunk 0 of frame (n2, t2) on exit
  auto frame = [&](int n, int t, f32x4 (&xc)[8], f32x4 (&xn)[8], int n2, int t2) __attribute__((always_inline)) {
    f32x16 H[K][NT];
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
      for (int jt = 0; jt < NT; ++jt)
#pragma unroll
        for (int i = 0; i < 16; ++i) H[k][jt][i] = 0.f;
    // One group = one 16-byte weight vector (4 steps) = 4 MFMAs of 64 cycles; the next group's vector is read one group
    // ahead and the order is pinned: left to itself the scheduler hoists all 24 reads of a chunk (96 registers) and spills.
    auto gemm = [&](int q, const f32x4 (&xq)[8]) __attribute__((always_inline)) {
      constexpr int NG = K * NT * 8;
      f32x4 wv = wlane[((0 * NQ + q) * 8) * 64];
#pragma unroll
      for (int gi = 0; gi < NG; ++gi) {
        const int s4 = gi & 7, kj = gi >> 3, k = kj / NT, jt = kj - k * NT;
        const int gn = gi + 1 < NG ? gi + 1 : gi;
        const int s4n = gn & 7, kjn = gn >> 3, kn = kjn / NT, jtn = kjn - kn * NT;
        const f32x4 wn = wlane[(((jtn * K + kn) * NQ + q) * 8 + s4n) * 64];
#pragma unroll
        for (int e = 0; e < 4; ++e)
          H[k][jt] = __builtin_amdgcn_mfma_f32_32x32x2f32(xq[s4][e], wv[e], H[k][jt], 0, 0, 0);
        wv = wn;
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    // chunks alternate between the two register sets; the load of the next chunk (or of the next frame's chunk 0) is
    // issued before the matrix work of the current one
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      if (q & 1) {
        if (q + 1 < NQ) loadq(n, t, q + 1, xc); else loadq(n2, t2, 0, xc);
        gemm(q, xn);
      } else {
        if (q + 1 < NQ) loadq(n, t, q + 1, xn); else loadq(n2, t2, 0, xn);
        gemm(q, xc);
      }
    }
    // aggregation + epilogue per output tile
    const rsrc_t ry = make_rsrc(P.y + (size_t)n * out_seq + (size_t)t * out_frm, yfrm_b);
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) {
      f32x16 Y;
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const f32x4 bv = btl[(jt * 4 + g4) * 64 + lane];
#pragma unroll
        for (int e = 0; e < 4; ++e) Y[4 * g4 + e] = bv[e];
      }
      {
        f32x4 av = atl[lane];
#pragma unroll
        for (int gi = 0; gi < K * 4; ++gi) {
          const int k = gi >> 2, g4 = gi & 3;
          const f32x4 an = atl[(gi + 1 < K * 4 ? gi + 1 : gi) * 64 + lane];
#pragma unroll
          for (int e = 0; e < 4; ++e)
            Y = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], H[k][jt][4 * g4 + e], Y, 0, 0, 0);
          av = an;
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      // rows w >= V are exact zeros (A^T rows and bias-term rows there are zero); stores of those rows fall outside the
      // frame's descriptor and are dropped
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int w = (i & 3) + 8 * (i >> 2) + 4 * h;
        const float yi = Y[i];            // (a bit cast applied to the vector element itself is compiled as element 0)
        s1[jt] += yi;
        s2[jt] = fmaf(yi, yi, s2[jt]);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(yi), ry, (unsigned)(w * P.Cout + cbase + 32 * jt + c) * 4u, 0, 0);
      }
    }
  };

  // ---- the walk over this worker's frames (NQ even: chunk 0 of the next frame lands in the set the frame started with;
  //      NQ odd: in the other one) ----
  const int fw = grp * 8 + wave8;
  int n = fw / P.Tlog, t = fw - n * P.Tlog;
  auto next = [&](int& nn, int& tt) __attribute__((always_inline)) {
    nn += P.step_n;
    tt += P.step_t;
    if (tt >= P.Tlog) { tt -= P.Tlog; ++nn; }
  };
  if (n < P.NM) {
    f32x4 xa[8], xb[8];
    loadq(n, t, 0, xa);
    for (;;) {
      int n2 = n, t2 = t;
      next(n2, t2);
      const bool more = n2 < P.NM;
      frame(n, t, xa, xb, more ? n2 : n, more ? t2 : t);
      if (!more) break;
      n = n2; t = t2;
      if constexpr (NQ & 1) {
        next(n2, t2);
        const bool more2 = n2 < P.NM;
        frame(n, t, xb, xa, more2 ? n2 : n, more2 ? t2 : t);
        if (!more2) break;
        n = n2; t = t2;
      }
    }
  }

  if (P.stats) {
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) {
      const float a = s1[jt] + __shfl_xor(s1[jt], 32), q = s2[jt] + __shfl_xor(s2[jt], 32);
      if (h == 0) {
        atomicAdd(&stat[32 * jt + c], a);
        atomicAdd(&stat[32 * NT + 32 * jt + c], q);
      }
    }
    __syncthreads();
    double* dst = P.stats + (size_t)(b % P.stats_rep) * 2 * P.Cout + cbase;
    for (int i = tid; i < 32 * NT; i += RC_NTH) {
      atomic_add_f64(dst + i, (double)stat[i]);
      atomic_add_f64(dst + P.Cout + i, (double)stat[32 * NT + i]);
    }
  }
}

template <int NQ, int K, int NT>
int rc_f32_launch(RcF32Params P, int grid_cap, hipStream_t stream) {
  auto kfn = gcn_rc_f32_fwd_kernel<NQ, K, NT>;
  const size_t lds = (size_t)NT * K * NQ * 8 * 64 * 16 + (size_t)(K + NT) * 4 * 64 * 16 + 2 * 32 * NT * 4;
  if (lds > 160 * 1024) return ISTGCN_EINVAL;
  static std::atomic<unsigned long long> optin{0};
  if (int ea = istgcn_lds_optin((const void*)kfn, optin)) return ea;
  int res = grid_cap > 0 ? grid_cap : istgcn_resident_blocks((const void*)kfn, RC_NTH, lds);
  P.gy = P.Cout / (32 * NT);
  int G = res / P.gy / 8 * 8;
  if (G < 8) G = 8;
  const long long frames = (long long)P.NM * P.Tlog;
  while (G > 8 && (long long)(G - 8) * 8 >= frames) G -= 8;
  P.nfw = G * 8;
  P.step_n = P.nfw / P.Tlog;
  P.step_t = P.nfw % P.Tlog;
  ISTGCN_LAUNCH(kfn, dim3(G * P.gy), dim3(RC_NTH), lds, stream, P);
  ISTGCN_CHECK_LAUNCH();
  return ISTGCN_OK;
}

template <int K>
int rc_f32_k(const RcF32Params& P, int grid_cap, hipStream_t stream) {
  switch (P.Cin) {
    case 64: return rc_f32_launch<1, K, 2>(P, grid_cap, stream);
    case 128: return rc_f32_launch<2, K, 2>(P, grid_cap, stream);
    case 256: return rc_f32_launch<4, K, 1>(P, grid_cap, stream);
  }
  return ISTGCN_EINVAL;
}

}  // namespace

// float32 section of the packed weights: element ((((jt*K + k)*(Cin/64) + q)*8 + s4)*64 + 32*h + c)*4 + e holds
// Wr[32*jt + c][k][64*q + 32*h + 4*s4 + e]  (istgcn.h)
extern "C" int istgcn_gcn_fwd_rc_f32(const void* x, const float* A, const void* Wq, const float* bterm, void* y, double* stats,
                                     int stats_rep, int NM, int Tin, int Tout, int Tlog, int V, int Cin, int Cout, int K,
                                     int in_t_stride, int out_t_stride, int grid_cap, void* stream) {
  if (!istgcn_gcn_rc_layout(Cin, Cout, K, 0) || V > 32) return ISTGCN_EINVAL;
  RcF32Params P{};
  P.x = reinterpret_cast<const float*>(x); P.Wq = reinterpret_cast<const float*>(Wq); P.A = A; P.bterm = bterm;
  P.y = reinterpret_cast<float*>(y); P.stats = stats;
  P.NM = NM; P.Tin = Tin; P.Tout = Tout; P.Tlog = Tlog; P.V = V; P.Cin = Cin; P.Cout = Cout;
  P.in_t_stride = in_t_stride; P.out_t_stride = out_t_stride; P.stats_rep = stats_rep < 1 ? 1 : stats_rep;
  switch (K) {
    case 1: return rc_f32_k<1>(P, grid_cap, (hipStream_t)stream);
    case 2: return rc_f32_k<2>(P, grid_cap, (hipStream_t)stream);
    case 3: return rc_f32_k<3>(P, grid_cap, (hipStream_t)stream);
  }
  return ISTGCN_EINVAL;
}

// =====================================================================================================================
// Weight gradient, float32 (autograd of net/utils/tgcn.py:79-86):
//   dW[k][c][i] += sum_{n,t,w} dy[t,w,c] * xa_k[t,w,i],  xa_k[w][i] = sum_v A_k[v][w] x[v][i];   S[w][c] += sum dy[t,w,c]
// A wave owns one 32-input-channel tile of the workgroup's (64 output, 64 input) channel block for BOTH output tiles and
// all K partitions (the aggregated tile is computed once per frame and partition) and walks every fourth frame of a batch:
//   XA_k = A_k^T . x_frame      16 steps of v_mfma_f32_32x32x2_f32; per-lane operands are single floats, so the frame
//                               images in LDS are read row-wise (ds_read_b32, conflict-free) -- no transposed reads in fp32
//   dW_k[ct] += dy_frame^T . XA_k   the accumulator tile XA_k is the B operand as it stands (gcn_rc_f32 forward, step 2)
//   S accumulates the dy values the lanes already hold (VALU).
// Round 2's fp32 path (tconv_wgrad_kernel<..AGG..>) aggregates on the VALU into LDS images first.
// =====================================================================================================================
namespace {

struct RcF32WgParams {
  const float* dy; const float* x; const float* A; float* dW; float* S; float* ws;
  long long ws_slice;
  int NM, T, V, Cin, Cout, K;
  int G, gy, nib;
};

constexpr int F32_RS = 64;                  // floats per image row
constexpr int F32_IMG = 32 * F32_RS;        // floats per frame image (32 rows; rows >= V are zeros)

template <int K>
__global__ __launch_bounds__(RC_NTH, 2) void gcn_rc_f32_wgrad_kernel(const RcF32WgParams P) {
  constexpr int FB = 4;                                       // frames per batch = frame groups
  constexpr int BUF = FB * 2 * F32_IMG;                       // floats per buffer: [frame][x | dy]
  constexpr int NIT = FB * 2 * 32 * 16 / RC_NTH;              // staging slots per thread (16-byte vectors)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* lds = reinterpret_cast<float*>(smem);
  f32x4* atl = reinterpret_cast<f32x4*>(lds + 2 * BUF);       // [K][4][64] A_k^T constants: lane (w, h), step s: A_k[16h + s][w]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int it = wave8 & 1, fg = wave8 >> 1;
  const int b = blockIdx.x;
  const int blk = (b >> 3) % P.gy;
  const int grp = (b / (8 * P.gy)) * 8 + (b & 7);
  const int ib = blk % P.nib, cb = blk / P.nib;
  const int V = P.V, Cin = P.Cin, Cout = P.Cout;
  const int c = lane & 31, h = lane >> 5;

  for (int i = tid; i < 2 * BUF / 4; i += RC_NTH) reinterpret_cast<f32x4*>(lds)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int idx = tid; idx < K * 4 * 64; idx += RC_NTH) {
    const int ln = idx & 63, g4 = (idx >> 6) & 3, k = idx >> 8;
    const int w = ln & 31, hh = ln >> 5;
    f32x4 a;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int v = 16 * hh + 4 * g4 + e;                     // step s = 4 g4 + e contracts joints s and 16 + s
      a[e] = (w < V && v < V) ? P.A[(k * V + min(v, V - 1)) * V + min(w, V - 1)] : 0.f;
    }
    atl[idx] = a;
  }

  // staging slots: item = (frame of the batch, tensor, row, 16-byte vector of the 64-channel slice)
  unsigned goff[NIT];
  int loff[NIT], gsel[NIT];
#pragma unroll
  for (int j = 0; j < NIT; ++j) {
    const int idx = tid + j * RC_NTH;
    const int q = idx & 15, v = (idx >> 4) & 31, ten = (idx >> 9) & 1, f = idx >> 10;
    const int C = ten ? Cout : Cin, c0 = ten ? cb * 64 : ib * 64;
    goff[j] = v < V ? (unsigned)(((f * V + v) * C + c0 + 4 * q) * 4) : 0x7ffffff0u;
    loff[j] = (f * 2 + ten) * F32_IMG + v * F32_RS + 4 * q;
    gsel[j] = ten;
  }
  const long long F = (long long)P.NM * P.T;
  const long long NB = (F + FB - 1) / FB;
  f32x4 rg[NIT];
  auto issue = [&](long long bt) __attribute__((always_inline)) {
    const long long g0 = bt * FB;
    const int nfr = (int)min((long long)FB, F - g0);
    const rsrc_t r0 = make_rsrc(P.x + g0 * V * Cin, (unsigned)(nfr * V * Cin) * 4u);
    const rsrc_t r1 = make_rsrc(P.dy + g0 * V * Cout, (unsigned)(nfr * V * Cout) * 4u);
#pragma unroll
    for (int j = 0; j < NIT; ++j) {
      // (the tensor of a slot is a per-thread constant: idx bit 9 -- 512 threads -> it alternates with j)
      const u32x4 v = (j & 1) ? __builtin_amdgcn_raw_buffer_load_b128(r1, goff[j], 0, 0)
                              : __builtin_amdgcn_raw_buffer_load_b128(r0, goff[j], 0, 0);
      rg[j] = __builtin_bit_cast(f32x4, v);
    }
  };
  auto commit = [&](int half) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < NIT; ++j) *reinterpret_cast<f32x4*>(lds + half * BUF + loff[j]) = rg[j];
  };

  f32x16 acc[K][2];
  float Sacc[2][16];
#pragma unroll
  for (int k = 0; k < K; ++k)
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[k][ct][i] = 0.f;
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int i = 0; i < 16; ++i) Sacc[ct][i] = 0.f;

  auto frame = [&](const float* buf) __attribute__((always_inline)) {
    const float* xi = buf + fg * 2 * F32_IMG;                 // this group's frame: x image, then dy image
    const float* di = xi + F32_IMG;
    float xv[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) xv[s] = xi[(16 * h + s) * F32_RS + 32 * it + c];              // B operand of XA: x[16h + s][i]
#pragma unroll
    for (int k = 0; k < K; ++k) {
      f32x16 XA;
#pragma unroll
      for (int i = 0; i < 16; ++i) XA[i] = 0.f;
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const f32x4 av = atl[(k * 4 + g4) * 64 + lane];
#pragma unroll
        for (int e = 0; e < 4; ++e) XA = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], xv[4 * g4 + e], XA, 0, 0, 0);
      }
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        // A operand: dy[row of XA register i][c] -- re-read per partition (one ds_read_b32 per 64-cycle MFMA; holding both
        // tiles' 32 values across the K partitions spilled)
        float dv[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) dv[i] = di[((i & 3) + 8 * (i >> 2) + 4 * h) * F32_RS + 32 * ct + c];
        if (k == 0 && it == 0) {
#pragma unroll
          for (int i = 0; i < 16; ++i) Sacc[ct][i] += dv[i];
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const float xa = XA[i];
          acc[k][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(dv[i], xa, acc[k][ct], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };

  long long bt = grp;
  __syncthreads();
  if (bt < NB) {
    issue(bt);
    commit(0);
  }
  __syncthreads();
  int half = 0;
  for (; bt < NB; bt += P.G) {
    const long long nxt = bt + P.G;
    issue(nxt < NB ? nxt : bt);
    __builtin_amdgcn_sched_barrier(0);
    frame(lds + half * BUF);
    commit(half ^ 1);
    half ^= 1;
    __syncthreads();
  }

  // ---- flush: the four frame groups' sums are combined through LDS in two rounds, group 0 writes ----
  float* red = lds;                                           // [2 waves][K*2*16 + 32][64]
  constexpr int NR = K * 2 * 16 + 32;
  auto put = [&](int slot) __attribute__((always_inline)) {
    float* r = red + (size_t)slot * NR * 64 + lane;
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int i = 0; i < 16; ++i) r[((k * 2 + ct) * 16 + i) * 64] = acc[k][ct][i];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int i = 0; i < 16; ++i) r[(K * 32 + ct * 16 + i) * 64] = Sacc[ct][i];
  };
  auto get = [&](int slot) __attribute__((always_inline)) {
    const float* r = red + (size_t)slot * NR * 64 + lane;
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[k][ct][i] += r[((k * 2 + ct) * 16 + i) * 64];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int i = 0; i < 16; ++i) Sacc[ct][i] += r[(K * 32 + ct * 16 + i) * 64];
  };
  if (fg >= 2) put((fg - 2) * 2 + it);
  __syncthreads();
  if (fg < 2) get(fg * 2 + it);
  __syncthreads();
  if (fg == 1) put(it);
  __syncthreads();
  if (fg == 0) {
    get(it);
    const int n0 = K * Cout * Cin;
    const int icol = ib * 64 + 32 * it + c;
    float* dW = P.ws ? P.ws + (size_t)grp * P.ws_slice : P.dW;
    float* S = P.ws ? P.ws + (size_t)grp * P.ws_slice + n0 : P.S;
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int r = cb * 64 + 32 * ct + (i & 3) + 8 * (i >> 2) + 4 * h;
          float* p = dW + ((size_t)k * Cout + r) * Cin + icol;
          if (P.ws) *p = acc[k][ct][i]; else atomicAdd(p, acc[k][ct][i]);
        }
    if (it == 0 && ib == 0 && P.S) {
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int w = (i & 3) + 8 * (i >> 2) + 4 * h;
          if (w < V) {
            float* p = S + w * Cout + cb * 64 + 32 * ct + c;
            if (P.ws) *p = Sacc[ct][i]; else atomicAdd(p, Sacc[ct][i]);
          }
        }
    }
  }
}

template <int K>
int rc_f32_wg_launch(RcF32WgParams P, int grid_cap, hipStream_t stream) {
  auto kfn = gcn_rc_f32_wgrad_kernel<K>;
  size_t lds = (size_t)2 * 4 * 2 * F32_IMG * 4 + (size_t)K * 4 * 64 * 16;
  const size_t redb = (size_t)4 * (K * 32 + 32) * 64 * 4;
  if (redb > lds) lds = redb;
  if (lds > 160 * 1024) return ISTGCN_EINVAL;
  static std::atomic<unsigned long long> optin{0};
  if (int ea = istgcn_lds_optin((const void*)kfn, optin)) return ea;
  int res = grid_cap > 0 ? grid_cap : istgcn_resident_blocks((const void*)kfn, RC_NTH, lds);
  P.nib = P.Cin / 64;
  P.gy = P.nib * (P.Cout / 64);
  int G = res / P.gy / 8 * 8;
  if (G < 8) G = 8;
  const long long NB = ((long long)P.NM * P.T + 3) / 4;
  while (G > 8 && G - 8 >= NB) G -= 8;
  P.G = G;
  const long long n0 = (long long)K * P.Cout * P.Cin, n1 = P.S ? (long long)P.V * P.Cout : 0;
  const bool use_ws = P.ws && (long long)G * (n0 + n1) <= P.ws_slice && G >= 32;
  if (use_ws) P.ws_slice = n0 + n1; else P.ws = nullptr;
  ISTGCN_LAUNCH(kfn, dim3(G * P.gy), dim3(RC_NTH), lds, stream, P);
  ISTGCN_CHECK_LAUNCH();
  if (use_ws) return istgcn_wgrad_reduce(P.ws, n0 + n1, G, P.dW, (int)n0, P.S, (int)n1, stream);
  return ISTGCN_OK;
}

}  // namespace

extern "C" int istgcn_gcn_wgrad_rc_f32(const void* dy, const void* x, const float* A, float* dW, float* S, int NM, int T,
                                       int V, int Cin, int Cout, int K, int grid_cap, float* ws, long long ws_floats,
                                       void* stream) {
  if (V > 32 || Cin < 64 || Cin % 64 || Cout < 64 || Cout % 64 || K < 1 || K > 3) return ISTGCN_EINVAL;
  RcF32WgParams P{};
  P.dy = reinterpret_cast<const float*>(dy); P.x = reinterpret_cast<const float*>(x); P.A = A; P.dW = dW; P.S = S;
  P.ws = ws_floats > 0 ? ws : nullptr; P.ws_slice = ws_floats;
  P.NM = NM; P.T = T; P.V = V; P.Cin = Cin; P.Cout = Cout; P.K = K;
  switch (K) {
    case 1: return rc_f32_wg_launch<1>(P, grid_cap, (hipStream_t)stream);
    case 2: return rc_f32_wg_launch<2>(P, grid_cap, (hipStream_t)stream);
    case 3: return rc_f32_wg_launch<3>(P, grid_cap, (hipStream_t)stream);
  }
  return ISTGCN_EINVAL;
}
