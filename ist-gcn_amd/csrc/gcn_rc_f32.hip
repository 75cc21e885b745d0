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
