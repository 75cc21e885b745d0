// Weight / bias gradient of the temporal (k,1) convolution, frame-tiled (round 3; 16-bit storage, 64-channel blocks).
//
//   dW[j][o][i] += sum_{n,m,v} dz[n,m,v,o] * u[n, s*m + tap_off[j], v, i],   u = relu(g*scale + shift) (the BatchNorm + ReLU
//   dbias[o]    += sum dz                                                     in front of the conv, applied on the way in)
// = autograd of nn.Conv2d(C, C, (9,1), (stride,1)) net/st_gcnold.py:167-173 and of the pre-summed 15-tap Inception-TCN
//   net/st_gcn_multi3_fix_3A_mstcn.py:160-180,212-215 (taps consecutive, stride 1 or 2).
//
// Why a second kernel: round 2's wave-specialised twg_ws holds 9 tap accumulators per compute wave and slides a halo
// window with immediate row offsets -- neither stretches to 15 taps or to a frame stride of 2, so the two stride-2
// layers of every model and ALL layers of the Inception-TCN models (configs 3/4/5) fell back to the round-1 kernel
// (443 us per stride-2 launch).  Here the tile is ONE FRAME (joints padded to 32 rows = two k-steps of a
// 32x32x16 MFMA): the contraction index is the joint, a tap is a choice of INPUT FRAME, so taps and strides are plain
// frame arithmetic -- no halo rows, no per-tap row offsets.
//   * a workgroup owns a (64 output, 64 input) channel block and walks SEGMENTS of consecutive output frames of one
//     sequence; its eight waves are the four (o-tile, i-tile) pairs x two tap groups of JT taps (JT = 5: up to 10 taps,
//     JT = 8: up to 16; taps beyond ntaps are computed against valid frames and never flushed);
//   * dz frames are staged per batch of FB output frames (double-buffered), transformed input frames stream through a
//     ring of frame images that holds the batch's window plus the frames prefetched for the next batch; all eight waves
//     stage (global -> registers during a whole batch -> LDS, 16-byte chunks XOR-swizzled: conflict-free transposed reads);
//   * per output frame a wave reads its dz^T fragments once and, per tap, the u^T fragments of that tap's input frame:
//     acc[tap] += dz_frame^T . u_frame (both operands through ds_read_b64_tr_b16, one step ahead of their MFMAs; K = 32
//     joints incl. 7 zero rows).  (A register ring that kept the u^T fragments of a wave's taps across output frames --
//     each read once instead of JT times -- measured the same: 136 vs 139 us at 64 channels; the kernel was bound by
//     bytes in flight, see the staging pipeline below.)
//   * frames outside [0, Tin) are zero images; the bias gradient is one more product against a constant of ones.
// Flush: per-workgroup partial sums to the workspace + the reduce kernel of tconv_wgrad.hip (atomics without one).
#include "gcn_rc.hpp"
#include <type_traits>

extern "C" int istgcn_wgrad_reduce(const float* ws, long long slice, int nsl, float* d0, int n0, float* d1, int n1, void* stream);

namespace {

struct TrcParams {
  const void* dz; const void* g; const float* pre; float* dW; float* dbias; float* ws;
  long long ws_slice;
  int NM, Tin, Tz, V, Cin, Cout, ntaps, tap0, pre_relu;
  int G, gy, nib;        // groups (grid-stride over segments), channel blocks per group, input-channel blocks
  int L, nsps, nseg;     // output frames per segment, segments per sequence, segments in all
};

constexpr int FRM = 32 * 64;              // elements of one frame image: 32 rows x 64 channels

// JT = taps per tap group, S = frame stride (1 | 2), FB = output frames per batch (power of two)
template <typename T, int JT, int S, int FB>
__global__ __launch_bounds__(RC_NTH, 2) void tconv_rc_wgrad_kernel(const TrcParams P) {
  using E = Elem<T>;
  typedef typename E::frag frag_t;
  constexpr int CH = S * FB;                                  // input frames per chunk of the u stream
  constexpr int LD = (2 * JT + CH - 1) / CH;                  // chunks the u stream runs ahead of the dz batches
  constexpr int UR = (LD + 2) * CH;                           // frames of the u ring
  constexpr int NITZ = FB * 32 * 8 / RC_NTH, NITU = CH * 32 * 8 / RC_NTH;     // staging slots per thread
  static_assert(NITZ >= 1 && NITU >= 1 && (FB & (FB - 1)) == 0, "batch size");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T* dzb = reinterpret_cast<T*>(smem);                        // [3][FB] frame images (a batch is staged two batches ahead)
  T* ub = dzb + 3 * FB * FRM;                                 // [UR] frame images
  float* bsum = reinterpret_cast<float*>(ub + UR * FRM);      // [64] bias-gradient column sums of this workgroup

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ot = wave8 & 1, it = (wave8 >> 1) & 1, tg = wave8 >> 2;
  const int b = blockIdx.x;
  const int blk = (b >> 3) % P.gy;
  const int grp = (b / (8 * P.gy)) * 8 + (b & 7);
  const int ib = blk % P.nib, ob = blk / P.nib;
  const int V = P.V, Cin = P.Cin, Cout = P.Cout;
  const int c = lane & 31, h = lane >> 5;
  const bool do_bias = ib == 0 && P.dbias != nullptr;

  // ---- setup ----
  for (int i = tid; i < (3 * FB + UR) * FRM / 8; i += RC_NTH) reinterpret_cast<u32x4*>(smem)[i] = u32x4{0u, 0u, 0u, 0u};
  if (tid < 64) bsum[tid] = 0.f;
  // staging slots: item = (frame of the batch / chunk, row, 16-byte channel vector); the vector index is the same for all
  // of a thread's items (512 % 8 == 0), so its BatchNorm coefficients live in registers
  const int sq = tid & 7;
  float sc[8], sh[8], dbs[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    sc[e] = P.pre ? P.pre[ib * 64 + 8 * sq + e] : 1.f;
    sh[e] = P.pre ? P.pre[Cin + ib * 64 + 8 * sq + e] : 0.f;
    dbs[e] = 0.f;
  }
  const bool relu = P.pre_relu != 0;
  int zrow[NITZ], zfrm[NITZ], urow[NITU], ufrm[NITU];
#pragma unroll
  for (int j = 0; j < NITZ; ++j) { const int idx = tid + j * RC_NTH; zrow[j] = (idx >> 3) & 31; zfrm[j] = idx >> 8; }
#pragma unroll
  for (int j = 0; j < NITU; ++j) { const int idx = tid + j * RC_NTH; urow[j] = (idx >> 3) & 31; ufrm[j] = idx >> 8; }
  auto img_off = [&](int v) __attribute__((always_inline)) { return v * 64 + ((sq ^ (4 * ((v >> 1) & 1))) * 8); };

  const T* dzg = reinterpret_cast<const T*>(P.dz);
  const T* gg = reinterpret_cast<const T*>(P.g);

  // ---- lane constants of the transposed reads ----
  const int qq = (lane & 15) >> 2, g1 = (lane >> 4) & 1;
  auto tr_lane = [&](int tile) __attribute__((always_inline)) {
    const int q = 4 * tile + 2 * g1 + ((lane & 3) >> 1);
    return (8 * h + qq) * 64 + ((q ^ (4 * ((qq >> 1) & 1))) * 8) + 4 * (lane & 1);
  };
  const int dlane = tr_lane(ot), ulane = tr_lane(it);

  f32x16 acc[JT];
#pragma unroll
  for (int j = 0; j < JT; ++j)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;

  constexpr bool DEEP = JT <= 5;           // two batches of loads in flight (the 16-tap variant has no registers for it)
  u32x4 rzA[NITZ], ruA[NITU], rzB[NITZ], ruB[NITU];
  for (int seg = grp; seg < P.nseg; seg += P.G) {
    const int n = seg / P.nsps, sl = seg - n * P.nsps;
    const int m0 = sl * P.L, m1 = min(P.Tz, m0 + P.L);
    const int Lseg = m1 - m0;
    const int f0 = S * m0 + P.tap0;                           // first input frame of the segment's u stream (may be < 0)
    const rsrc_t rg = make_rsrc(gg + (size_t)n * P.Tin * V * Cin, (unsigned)(P.Tin * V * Cin) * 2u);
    auto issue_z = [&](int bt, u32x4 (&rr)[NITZ]) __attribute__((always_inline)) {
      const int mb = m0 + bt * FB;
      const int nfr = max(0, min(FB, m1 - mb));
      const rsrc_t r = make_rsrc(dzg + ((size_t)n * P.Tz + min(mb, P.Tz - 1)) * V * Cout, (unsigned)(nfr * V * Cout) * 2u);
#pragma unroll
      for (int j = 0; j < NITZ; ++j) {
        const unsigned off = zrow[j] < V ? (unsigned)(((zfrm[j] * V + zrow[j]) * Cout + ob * 64 + 8 * sq) * 2) : 0x7ffffff0u;
        rr[j] = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
      }
    };
    auto commit_z = [&](int bt, u32x4 (&rr)[NITZ]) __attribute__((always_inline)) {
      T* dst = dzb + (bt % 3) * FB * FRM;
#pragma unroll
      for (int j = 0; j < NITZ; ++j) {
        *reinterpret_cast<u32x4*>(dst + zfrm[j] * FRM + img_off(zrow[j])) = rr[j];
        if (do_bias) {                                         // bias gradient: column sums of dz, summed as it is staged
          const frag_t v = __builtin_bit_cast(frag_t, rr[j]);
#pragma unroll
          for (int e = 0; e < 8; ++e) dbs[e] += E::to_f(v[e]);
        }
      }
    };
    auto issue_u = [&](int ck, u32x4 (&rr)[NITU]) __attribute__((always_inline)) {
#pragma unroll
      for (int j = 0; j < NITU; ++j) {
        const int f = f0 + ck * CH + ufrm[j];
        const bool ok = urow[j] < V && f >= 0 && f < P.Tin;
        const unsigned off = ok ? (unsigned)(((f * V + urow[j]) * Cin + ib * 64 + 8 * sq) * 2) : 0x7ffffff0u;
        rr[j] = __builtin_amdgcn_raw_buffer_load_b128(rg, off, 0, 0);
      }
    };
    auto commit_u = [&](int ck, u32x4 (&rr)[NITU]) __attribute__((always_inline)) {
#pragma unroll
      for (int j = 0; j < NITU; ++j) {
        const int fi = ck * CH + ufrm[j];                       // index in the segment's stream
        const int f = f0 + fi;
        const bool ok = urow[j] < V && f >= 0 && f < P.Tin;
        frag_t v = __builtin_bit_cast(frag_t, rr[j]);
        if (P.pre) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float x = E::to_f(v[e]) * sc[e] + sh[e];
            if (relu) x = fmaxf(x, 0.f);
            v[e] = E::from_f(ok ? x : 0.f);
          }
        }
        *reinterpret_cast<frag_t*>(ub + (fi % UR) * FRM + img_off(urow[j])) = v;
      }
    };
    // LDS -> operand registers
    auto read_dz = [&](int r, frag_t (&d)[2]) __attribute__((always_inline)) {        // dz^T fragments of output frame r
      const T* dp = dzb + (((r / FB) % 3) * FB + (r & (FB - 1))) * FRM + dlane;
      d[0] = tr_pair<T>(dp, dp + 4 * 64);
      d[1] = tr_pair<T>(dp + 16 * 64, dp + 20 * 64);
    };
    auto read_u = [&](int fi, frag_t (&u)[2]) __attribute__((always_inline)) {        // u^T fragments of stream frame fi
      const T* up = ub + (fi % UR) * FRM + ulane;
      u[0] = tr_pair<T>(up, up + 4 * 64);
      u[1] = tr_pair<T>(up + 16 * 64, up + 20 * 64);
    };

    // Staging pipeline.  "Stage k" = (dz batch k+2, u chunk k+1+LD): committed to LDS at the end of batch k, its global
    // loads issued at the START OF BATCH k-1 into one of two register sets -- two batches of loads (~50 KB per CU) are in
    // flight at any time.  (With one batch in flight the kernel ran at 1.8 TB/s: bytes in flight / memory latency.)
    // Prologue: dz batches 0 and 1, chunks 0..LD and stage 0, all loads in flight together.  (The previous segment's last
    // barrier has passed: no wave still reads the buffers.)
    const int nb = (Lseg + FB - 1) / FB;
    {
      u32x4 rp[LD + 1][NITU], rq[2][NITZ];
      issue_z(0, rq[0]);
      issue_z(1, rq[1]);
#pragma unroll
      for (int ck = 0; ck <= LD; ++ck) issue_u(ck, rp[ck]);
      if constexpr (DEEP) {
        issue_z(2, rzA);
        issue_u(1 + LD, ruA);
      }
      commit_z(0, rq[0]);
      commit_z(1, rq[1]);
#pragma unroll
      for (int ck = 0; ck <= LD; ++ck) commit_u(ck, rp[ck]);
    }
    __syncthreads();

    // Operands are read from LDS one step ahead of their MFMAs (dz^T: one frame ahead, u^T: one tap ahead); read straight
    // before use they cost ~150 cycles per MFMA pair.
    frag_t dT[2], dTn[2], uc[2], un[2];
    read_dz(0, dT);
    read_u(tg * JT, uc);
    auto batch = [&](auto odd_tag, int bt) __attribute__((always_inline)) {
      constexpr bool ODD = decltype(odd_tag)::value;           // parity of bt: which register set holds which stage
      if constexpr (!DEEP) { issue_z(bt + 2, rzA); issue_u(bt + 1 + LD, ruA); }    // (16 taps: one set, one batch ahead)
      else if constexpr (ODD) { issue_z(bt + 3, rzA); issue_u(bt + 2 + LD, ruA); } // stage bt+1 (even) -> set A
      else { issue_z(bt + 3, rzB); issue_u(bt + 2 + LD, ruB); }                     // stage bt+1 (odd)  -> set B
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int fo = 0; fo < FB; ++fo) {
        const int r = bt * FB + fo;
        read_dz(r + 1, dTn);                                   // (beyond the segment: a zero or stale frame, never used)
#pragma unroll
        for (int j = 0; j < JT; ++j) {
          read_u(j + 1 < JT ? S * r + tg * JT + j + 1 : S * (r + 1) + tg * JT, un);
          mma_kgroup(acc[j], dT[0], uc[0]);
          mma_kgroup(acc[j], dT[1], uc[1]);
          uc[0] = un[0];
          uc[1] = un[1];
        }
        dT[0] = dTn[0];
        dT[1] = dTn[1];
      }
      if constexpr (DEEP && ODD) { commit_z(bt + 2, rzB); commit_u(bt + 1 + LD, ruB); }    // stage bt (odd)  <- set B
      else { commit_z(bt + 2, rzA); commit_u(bt + 1 + LD, ruA); }                           // stage bt (even) <- set A
      __syncthreads();
    };
    for (int bt = 0; bt < nb; bt += 2) {
      batch(std::false_type{}, bt);
      if (bt + 1 < nb) batch(std::true_type{}, bt + 1);
    }
  }

  // ---- flush: acc[j] rows = output channel (registers), lane = input channel ----
  const int n0 = P.ntaps * Cout * Cin;
  const int orow = ob * 64 + 32 * ot, icol = ib * 64 + 32 * it + c;
  float* dstW = P.ws ? P.ws + (size_t)grp * P.ws_slice : P.dW;
  float* dstB = P.ws ? P.ws + (size_t)grp * P.ws_slice + n0 : P.dbias;
#pragma unroll
  for (int j = 0; j < JT; ++j) {
    const int tap = tg * JT + j;
    if (tap < P.ntaps) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int r = (i & 3) + 8 * (i >> 2) + 4 * h;
        float* p = dstW + ((size_t)tap * Cout + orow + r) * Cin + icol;
        if (P.ws) *p = acc[j][i]; else atomicAdd(p, acc[j][i]);
      }
    }
  }
  if (do_bias) {
#pragma unroll
    for (int e = 0; e < 8; ++e) atomicAdd(&bsum[8 * sq + e], dbs[e]);
    __syncthreads();
    if (tid < 64) {
      if (P.ws) dstB[ob * 64 + tid] = bsum[tid]; else atomicAdd(dstB + ob * 64 + tid, bsum[tid]);
    }
  }
}

template <typename T, int JT, int S, int FB>
int trc_launch(TrcParams P, int grid_cap, hipStream_t stream) {
  auto kfn = tconv_rc_wgrad_kernel<T, JT, S, FB>;
  constexpr int CH = S * FB, LD = (2 * JT + CH - 1) / CH, UR = (LD + 2) * CH;
  const size_t lds = (size_t)(3 * FB + UR) * FRM * 2 + 64 * 4;
  if (lds > 160 * 1024) return ISTGCN_EINVAL;
  static std::atomic<unsigned long long> optin{0};
  if (int ea = istgcn_lds_optin((const void*)kfn, optin)) return ea;
  int res = grid_cap > 0 ? grid_cap : istgcn_resident_blocks((const void*)kfn, RC_NTH, lds);
  P.nib = P.Cin / 64;
  P.gy = P.nib * (P.Cout / 64);
  int G = res / P.gy / 8 * 8;
  if (G < 8) G = 8;
  // segments: about one per group (a segment start re-stages its halo and waits out one memory round trip), at least 16
  // output frames each
  int nsps = (G + P.NM - 1) / P.NM;
  if (nsps > P.Tz / 16) nsps = P.Tz / 16;
  if (nsps < 1) nsps = 1;
  P.L = (P.Tz + nsps - 1) / nsps;
  P.nsps = (P.Tz + P.L - 1) / P.L;
  P.nseg = P.NM * P.nsps;
  while (G > 8 && G - 8 >= P.nseg) G -= 8;
  P.G = G;
  const long long n0 = (long long)P.ntaps * P.Cout * P.Cin, n1 = P.dbias ? P.Cout : 0;
  const bool use_ws = P.ws && (long long)G * (n0 + n1) <= P.ws_slice && G >= 32;
  if (use_ws) P.ws_slice = n0 + n1; else P.ws = nullptr;
  ISTGCN_LAUNCH(kfn, dim3(G * P.gy), dim3(RC_NTH), lds, stream, P);
  ISTGCN_CHECK_LAUNCH();
  if (use_ws) return istgcn_wgrad_reduce(P.ws, n0 + n1, G, P.dW, (int)n0, P.dbias, (int)n1, stream);
  return ISTGCN_OK;
}

template <typename T>
int trc_T(const TrcParams& P, int stride, int grid_cap, hipStream_t stream) {
  if (P.ntaps <= 10) {
    if (stride == 1) return trc_launch<T, 5, 1, 4>(P, grid_cap, stream);
    return trc_launch<T, 5, 2, 2>(P, grid_cap, stream);
  }
  if (stride == 1) return trc_launch<T, 8, 1, 4>(P, grid_cap, stream);
  return trc_launch<T, 8, 2, 2>(P, grid_cap, stream);
}

}  // namespace

extern "C" int istgcn_tconv_wgrad_rc_ok(int V, int Cin, int Cout, int ntaps, const int* tap_off, int in_mul, int dtype) {
  if ((dtype != 1 && dtype != 2) || V > 32 || Cin < 64 || Cin % 64 || Cout < 64 || Cout % 64) return 0;
  if (ntaps < 2 || ntaps > 16 || (in_mul != 1 && in_mul != 2)) return 0;
  for (int j = 1; j < ntaps; ++j) if (tap_off[j] != tap_off[0] + j) return 0;
  return 1;
}

extern "C" int istgcn_tconv_wgrad_rc(const void* dz, const void* g, const float* pre, int pre_relu, float* dW, float* dbias,
                                     int NM, int Tin, int Tz, int V, int Cin, int Cout, int ntaps, const int* tap_off,
                                     int in_mul, int dtype, int grid_cap, float* ws, long long ws_floats, void* stream) {
  if (!istgcn_tconv_wgrad_rc_ok(V, Cin, Cout, ntaps, tap_off, in_mul, dtype)) return ISTGCN_EINVAL;
  TrcParams P{};
  P.dz = dz; P.g = g; P.pre = pre; P.dW = dW; P.dbias = dbias; P.ws = ws_floats > 0 ? ws : nullptr; P.ws_slice = ws_floats;
  P.NM = NM; P.Tin = Tin; P.Tz = Tz; P.V = V; P.Cin = Cin; P.Cout = Cout; P.ntaps = ntaps; P.tap0 = tap_off[0];
  P.pre_relu = pre_relu;
  if (dtype == 1) return trc_T<__bf16>(P, in_mul, grid_cap, (hipStream_t)stream);
  return trc_T<_Float16>(P, in_mul, grid_cap, (hipStream_t)stream);
}
