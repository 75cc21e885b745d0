// Tiling geometry of the wave-specialised temporal-conv kernels (tconv.hip, tconv_lean.hip): the launchers and the
// geometry query the host packs weights by share ONE decision, so both kernels read the same packed weights.
#pragma once
#include "common.hpp"
#include <cstdlib>

namespace tconv_geo {

constexpr int NROLE = 256;   // threads per role (4 waves)
constexpr int UL = 8;        // 16-byte vectors of a staged chunk per memory-wave thread (rows x vectors <= UL * 256, checked here)

// Tiling decision shared by the launcher and the geometry query (the host packs weights to match).
struct TconvGeom { int CC, nch, NKG, MT, MTtot, gy, NT, F, Fin, min_off, lds, off_stat, off_u0, off_u1, off_o, us_stride, out_stride; };

// nt_max = 1: the short-tile variant (128 rows) for launches too small to give every CU a 256-row tile; the launcher takes
// it only when the channel chunking -- which the packed weights depend on -- comes out the same as for nt_max = 2.
inline int tconv_geom(int V, int Cin, int Cout, int ntaps, const int* tap_off, int in_mul, int dtype, TconvGeom* G, int nt_max = 2) {
  const int epl = dtype == 0 ? 4 : 8, kgs = 2 * epl, esz = dtype == 0 ? 4 : 2;
  int mn = tap_off[0], mx = tap_off[0];
  for (int j = 1; j < ntaps; ++j) { mn = tap_off[j] < mn ? tap_off[j] : mn; mx = tap_off[j] > mx ? tap_off[j] : mx; }
  G->min_off = mn;
  G->MT = Cout <= 32 ? 1 : Cout <= 64 ? 2 : 4;
  G->gy = ceil_div(Cout, G->MT * 32);
  G->MTtot = G->gy * G->MT;
  const int ow = 256 / esz;
  G->out_stride = ow + epl;
  // double-buffered chunk tile + output image + tables in one CU's LDS; the widest chunk and the tallest tile that fit
  // (a staged row is CC + EPL elements: 80 bytes at the full chunk width, conflict-free for the 16-byte fragment reads)
  const int cc_max = dtype == 0 ? 16 : 32;
  const int budget = 160 * 1024;
  int best_nt = 0, best_cc = 0;
  for (int nt = nt_max; nt >= 1 && !best_nt; --nt) {
    const int F = nt * 128 / V;
    if (F < 1) continue;
    for (int cc = cc_max; cc >= kgs; cc >>= 1) {
      int cce = Cin < cc ? round_up(Cin, kgs) : cc;
      const int Fin = in_mul * (F - 1) + (mx - mn) + 1;
      const long rows = (long)Fin * V;
      const long tables = 1024 + (long)(3 * G->MT * 32 + 2 * round_up(Cin, cce)) * 4 + 64;
      const long need = tables + 2 * rows * (cce + epl) * esz + (long)128 * nt * G->out_stride * esz;
      if (need <= budget && rows * (cce / epl) <= UL * NROLE) { best_nt = nt; best_cc = cce; break; }
    }
  }
  if (!best_nt) return ISTGCN_EINVAL;
  const int cc = best_cc;
  G->NT = best_nt; G->CC = cc; G->nch = ceil_div(Cin, cc); G->NKG = cc / kgs;
  if (G->NKG & (G->NKG - 1)) return ISTGCN_EINVAL;
  G->F = best_nt * 128 / V;
  G->Fin = in_mul * (G->F - 1) + (mx - mn) + 1;
  G->us_stride = cc + epl;
  size_t off = (size_t)2 * 128 * best_nt * sizeof(unsigned short);
  off = (off + 15) & ~(size_t)15; G->off_stat = (int)off;
  off += (size_t)(3 * G->MT * 32 + 2 * G->nch * cc) * 4;                     // BN partial sums, conv bias, `pre` rows
  const size_t ubytes = (((size_t)G->Fin * V * G->us_stride * esz) + 15) & ~(size_t)15;
  off = (off + 15) & ~(size_t)15; G->off_u0 = (int)off; off += ubytes;
  G->off_u1 = (int)off; off += ubytes;
  G->off_o = (int)off; off += (size_t)128 * best_nt * G->out_stride * esz;
  G->lds = (int)off;
  return off <= 160 * 1024 ? ISTGCN_OK : ISTGCN_EINVAL;
}


// ---- the lean kernel (tconv_lean.hip): ONE decision shared by the geometry query (= the weight packer) and the launcher.
// Staged rows of 64 bytes (32-channel chunks, swizzled), UL * 64 rows per buffer, 256-row tiles, MT * 32 output channels
// per workgroup.  A shape is "lean" when every mode a caller can ask for with this packing has a lean instantiation
// (9 / 15 taps: forward, data gradient, inference) or when tconv.hip packs the same way and serves the rest (4 / 5 taps:
// the phases of a stride-2 data gradient go to the lean kernel, anything else with those taps to tconv.hip).
struct LeanGeom { int UL, MT, gy, MTtot, nch, F, Fin, min_off, off_stat, off_u0, off_u1, off_o, lds; };

inline bool tconv_lean_geom(int V, int Cin, int Cout, int ntaps, const int* tap_off, int in_mul, int dtype, LeanGeom* L) {
  static const bool off = [] { const char* e = getenv("ISTGCN_TCONV_LEAN"); return e && atoi(e) == 0; }();   // dispatch override, read once
  if (off || dtype == 0 || V < 2 || V > 128 || Cin % 32 != 0 || Cout % 64 != 0) return false;
  if (ntaps != 4 && ntaps != 5 && ntaps != 9 && ntaps != 15) return false;
  const int d = tap_off[1] - tap_off[0];               // (equally spaced, ascending or descending: the data gradient lists them flipped)
  if (d == 0) return false;
  for (int j = 2; j < ntaps; ++j) if (tap_off[j] - tap_off[j - 1] != d) return false;
  L->MT = Cout <= 64 ? 2 : 4;
  if (Cout % (L->MT * 32) != 0) return false;
  L->gy = Cout / (L->MT * 32);
  L->MTtot = L->gy * L->MT;
  L->nch = Cin / 32;
  if (L->nch < L->MT) return false;                   // the image of a tile is streamed out in MT parts, one per item of the next tile
  const int t_lo = d > 0 ? tap_off[0] : tap_off[ntaps - 1], t_hi = d > 0 ? tap_off[ntaps - 1] : tap_off[0];
  L->min_off = t_lo;
  L->F = 256 / V;
  L->Fin = in_mul * (L->F - 1) + (t_hi - t_lo) + 1;
  const int rows = L->Fin * V;
  if (rows > 11 * 64) return false;
  L->UL = (ntaps == 15 || rows > 8 * 64) ? 11 : 8;
  if (ntaps <= 5) {
    // (tconv.hip must pack these shapes identically: it serves every mode but the data gradient for them)
    if (L->UL != 8) return false;
    TconvGeom G;
    if (tconv_geom(V, Cin, Cout, ntaps, tap_off, in_mul, dtype, &G) != ISTGCN_OK || G.CC != 32 || G.nch != L->nch || G.MTtot != L->MTtot) return false;
  }
  size_t o = 0;
  L->off_stat = 0;
  o += (size_t)(3 * L->MT * 32 + 2 * Cin) * 4;
  o = (o + 15) & ~(size_t)15;
  const size_t ub = (size_t)L->UL * 64 * 64;
  L->off_u0 = (int)o; o += ub;
  L->off_u1 = (int)o; o += ub;
  L->off_o = (int)o; o += (size_t)256 * 136 * 2;
  L->lds = (int)o;
  return o <= 160 * 1024;
}

}  // namespace tconv_geo

// tconv_lean.hip: the lean kernel's launcher (the caller has checked tconv_lean_geom and tconv_lean_serves)
struct BnTail;
bool tconv_lean_serves(int mode, int ntaps, int ul);
int tconv_lean_launch(const void* in, const void* Wp, const float* bias, const float* pre, int pre_relu, const void* aux,
                      const float* maux, void* out, double* stats, int stats_rep, int mode, int NM, int Tin, int Tout,
                      int Mlog, int V, int Cin, int Cout, int ntaps, const int* tap_off, int in_mul, int out_mul, int out_off,
                      int dtype, int grid_cap, const tconv_geo::LeanGeom& G, const BnTail& tail, hipStream_t stream);
