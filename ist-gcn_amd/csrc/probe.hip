// Hardware-convention probes (tests only): they pin the MFMA operand / result lane maps and the
// ds_read_b64_tr_b16 gather that the production kernels assume, with exact integer-valued data.
#include "common.hpp"

namespace {

// D[32][32] = A[32][KGS] * Bt[32][KGS]^T with the library's k-group convention (common.hpp).
template <typename T>
__global__ void probe_mfma_kernel(const T* A, const T* Bt, float* D) {
  using E = Elem<T>;
  typedef typename E::frag frag_t;
  const int lane = threadIdx.x;
  const int r = lane & 31, h = lane >> 5;
  const frag_t a = *reinterpret_cast<const frag_t*>(A + r * E::KGS + h * E::EPL);
  const frag_t b = *reinterpret_cast<const frag_t*>(Bt + r * E::KGS + h * E::EPL);
  f32x16 acc;
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  mma_kgroup(acc, a, b);
  for (int i = 0; i < 16; ++i) D[mfma_row(i, lane) * 32 + (lane & 31)] = acc[i];
}

__global__ void probe_tr16_kernel(const unsigned short* src, int nelem, const int* lane_byte_off,
                                  unsigned short* out) {
  extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
  for (int i = threadIdx.x; i < nelem; i += 64) lds[i] = src[i];
  __syncthreads();
  const int off = lane_byte_off[threadIdx.x];
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4*)((__attribute__((address_space(3))) char*)lds + off));
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (unsigned short)v[j];
}

}  // namespace

extern "C" int istgcn_probe_mfma(const void* A, const void* Bt, float* D, int dtype, void* stream) {
  if (!A || !Bt || !D) return ISTGCN_EINVAL;
  if (dtype == 0)
    ISTGCN_LAUNCH(probe_mfma_kernel<float>, dim3(1), dim3(64), 0, (hipStream_t)stream, (const float*)A,
                       (const float*)Bt, D);
  else if (dtype == 1)
    ISTGCN_LAUNCH(probe_mfma_kernel<__bf16>, dim3(1), dim3(64), 0, (hipStream_t)stream, (const __bf16*)A,
                       (const __bf16*)Bt, D);
  else if (dtype == 2)
    ISTGCN_LAUNCH(probe_mfma_kernel<_Float16>, dim3(1), dim3(64), 0, (hipStream_t)stream, (const _Float16*)A,
                       (const _Float16*)Bt, D);
  else
    return ISTGCN_EINVAL;
  ISTGCN_CHECK_LAUNCH();
  return ISTGCN_OK;
}

extern "C" int istgcn_probe_tr16(const void* src, int nelem, const int* lane_byte_off, void* out, void* stream) {
  if (!src || !lane_byte_off || !out || nelem < 1 || nelem > 16384) return ISTGCN_EINVAL;
  ISTGCN_LAUNCH(probe_tr16_kernel, dim3(1), dim3(64), nelem * 2, (hipStream_t)stream,
                     (const unsigned short*)src, nelem, lane_byte_off, (unsigned short*)out);
  ISTGCN_CHECK_LAUNCH();
  return ISTGCN_OK;
}
