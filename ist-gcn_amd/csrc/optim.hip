// SGD (momentum, Nesterov, weight decay) over ONE flat fp32 range: the update of processor/recognition.py:154-159
// (optim.SGD(lr, momentum=0.9, nesterov, weight_decay)) + :289 (optimizer.step()) for every live parameter of the model
// in a single launch.  The host keeps parameters, gradients and momentum as views of three flat buffers (harness.FlatSGD);
// the gradient buffer is the very buffer the data-parallel all-reduce runs on (dp.FlatGradSync), so a step is
// all-reduce + this kernel.
//
//   g' = grad_scale * g + weight_decay * p          (grad_scale: 1/world for the all-reduce SUM, 1/loss_scale for fp16)
//   m  = momentum * m + g'                          (m starts at zero: the first step gives m = g', as torch does)
//   p -= lr * (nesterov ? g' + momentum * m : m)
//
// Non-finite gradients (an overflowed float16 activation gradient under a static loss scale, which an all-reduce then
// spreads to every rank) are never applied.  The way torch.cuda.amp.GradScaler does it: istgcn_grad_nonfinite reduces
// "any inf / NaN in the flat gradient buffer" into a device flag first, and istgcn_sgd_step given that flag as `skip_if`
// is a no-op for the WHOLE step when it is set (a partial update with the overflowed scale would be a silent wrong
// step).  Both raise *found_inf, which the host polls when it chooses to (harness.FlatSGD.check_overflow backs the loss
// scale off).  Without `skip_if` the update still never applies an element whose own gradient is inf / NaN.
//
// HBM-bound: 3 reads + 2 writes of 4 bytes per parameter (12.6 MB of parameters for config 2 -> ~63 MB per step).
#include "common.hpp"

namespace {

__global__ __launch_bounds__(256) void sgd_step_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                       float* __restrict__ m, long long n, float lr, float momentum,
                                                       float wd, int nesterov, float gscale, int* __restrict__ found_inf,
                                                       const int* __restrict__ skip_if) {
  if (skip_if && *skip_if != 0) {        // the pre-pass found a non-finite gradient: nothing is applied (uniform over the launch)
    if (found_inf && blockIdx.x == 0 && threadIdx.x == 0) atomicOr(found_inf, 1);
    return;
  }
  bool bad = false;
  const long long n4 = n >> 2;
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    f32x4 pv = reinterpret_cast<f32x4*>(p)[i];
    const f32x4 gv = reinterpret_cast<const f32x4*>(g)[i];
    f32x4 mv = reinterpret_cast<f32x4*>(m)[i];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float gg = gscale * gv[j] + wd * pv[j];
      const bool ok = __builtin_isfinite(gv[j]);
      bad |= !ok;
      const float mn = momentum * mv[j] + gg;
      const float pn = pv[j] - lr * (nesterov ? gg + momentum * mn : mn);
      mv[j] = ok ? mn : mv[j];
      pv[j] = ok ? pn : pv[j];
    }
    reinterpret_cast<f32x4*>(p)[i] = pv;
    reinterpret_cast<f32x4*>(m)[i] = mv;
  }
  // tail (n not a multiple of 4)
  const long long t = (n4 << 2) + (long long)blockIdx.x * 256 + threadIdx.x;
  if (t < n) {
    const float gg = gscale * g[t] + wd * p[t];
    const float mm = momentum * m[t] + gg;
    if (__builtin_isfinite(g[t])) {
      m[t] = mm;
      p[t] -= lr * (nesterov ? gg + momentum * mm : mm);
    } else bad = true;
  }
  if (found_inf && __any(bad) && (threadIdx.x & 63) == 0) atomicOr(found_inf, 1);
}

// any(!isfinite(g)) over the flat gradient buffer -> *flag |= 1 (one 4-byte read per gradient: 12.6 MB for config 2)
__global__ __launch_bounds__(256) void grad_nonfinite_kernel(const float* __restrict__ g, long long n, int* __restrict__ flag) {
  bool bad = false;
  const long long n4 = n >> 2;
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    const f32x4 gv = reinterpret_cast<const f32x4*>(g)[i];
    // inf - inf and NaN - NaN are NaN: one subtraction and one compare per vector element pair
    const float t = (gv[0] - gv[0]) + (gv[1] - gv[1]) + (gv[2] - gv[2]) + (gv[3] - gv[3]);
    bad |= !(t == 0.f);
  }
  const long long t = (n4 << 2) + (long long)blockIdx.x * 256 + threadIdx.x;
  if (t < n) bad |= !__builtin_isfinite(g[t]);
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

}  // namespace

extern "C" int istgcn_grad_nonfinite(const float* grads, long long n, int* flag, void* stream) {
  if (!grads || !flag || n < 0) return ISTGCN_EINVAL;
  if ((uintptr_t)grads & 15) return ISTGCN_EINVAL;
  if (n == 0) return ISTGCN_OK;
  long long blocks = ((n >> 2) + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  if (blocks < 1) blocks = 1;
  ISTGCN_LAUNCH(grad_nonfinite_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, grads, n, flag);
  ISTGCN_CHECK_LAUNCH();
  return ISTGCN_OK;
}

extern "C" int istgcn_sgd_step(float* params, const float* grads, float* momentum_buf, long long n, float lr,
                               float momentum, float weight_decay, int nesterov, float grad_scale, int* found_inf,
                               const int* skip_if, void* stream) {
  if (!params || !grads || !momentum_buf || n < 0) return ISTGCN_EINVAL;
  if (((uintptr_t)params | (uintptr_t)grads | (uintptr_t)momentum_buf) & 15) return ISTGCN_EINVAL;   // 16-byte vectors
  if (n == 0) return ISTGCN_OK;
  long long blocks = ((n >> 2) + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  ISTGCN_LAUNCH(sgd_step_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, params, grads, momentum_buf, n, lr,
                momentum, weight_decay, nesterov, grad_scale, found_inf, skip_if);
  ISTGCN_CHECK_LAUNCH();
  return ISTGCN_OK;
}
