// Entry of the lean temporal-conv weight gradient (tconv_wgrad_lean.hip) for the dispatcher in tconv_wgrad.hip.
#pragma once
#include <hip/hip_runtime.h>

// shapes the kernel serves (16-bit storage, 64-channel blocks, no conv-bias gradient asked for; stride 1 with 4..15 taps,
// stride 2 with 8..15 consecutive taps as one unit-stride launch per tap parity)
bool twg_lean_ok(int V, int Cin, int Cout, int ntaps, const int* tap_off, int in_mul, int dtype, int Tin, int Tz);
// returns an ISTGCN_* code, or -1 when the LDS plan does not fit (the caller falls back to the round-1 kernel of tconv_wgrad.hip)
int twg_lean_launch(const void* dz, const void* g, const float* pre, int pre_relu, float* dW, int NM, int Tin, int Tz, int V,
                    int Cin, int Cout, int ntaps, const int* tap_off, int in_mul, int dtype, int grid_cap, float* ws, long long ws_floats,
                    hipStream_t stream);
// dbias[o] += column sums of dz (16-bit storage, Cout in {64, 128, 256}): the bias gradient next to a lean launch
int twg_lean_dbias(const void* dz, float* dbias, int NM, int Tz, int V, int Cout, int dtype, hipStream_t stream);
