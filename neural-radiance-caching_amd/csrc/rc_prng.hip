// Counter-based random fill in HBM with the stream layout of the reference's jax.random (jax 0.4.16: threefry2x32,
// non-partitionable counters) -- SURVEY.md §8(f) rank 3.  Host twin and the conventions: ../prng.py.
//
// Layout (jax/_src/prng.py of the pinned jax, restated): the counters iota(n) are cut in two halves; block i has
// counter words (i, i + half) and its two output words land at out[i] and out[i + half]; an odd n is padded with
// a zero counter whose output is dropped.  One lane per block: 20 rounds of 32-bit add/rotate/xor, two coalesced
// 4-byte stores -- a pure streaming kernel (8 bytes written per block, nothing read).
#include <hip/hip_runtime.h>

#include "rc_internal.h"

namespace {

__device__ __forceinline__ uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

__device__ __forceinline__ void threefry2x32(uint32_t k0, uint32_t k1, uint32_t& x0, uint32_t& x1) {
  const uint32_t ks[3] = {k0, k1, k0 ^ k1 ^ 0x1BD11BDAu};
  constexpr int R[2][4] = {{13, 15, 26, 6}, {17, 29, 16, 24}};
  x0 += ks[0];
  x1 += ks[1];
#pragma unroll
  for (int i = 0; i < 5; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      x0 += x1;
      x1 = rotl32(x1, R[i & 1][j]) ^ x0;
    }
    x0 += ks[(i + 1) % 3];
    x1 += ks[(i + 2) % 3] + (uint32_t)(i + 1);
  }
}

// XLA's single-precision erfinv (Giles' polynomial), evaluated in the same order as ../prng.py
__device__ __forceinline__ float erfinv32(float x) {
  const float w = -log1pf(-x * x);
  float p, ww;
  if (w < 5.0f) {
    ww = w - 2.5f;
    p = 2.81022636e-08f;
    p = 3.43273939e-07f + p * ww;
    p = -3.5233877e-06f + p * ww;
    p = -4.39150654e-06f + p * ww;
    p = 0.00021858087f + p * ww;
    p = -0.00125372503f + p * ww;
    p = -0.00417768164f + p * ww;
    p = 0.246640727f + p * ww;
    p = 1.50140941f + p * ww;
  } else {
    ww = sqrtf(w) - 3.0f;
    p = -0.000200214257f;
    p = 0.000100950558f + p * ww;
    p = 0.00134934322f + p * ww;
    p = -0.00367342844f + p * ww;
    p = 0.00573950773f + p * ww;
    p = -0.0076224613f + p * ww;
    p = 0.00943887047f + p * ww;
    p = 1.00167406f + p * ww;
    p = 2.83297682f + p * ww;
  }
  return p * x;
}

__device__ __forceinline__ float unit_float(uint32_t bits) { return __uint_as_float((bits >> 9) | 0x3F800000u) - 1.0f; }

__device__ __forceinline__ uint32_t shape_value(uint32_t bits, int mode, float lo, float hi) {
  if (mode == RC_PRNG_BITS) return bits;
  const float u = fmaxf(lo, unit_float(bits) * (hi - lo) + lo);
  float v = u;
  if (mode == RC_PRNG_NORMAL) v = 1.41421356237309515f * erfinv32(u);
  if (mode == RC_PRNG_GUMBEL) v = -logf(-logf(u));
  return __float_as_uint(v);
}

__global__ __launch_bounds__(256) void k_prng_fill(RcPrngArgs a) {
  const int64_t half = (a.n + 1) >> 1;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < half; i += (int64_t)gridDim.x * 256) {
    uint32_t x0 = (uint32_t)i;
    uint32_t x1 = i + half < a.n ? (uint32_t)(i + half) : 0u;     // odd n: zero pad
    threefry2x32(a.key0, a.key1, x0, x1);
    a.out[i] = shape_value(x0, a.mode, a.lo, a.hi);
    if (i + half < a.n) a.out[i + half] = shape_value(x1, a.mode, a.lo, a.hi);
  }
}

}  // namespace

void rc_launch_prng_fill(const RcPrngArgs& a, hipStream_t stream) {
  const int64_t half = (a.n + 1) >> 1;
  int64_t blocks = (half + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;       // grid-stride beyond 2M blocks of counters
  hipLaunchKernelGGL(k_prng_fill, dim3((unsigned)blocks), dim3(256), 0, stream, a);
}
