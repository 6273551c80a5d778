// Device-side multiresolution grid lookup for one (point, level) (shared by rc_hashgrid.hip and
// rc_fused.hip).  See rc_hashgrid.hip for the reference mapping.
#pragma once
#include "rc_internal.h"

namespace rcdev {


constexpr uint32_t kPi2 = 19349663u;   // grid_utils.py:102
constexpr uint32_t kPi3 = 83492791u;   // grid_utils.py:103

template <int F> struct Vec;
template <> struct Vec<1> { float v[1]; };
template <> struct Vec<4> { float v[4]; };

template <int F>
__device__ __forceinline__ Vec<F> load_entry(const float* __restrict__ table, uint32_t idx) {
  Vec<F> r;
#ifdef RC_GATHER_FAKE
  idx &= RC_GATHER_FAKE;
#endif
  if constexpr (F == 4) {
    const float4 q = reinterpret_cast<const float4*>(table)[idx];
    r.v[0] = q.x; r.v[1] = q.y; r.v[2] = q.z; r.v[3] = q.w;
  } else {
    r.v[0] = table[idx];
  }
  return r;
}

// coord.contract(x / radius)
#ifndef RC_DEV_CONTRACT3
#define RC_DEV_CONTRACT3
__device__ __forceinline__ void contract3(float& x, float& y, float& z, float radius) {
  x = rc_div(x, radius); y = rc_div(y, radius); z = rc_div(z, radius);
  float mag = x * x + y * y + z * z;
  mag = fmaxf(1.0f, mag);
  const float scale = (2.0f * sqrtf(mag) - 1.0f) / mag;
  x = scale * x; y = scale * y; z = scale * z;
}
#endif


// Trilinear lookup of level L at the (already contracted) position (x, y, z), in two halves so that a
// caller can put the fetches of several levels in flight before combining any of them:
//   grid_fetch   : cell + weights, issues the 8 corner loads (straight-line code: dense and hashed
//                  addressing are both computed and selected, a zero-padded corner loads entry 0 and is
//                  masked afterwards)
//   grid_combine : acc[F] interpolated features (NOT yet scaled by the precondition factor);
//                  jacc[3*F] (JAC): d feature / d loc_a for the three location axes in the level's own
//                  axis order.  Corners are combined in the reference's order (b2 fastest).
template <int F> struct Corners { Vec<F> val[8]; float cw[3]; uint32_t zero_mask; };

// (x01, y01, z01) = unit_box(bbox, contracted position): level independent, computed once per point.
// `dense`, `size`, `mask`, `entries` must be wave-uniform (one branch per level, both sides straight-line);
// `table` may differ per lane.  The per-axis terms of the index are computed once per level (two
// candidates per axis), a corner then costs an xor/add, the address and the load.
// x01 = (x - bbox_min) / (bbox_max - bbox_min) (grid_utils.py:820, 863)
__device__ __forceinline__ float unit_box(float bbox, float x) { return rc_div(x - (-bbox), bbox - (-bbox)); }

// POW2: the caller guarantees power-of-two hash tables (mask != 0): no modulo path.
// STRIDE: distance between consecutive entries in units of F floats (2 for the interleaved [density | appearance]
// tables of the fused kernel, where `table` already points at this lane's half of an entry pair).
// CELL (dense levels of the fused kernel): `table` is a cell table built by the host -- for every cell origin of the
// zero-padded volume, (N + 3)^3 of them, the 8 corner entries in combine order, zeros for the padding already in
// place -- so a lane reads its 8 corners from ONE contiguous block (32 bytes for F = 1: two 16-byte loads; with
// STRIDE = 2 a 256-byte block of [density | appearance] pairs) instead of four scattered line pairs, and the clamp /
// zero-mask arithmetic per corner disappears.
template <int F, bool POW2 = false, int STRIDE = 1, bool CELL = false>
__device__ __forceinline__ void grid_fetch(const float* __restrict__ table, int size, uint32_t mask, uint32_t entries,
                                           bool dense, float x01, float y01, float z01, Corners<F>& C) {
  const float N = (float)size;
  // x01 * grid_size (grid_utils.py:820, 863)
  const float cx = x01 * N;
  const float cy = y01 * N;
  const float cz = z01 * N;
  C.zero_mask = 0;
  if (dense) {
    // trilerp 'grid' branch: flip(coords - 0.5) then +1 for the zero padding (grid_utils.py:711, 390)
    const float loc[3] = {(cz - 0.5f) + 1.0f, (cy - 0.5f) + 1.0f, (cx - 0.5f) + 1.0f};
    int base[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float fl = floorf(loc[a]);
      C.cw[a] = loc[a] - fl;
      base[a] = (int)fl;
    }
    if constexpr (CELL) {
      // cell origin clamped to [-1, N + 1]: beyond that both corners of an axis are padding anyway
      const int M = size + 3;
      const int q0 = min(max(base[0], -1), size + 1) + 1, q1 = min(max(base[1], -1), size + 1) + 1,
                q2 = min(max(base[2], -1), size + 1) + 1;
      const uint32_t cell = ((uint32_t)q2 * (uint32_t)M + (uint32_t)q1) * (uint32_t)M + (uint32_t)q0;
      if constexpr (F == 1 && STRIDE == 1) {
        const float4* cp = reinterpret_cast<const float4*>(table) + (size_t)cell * 2;
        const float4 lo = cp[0], hi = cp[1];
        C.val[0].v[0] = lo.x; C.val[1].v[0] = lo.y; C.val[2].v[0] = lo.z; C.val[3].v[0] = lo.w;
        C.val[4].v[0] = hi.x; C.val[5].v[0] = hi.y; C.val[6].v[0] = hi.z; C.val[7].v[0] = hi.w;
      } else {
#pragma unroll
        for (int c = 0; c < 8; ++c) C.val[c] = load_entry<F>(table, (cell * 8u + (uint32_t)c) * STRIDE);
      }
      return;
    }
    // clamp to the padded volume [0, N+1]; the pad (0 and N+1) holds zeros (grid_utils.py:384-390, 435-438);
    // data[loc2, loc1, loc0] = grid[x, y, z]: idx = ((k2-1) N + (k1-1)) N + (k0-1)
    uint32_t term[3][2];
    bool out[3][2];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int k = min(max(base[a] + b, 0), size + 1);
        out[a][b] = (k < 1) | (k > size);
        const uint32_t km = (uint32_t)(k - 1);
        term[a][b] = a == 0 ? km : (a == 1 ? km * (uint32_t)size : km * (uint32_t)size * (uint32_t)size);
      }
    // fetch order: the two corners that differ in b0 (adjacent entries, almost always one cache line) back to back
#pragma unroll
    for (int o = 0; o < 8; ++o) {
      const int c = ((o & 1) << 2) | (o >> 1);
      const int b0 = (c >> 2) & 1, b1 = (c >> 1) & 1, b2 = c & 1;
      const bool zero = out[0][b0] | out[1][b1] | out[2][b2];
      const uint32_t idx = zero ? 0u : term[2][b2] + term[1][b1] + term[0][b0];
      C.zero_mask |= (zero ? 1u : 0u) << c;
      C.val[c] = load_entry<F>(table, idx * STRIDE);
    }
  } else {
    const float loc[3] = {cx - 0.5f, cy - 0.5f, cz - 0.5f};     // grid_utils.py:61
    int base[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float fl = floorf(loc[a]);
      C.cw[a] = loc[a] - fl;
      base[a] = (int)fl;
    }
    // int32 -> uint32 wraparound hash (grid_utils.py:99-111): x ^ y * pi2 ^ z * pi3
    const uint32_t hx[2] = {(uint32_t)base[0], (uint32_t)base[0] + 1u};
    const uint32_t y0 = (uint32_t)base[1] * kPi2, z0 = (uint32_t)base[2] * kPi3;
    const uint32_t hy[2] = {y0, y0 + kPi2}, hz[2] = {z0, z0 + kPi3};
    // fetch order: x and x + 1 (index differs in the low bits only: same cache line 7 times out of 8) back to back
#pragma unroll
    for (int o = 0; o < 8; ++o) {
      const int c = ((o & 1) << 2) | (o >> 1);
      const int b0 = (c >> 2) & 1, b1 = (c >> 1) & 1, b2 = c & 1;
      const uint32_t hsh = hx[b0] ^ hy[b1] ^ hz[b2];
      const uint32_t idx = (POW2 || mask) ? (hsh & mask) : (hsh % entries);
      C.val[c] = load_entry<F>(table, idx * STRIDE);
    }
  }
}

template <int F, bool JAC>
__device__ __forceinline__ void grid_combine(const Corners<F>& C, float (&acc)[F], float (&jacc)[JAC ? 3 * F : 1]) {
  float fw[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) fw[a] = 1.0f - C.cw[a];
#pragma unroll
  for (int f = 0; f < F; ++f) acc[f] = 0.0f;
  float v[8][F];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int b0 = (c >> 2) & 1, b1 = (c >> 1) & 1, b2 = c & 1;
    const float w0 = b0 ? C.cw[0] : fw[0], w1 = b1 ? C.cw[1] : fw[1], w2 = b2 ? C.cw[2] : fw[2];
    const float w = (w0 * w1) * w2;
    const bool zero = (C.zero_mask >> c) & 1u;
#pragma unroll
    for (int f = 0; f < F; ++f) v[c][f] = zero ? 0.0f : C.val[c].v[f];
#pragma unroll
    for (int f = 0; f < F; ++f) acc[f] = acc[f] + v[c][f] * w;
  }
  if constexpr (JAC) {
    // d feature / d loc_a = sum over the 4 corner pairs along axis a of (product of the two other axes' weights) x
    // (difference of the pair): 12 differences + 12 fused multiply-adds per feature.  (The value above keeps the
    // reference's corner order and separate roundings; the derivative only feeds the analytic normals, whose own
    // reference is an autodiff gradient with no prescribed summation order.)
    const float w0s[2] = {fw[0], C.cw[0]}, w1s[2] = {fw[1], C.cw[1]}, w2s[2] = {fw[2], C.cw[2]};
    float p12[2][2], p02[2][2], p01[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int k = 0; k < 2; ++k) { p12[i][k] = w1s[i] * w2s[k]; p02[i][k] = w0s[i] * w2s[k]; p01[i][k] = w0s[i] * w1s[k]; }
#pragma unroll
    for (int f = 0; f < F; ++f) {
      float g0 = 0.0f, g1 = 0.0f, g2 = 0.0f;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          // corner index c = 4 b0 + 2 b1 + b2
          g0 = __builtin_fmaf(v[4 + 2 * i + k][f] - v[2 * i + k][f], p12[i][k], g0);          // (b1, b2) = (i, k)
          g1 = __builtin_fmaf(v[4 * i + 2 + k][f] - v[4 * i + k][f], p02[i][k], g1);          // (b0, b2) = (i, k)
          g2 = __builtin_fmaf(v[4 * i + 2 * k + 1][f] - v[4 * i + 2 * k][f], p01[i][k], g2);  // (b0, b1) = (i, k)
        }
      jacc[0 * F + f] = g0; jacc[1 * F + f] = g1; jacc[2 * F + f] = g2;
    }
  }
}

template <int F, bool JAC>
__device__ __forceinline__ void grid_level(const RcGridLevel& L, float bbox, float x, float y, float z, float (&acc)[F],
                                           float (&jacc)[JAC ? 3 * F : 1]) {
  Corners<F> C;
  if constexpr (F == 1 && !JAC) {
    // one contiguous 32-byte read per cell instead of eight scattered corners (wave-uniform branch: blockIdx.y = level)
    if (L.cell) {
      grid_fetch<1, false, 1, true>(L.cell, L.size, L.mask, L.entries, true, unit_box(bbox, x), unit_box(bbox, y), unit_box(bbox, z), C);
      grid_combine<F, JAC>(C, acc, jacc);
      return;
    }
  }
  grid_fetch<F>(L.table, L.size, L.mask, L.entries, L.dense != 0, unit_box(bbox, x), unit_box(bbox, y), unit_box(bbox, z), C);
  grid_combine<F, JAC>(C, acc, jacc);
}

}  // namespace rcdev
