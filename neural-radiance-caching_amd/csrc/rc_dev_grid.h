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
  if constexpr (F == 4) {
    const float4 q = reinterpret_cast<const float4*>(table)[idx];
    r.v[0] = q.x; r.v[1] = q.y; r.v[2] = q.z; r.v[3] = q.w;
  } else {
    r.v[0] = table[idx];
  }
  return r;
}

// coord.contract(x / radius)
__device__ __forceinline__ void contract3(float& x, float& y, float& z, float radius) {
  x = x / radius; y = y / radius; z = z / radius;
  float mag = x * x + y * y + z * z;
  mag = fmaxf(1.0f, mag);
  const float scale = (2.0f * sqrtf(mag) - 1.0f) / mag;
  x = scale * x; y = scale * y; z = scale * z;
}


// Trilinear lookup of level L at the (already contracted) position (x, y, z).
// acc[F]: interpolated features (NOT yet scaled by the precondition factor);
// jacc[3*F] (JAC): d feature / d loc_a for the three location axes in the level's own axis order.
template <int F, bool JAC>
__device__ __forceinline__ void grid_level(const RcGridLevel& L, float bbox, float x, float y, float z, float (&acc)[F],
                                           float (&jacc)[JAC ? 3 * F : 1]) {
  const float lo = -bbox, hi = bbox;
  const float N = (float)L.size;
  // x01 * grid_size (grid_utils.py:820, 863)
  const float cx = ((x - lo) / (hi - lo)) * N;
  const float cy = ((y - lo) / (hi - lo)) * N;
  const float cz = ((z - lo) / (hi - lo)) * N;

  float loc[3];
  if (L.dense) {
    // trilerp 'grid' branch: flip(coords - 0.5) then +1 for the zero padding (grid_utils.py:711, 390)
    loc[0] = (cz - 0.5f) + 1.0f; loc[1] = (cy - 0.5f) + 1.0f; loc[2] = (cx - 0.5f) + 1.0f;
  } else {
    loc[0] = cx - 0.5f; loc[1] = cy - 0.5f; loc[2] = cz - 0.5f;     // grid_utils.py:61
  }
  float fl[3], cw[3], fw[3];
  int base[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    fl[a] = floorf(loc[a]);
    cw[a] = loc[a] - fl[a];
    fw[a] = 1.0f - cw[a];
    base[a] = (int)fl[a];
  }

  // Issue the 8 corner fetches, then combine in the reference's corner order (b2 fastest).
  Vec<F> val[8];
  const int Ni = L.size;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int b0 = (c >> 2) & 1, b1 = (c >> 1) & 1, b2 = c & 1;
    const int i0 = base[0] + b0, i1 = base[1] + b1, i2 = base[2] + b2;
    if (L.dense) {
      // clamp to the padded volume [0, N+1]; the pad (0 and N+1) holds zeros (grid_utils.py:384-390, 435-438)
      const int k0 = min(max(i0, 0), Ni + 1), k1 = min(max(i1, 0), Ni + 1), k2 = min(max(i2, 0), Ni + 1);
      const bool inside = (k0 >= 1) & (k0 <= Ni) & (k1 >= 1) & (k1 <= Ni) & (k2 >= 1) & (k2 <= Ni);
      // data[loc2, loc1, loc0] = grid[x, y, z]
      const uint32_t idx = ((uint32_t)(k2 - 1) * (uint32_t)Ni + (uint32_t)(k1 - 1)) * (uint32_t)Ni + (uint32_t)(k0 - 1);
      if (inside) {
        val[c] = load_entry<F>(L.table, idx);
      } else {
#pragma unroll
        for (int f = 0; f < F; ++f) val[c].v[f] = 0.0f;
      }
    } else {
      // int32 -> uint32 wraparound hash (grid_utils.py:99-111)
      const uint32_t h = (uint32_t)i0 ^ ((uint32_t)i1 * kPi2) ^ ((uint32_t)i2 * kPi3);
      const uint32_t idx = L.mask ? (h & L.mask) : (h % L.entries);
      val[c] = load_entry<F>(L.table, idx);
    }
  }

#pragma unroll
  for (int f = 0; f < F; ++f) acc[f] = 0.0f;
  if constexpr (JAC) {
#pragma unroll
    for (int f = 0; f < 3 * F; ++f) jacc[f] = 0.0f;
  }
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int b0 = (c >> 2) & 1, b1 = (c >> 1) & 1, b2 = c & 1;
    const float w0 = b0 ? cw[0] : fw[0], w1 = b1 ? cw[1] : fw[1], w2 = b2 ? cw[2] : fw[2];
    const float w = (w0 * w1) * w2;
#pragma unroll
    for (int f = 0; f < F; ++f) acc[f] = acc[f] + val[c].v[f] * w;
    if constexpr (JAC) {
      // d w / d loc_a = +-(product of the two other weights)
      const float d0 = (b0 ? 1.0f : -1.0f) * (w1 * w2);
      const float d1 = (b1 ? 1.0f : -1.0f) * (w0 * w2);
      const float d2 = (b2 ? 1.0f : -1.0f) * (w0 * w1);
#pragma unroll
      for (int f = 0; f < F; ++f) {
        jacc[0 * F + f] += val[c].v[f] * d0;
        jacc[1 * F + f] += val[c].v[f] * d1;
        jacc[2 * F + f] += val[c].v[f] * d2;
      }
    }
  }

}

}  // namespace rcdev
