// Multiresolution dense+hash grid lookup for gfx950 (HBM/L2-bound gather).
//
// Replaces HashEncoding.__call__ (internal/grid_utils.py:808-905) with its two trilinear
// resamplers jax_resample_3d (:352-445, dense levels, zero padded, [x,y,z] indexed) and
// jax_hash_resample_3d (:41-121, hashed levels), fused with the scene contraction
// coord.contract_radius_2 (internal/coord.py:37-38, 63-69).
//
// Mapping: one thread per (point, level); blockIdx.y = level so dense/hash and the table base are
// wave-uniform (SGPRs).  A wave covers 64 consecutive points = consecutive samples of one ray, so
// coarse-level corners are shared inside a wave and served by L1/L2; the 8 corner fetches of a
// thread are independent loads kept in flight together.  F=4 levels fetch one 16-byte vector per
// corner.  The result is written feature-major ([L*F][n]) so that both these stores and the MFMA
// kernel's B-operand loads are fully coalesced.
#include "rc_dev_grid.h"

using namespace rcdev;

namespace {

template <int F, bool JAC>
__device__ __forceinline__ void hashgrid_point(const RcGridDev& g, const float* __restrict__ pts, int soa_in,
                                               const int32_t* __restrict__ src, int64_t n_src, int64_t n,
                                               float* __restrict__ out, int feature_major, int64_t ldo,
                                               float contract_radius, float* __restrict__ jac, int64_t p, int l) {
  if (p >= n) return;
  const int64_t q = src ? (int64_t)src[p] : p;
  float x, y, z;
  if (soa_in) {
    x = pts[q]; y = pts[n_src + q]; z = pts[2 * n_src + q];
  } else {
    x = pts[3 * q]; y = pts[3 * q + 1]; z = pts[3 * q + 2];
  }
  if (contract_radius > 0.0f) contract3(x, y, z, contract_radius);

  const RcGridLevel L = g.lvl[l];
  const float lo = -g.bbox, hi = g.bbox;
  const float N = (float)L.size;
  float acc[F];
  float jacc[JAC ? 3 * F : 1];
  grid_level<F, JAC>(L, g.bbox, x, y, z, acc, jacc);

  const int LF = g.num_levels * F;
#pragma unroll
  for (int f = 0; f < F; ++f) {
    const float v = acc[f] * g.precondition;
    if (feature_major) out[(int64_t)(l * F + f) * ldo + p] = v;
    else out[p * LF + l * F + f] = v;
  }
  if constexpr (JAC) {
    // Jacobian w.r.t. the *contracted* coordinate (x, y, z order), feature-major:
    // jac[(axis * LF + l*F + f) * ldo + p];  d loc / d warped = N / (hi - lo).
    const float s = g.precondition * N / (hi - lo);
#pragma unroll
    for (int f = 0; f < F; ++f) {
      // dense levels: loc = (z, y, x); hash levels: loc = (x, y, z)
      const float gx = L.dense ? jacc[2 * F + f] : jacc[0 * F + f];
      const float gy = jacc[1 * F + f];
      const float gz = L.dense ? jacc[0 * F + f] : jacc[2 * F + f];
      jac[(int64_t)(0 * LF + l * F + f) * ldo + p] = gx * s;
      jac[(int64_t)(1 * LF + l * F + f) * ldo + p] = gy * s;
      jac[(int64_t)(2 * LF + l * F + f) * ldo + p] = gz * s;
    }
  }
}

template <int F, bool JAC>
__global__ __launch_bounds__(256) void k_hashgrid_fwd(RcGridDev g, const float* __restrict__ pts, int soa_in,
                                                        const int32_t* __restrict__ src, int64_t n_src,
                                                        int64_t n, float* __restrict__ out, int feature_major,
                                                        int64_t ldo, float contract_radius,
                                                        float* __restrict__ jac) {
  hashgrid_point<F, JAC>(g, pts, soa_in, src, n_src, n, out, feature_major, ldo, contract_radius, jac,
                         (int64_t)blockIdx.x * 256 + threadIdx.x, blockIdx.y);
}

// Two F = 4 grids (any geometry) at the same row-major points in one launch: blockIdx.z picks the grid.  (The material and
// the light grid at the 1024 shading points of a material step: two 8-us launches on two streams before.)
__global__ __launch_bounds__(256) void k_hashgrid_two(RcGridDev ga, RcGridDev gb, const float* __restrict__ pts, int64_t n,
                                                        float* __restrict__ out_a, float* __restrict__ out_b,
                                                        float contract_radius) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (blockIdx.z == 0) { if ((int)blockIdx.y < ga.num_levels) hashgrid_point<4, false>(ga, pts, 0, nullptr, n, n, out_a, 0, 32, contract_radius, nullptr, p, blockIdx.y); }
  else { if ((int)blockIdx.y < gb.num_levels) hashgrid_point<4, false>(gb, pts, 0, nullptr, n, n, out_b, 0, 32, contract_radius, nullptr, p, blockIdx.y); }
}

// Two F = 4 grids of identical level geometry looked up at the same points in ONE pass over their interleaved tables
// (rc_api.hip build_fused_tables: hashed levels [density 16 B | appearance 16 B] per entry, dense levels as cell tables
// of such pairs): lanes j and j + 32 of a wave take the two halves of every pair of point j, so one 32-byte read serves
// both grids -- half the sector requests of two k_hashgrid_fwd<4> passes (the picked samples of the resampling pass:
// density grid of the last level + appearance grid).  Same index, weights and corner order: bitwise the same features.
__global__ __launch_bounds__(256) void k_hashgrid_pair(RcGridDev g, RcPairTables pt, const float* __restrict__ pts,
                                                         const int32_t* __restrict__ src, int64_t n_src, int64_t n,
                                                         float* __restrict__ out_a, float* __restrict__ out_b, int64_t ldo,
                                                         float contract_radius) {
  const int lane = threadIdx.x & 63, j = lane & 31, which = lane >> 5;
  const int64_t p = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 32 + j;
  const int l = blockIdx.y;
  if (p >= n) return;
  const int64_t q = src ? (int64_t)src[p] : p;
  float x = pts[q], y = pts[n_src + q], z = pts[2 * n_src + q];
  if (contract_radius > 0.0f) contract3(x, y, z, contract_radius);
  const RcGridLevel L = g.lvl[l];
  Corners<4> C;
  grid_fetch<4, true, 2, true>(pt.t[l] + 4 * which, L.size, L.mask, 0u, L.dense != 0, unit_box(g.bbox, x), unit_box(g.bbox, y),
                               unit_box(g.bbox, z), C);
  float acc[4], jd[1];
  grid_combine<4, false>(C, acc, jd);
  float* out = which ? out_b : out_a;
#pragma unroll
  for (int f = 0; f < 4; ++f) out[(int64_t)(l * 4 + f) * ldo + p] = acc[f] * g.precondition;
}

// Cell records of a hashed F = 1 level (rc_internal.h rc_launch_build_hrec): one thread per (record, corner).
__global__ __launch_bounds__(256) void k_build_hrec(const float* __restrict__ table, int N, uint32_t mask, int F, float* __restrict__ dst) {
  const int64_t M = N + 1, total = M * M * M * 8;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i & 7);
  const int64_t r = i >> 3;
  const int qz = (int)(r % M), qy = (int)((r / M) % M), qx = (int)(r / (M * M));
  // int32 -> uint32 wraparound hash of the corner (grid_utils.py:99-111), as grid_fetch computes it
  const uint32_t px = (uint32_t)(qx - 1 + ((c >> 2) & 1)), py = (uint32_t)(qy - 1 + ((c >> 1) & 1)), pz = (uint32_t)(qz - 1 + (c & 1));
  const uint32_t e = (px ^ (py * kPi2) ^ (pz * kPi3)) & mask;
  for (int f = 0; f < F; ++f) dst[i * F + f] = table[(size_t)e * F + f];
}

}  // namespace

void rc_launch_build_hrec(const float* table, int N, uint32_t mask, int F, float* dst, hipStream_t stream) {
  const int64_t total = (int64_t)(N + 1) * (N + 1) * (N + 1) * 8;
  hipLaunchKernelGGL(k_build_hrec, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, table, N, mask, F, dst);
}

void rc_launch_hashgrid_src(const RcGridDev& g, const float* points, int soa_in, const int32_t* src, int64_t n_src,
                            int64_t n, float* out, int feature_major, int64_t ldo, float contract_radius,
                            float* jac_out, hipStream_t stream) {
  if (n <= 0) return;
  dim3 grid((unsigned)((n + 255) / 256), (unsigned)g.num_levels);
  dim3 block(256);
  if (g.num_features == 4) {
    if (jac_out)
      hipLaunchKernelGGL((k_hashgrid_fwd<4, true>), grid, block, 0, stream, g, points, soa_in, src, n_src, n, out,
                         feature_major, ldo, contract_radius, jac_out);
    else
      hipLaunchKernelGGL((k_hashgrid_fwd<4, false>), grid, block, 0, stream, g, points, soa_in, src, n_src, n, out,
                         feature_major, ldo, contract_radius, jac_out);
  } else {
    if (jac_out)
      hipLaunchKernelGGL((k_hashgrid_fwd<1, true>), grid, block, 0, stream, g, points, soa_in, src, n_src, n, out,
                         feature_major, ldo, contract_radius, jac_out);
    else
      hipLaunchKernelGGL((k_hashgrid_fwd<1, false>), grid, block, 0, stream, g, points, soa_in, src, n_src, n, out,
                         feature_major, ldo, contract_radius, jac_out);
  }
}

void rc_launch_hashgrid(const RcGridDev& g, const float* points, int soa_in, int64_t n, float* out,
                        int feature_major, int64_t ldo, float contract_radius, float* jac_out,
                        hipStream_t stream) {
  rc_launch_hashgrid_src(g, points, soa_in, nullptr, n, n, out, feature_major, ldo, contract_radius, jac_out,
                         stream);
}

// g: the level geometry both grids share (fused_geometry_ok); feature-major outputs [L * 4][ldo]
void rc_launch_hashgrid_pair(const RcGridDev& g, const RcPairTables& pt, const float* points_soa, const int32_t* src,
                             int64_t n_src, int64_t n, float* out_a, float* out_b, int64_t ldo, float contract_radius,
                             hipStream_t stream) {
  if (n <= 0) return;
  dim3 grid((unsigned)((n + 127) / 128), (unsigned)g.num_levels), block(256);
  hipLaunchKernelGGL(k_hashgrid_pair, grid, block, 0, stream, g, pt, points_soa, src, n_src, n, out_a, out_b, ldo, contract_radius);
}

void rc_launch_hashgrid_two(const RcGridDev& ga, const RcGridDev& gb, const float* points, int64_t n, float* out_a, float* out_b,
                            float contract_radius, hipStream_t stream) {
  if (n <= 0) return;
  if (ga.num_features != 4 || gb.num_features != 4) {
    rc_launch_hashgrid(ga, points, 0, n, out_a, 0, ga.num_levels * ga.num_features, contract_radius, nullptr, stream);
    rc_launch_hashgrid(gb, points, 0, n, out_b, 0, gb.num_levels * gb.num_features, contract_radius, nullptr, stream);
    return;
  }
  const int levels = ga.num_levels > gb.num_levels ? ga.num_levels : gb.num_levels;
  hipLaunchKernelGGL(k_hashgrid_two, dim3((unsigned)((n + 255) / 256), (unsigned)levels, 2u), dim3(256), 0, stream, ga, gb, points, n,
                     out_a, out_b, contract_radius);
}
