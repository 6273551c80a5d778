// Material stage: per-shading-point heads, BRDF importance sampling of secondary rays and the
// Monte-Carlo BRDF integration against the radiance cache (gfx950).
//
// Replaces (reference file:line):
//   MaterialMLP._predict_material_and_feature / _get_microfacet_material  internal/material.py:2073-2123, 1290-1322
//   LightMLP.predict_lighting / get_vmfs                                  internal/light_sampler.py:135-214
//   get_rotation_matrix, CosineSampler, MicrofacetSampler, LightSampler,
//     sample_vmf, eval_vmf, importance_sample_rays (power-heuristic MIS),
//     get_secondary_rays                                                  internal/inverse_render/render_utils.py:145-168,
//                                                                         417-546, 722-1056, 1335-1490
//   get_lobe (Disney-GGX), integrate_reflect_rays                         internal/inverse_render/render_utils.py:566-695, 1102-1193
//   MaterialMLP.get_outgoing_radiance(_helper), integration strategy      internal/material.py:1352-1565, 1684-1864, 2705-2808
//   MaterialIntegrator composite of the one filtered sample, _handle_brdf_pass  internal/models.py:1531-1694, 1845-1912
//
// These are small element-wise / per-point kernels (a few MFLOP per batch): one wavefront per shading
// point, lanes = secondary samples or vMF lobes; the heavy part of the stage is the batched secondary
// trace, which re-enters the cache kernels (rc_api.hip) on R*K rays.
#include "rc_internal.h"

namespace {

constexpr float kPi = 3.14159265358979323846f;
constexpr float kDenomEps = 1e-5f;   // render_utils.DENOMINATOR_EPS

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}
__device__ __forceinline__ float wmax(float v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v = fmaxf(v, __shfl_xor(v, d, 64));
  return v;
}
__device__ __forceinline__ float sigmoidf(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float softplusf(float x) { return fmaxf(x, 0.0f) + log1pf(expf(-fabsf(x))); }

struct V3 { float x, y, z; };
__device__ __forceinline__ V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

// render_utils.get_rotation_matrix (y_up=False): columns (new_x, new_y, normal)
struct Frame { V3 x, y, z; };
__device__ __forceinline__ Frame make_frame(V3 n) {
  const V3 up = fabsf(n.z) < 0.9f ? V3{0.0f, 0.0f, 1.0f} : V3{0.0f, 1.0f, 0.0f};
  V3 nx = cross(up, n);
  float l = sqrtf(dot(nx, nx)) + 1e-10f;
  nx = {nx.x / l, nx.y / l, nx.z / l};
  V3 ny = cross(n, nx);
  l = sqrtf(dot(ny, ny)) + 1e-10f;
  ny = {ny.x / l, ny.y / l, ny.z / l};
  return {nx, ny, n};
}
// global_to_local: d0 * R[0,:] + d1 * R[1,:] + d2 * R[2,:] with R[i,:] = (x_i, y_i, z_i)
__device__ __forceinline__ V3 to_local(V3 d, const Frame& f) {
  return {d.x * f.x.x + d.y * f.x.y + d.z * f.x.z, d.x * f.y.x + d.y * f.y.y + d.z * f.y.z,
          d.x * f.z.x + d.y * f.z.y + d.z * f.z.z};
}
// local_to_global: d0 * R[:,0] + d1 * R[:,1] + d2 * R[:,2]
__device__ __forceinline__ V3 to_global(V3 d, const Frame& f) {
  return {d.x * f.x.x + d.y * f.y.x + d.z * f.z.x, d.x * f.x.y + d.y * f.y.y + d.z * f.z.y,
          d.x * f.x.z + d.y * f.y.z + d.z * f.z.z};
}
__device__ __forceinline__ V3 ir_normalize(V3 v) {           // inverse_render/math.normalize
  const float l = sqrtf(1e-10f + dot(v, v));
  return {v.x / l, v.y / l, v.z / l};
}
__device__ __forceinline__ V3 l2_normalize(V3 v) {           // ref_utils.l2_normalize
  const float dsq = dot(v, v);
  const float l = sqrtf(fmaxf(RC_TINY, dsq));
  if (dsq < RC_TINY) return {0.0f, 0.0f, 0.0f};
  return {v.x / l, v.y / l, v.z / l};
}
__device__ __forceinline__ float ggx_d(float c, float a) {
  const float t = c * c * (a * a - 1.0f) + 1.0f;
  return (a * a) / fmaxf(RC_EPS, kPi * (t * t));
}

// ---------------------------------------------------------------------------------------------
// Material head: grid features (32) -> Dense 128 -> Dense 10 -> microfacet parameters
// ---------------------------------------------------------------------------------------------
// One workgroup takes MH_PTS points: thread t keeps column t of the first layer in registers, so the 16 KB of w0 is
// read once per MH_PTS points instead of once per point.  Sums run in the same order as the one-point form.
// MH_PTS: 16 for large batches; 4 for the 1024 shading points of a material step (256 workgroups instead of 64: that
// call sits on the step's critical path and is latency, not traffic).
template <int MH_PTS>
__device__ __forceinline__ void material_head_block(const RcMatHeadArgs& a, int64_t block) {
  __shared__ float s_feat[MH_PTS][32], s_h[MH_PTS][128], s_out[MH_PTS][10];
  const int64_t p0 = block * MH_PTS;
  const int np = (int)((a.n - p0) < MH_PTS ? (a.n - p0) : MH_PTS);
  const int t = threadIdx.x;
  for (int e = t; e < np * 32; e += 128) s_feat[e >> 5][e & 31] = a.feat[p0 * 32 + e];
  float w[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) w[i] = a.w0[i * 128 + t];
  const float b0 = a.b0[t];
  __syncthreads();
  for (int q = 0; q < np; ++q) {
    float acc = 0.0f;
#pragma unroll
    for (int i = 0; i < 32; ++i) acc = acc + s_feat[q][i] * w[i];
    s_h[q][t] = acc + b0;
  }
  __syncthreads();
  for (int e = t; e < np * 10; e += 128) {
    const int q = e / 10, c = e - q * 10;
    float o = 0.0f;
    for (int j = 0; j < 128; ++j) o = o + s_h[q][j] * a.w1[j * 10 + c];
    s_out[q][c] = o + a.b1[c];
  }
  __syncthreads();
  if (t < np) {
    float* m = a.mat + (p0 + t) * RC_MAT_CH;
    const float* so = s_out[t];
    const float r0 = a.min_roughness * a.min_roughness;
    m[0] = sigmoidf(so[0] - 1.0f); m[1] = sigmoidf(so[1] - 1.0f); m[2] = sigmoidf(so[2] - 1.0f);   // albedo
    m[3] = sigmoidf(so[6] - 1.0f) * (1.0f - r0) + r0;                                               // roughness
    m[4] = sigmoidf(so[8] + 0.0f);                                                                  // metalness
  }
}
template <int MH_PTS>
__global__ __launch_bounds__(128) void k_material_head(RcMatHeadArgs a) { material_head_block<MH_PTS>(a, blockIdx.x); }

// Composite of the material-only pass over all samples (models.py:1845-1912): sum_s w_s * mat_s.
__global__ void k_material_composite_all(int64_t n, int S, const float* weights, const float* mat, float* out_albedo,
                                         float* out_rough, float* out_metal, float* out_f0, float f0) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  float acc[RC_MAT_CH] = {0, 0, 0, 0, 0};
  float wsum_ = 0.0f;
  for (int s = 0; s < S; ++s) {
    const float w = weights[r * S + s];
    wsum_ += w * f0;
    for (int c = 0; c < RC_MAT_CH; ++c) acc[c] += w * mat[(r * S + s) * RC_MAT_CH + c];
  }
  if (out_albedo) { out_albedo[3 * r] = acc[0]; out_albedo[3 * r + 1] = acc[1]; out_albedo[3 * r + 2] = acc[2]; }
  if (out_rough) out_rough[r] = acc[3];
  if (out_metal) out_metal[r] = acc[4];
  if (out_f0) out_f0[r] = wsum_;
}

// ---------------------------------------------------------------------------------------------
// Light head: grid features (32) -> 64 -> 64 -> 640 -> 128 vMF lobes (normalised mean, kappa,
// softmax weight, normalisation kappa / (4 pi sinh kappa))
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void light_head_point(const RcLightHeadArgs& a, int64_t p) {
  __shared__ float s_x[32], s_h0[64], s_h1[64], s_p[640], s_red[128];
  const int t = threadIdx.x;
  if (t < 32) s_x[t] = a.feat[p * 32 + t];
  __syncthreads();
  if (t < 64) {
    float acc = 0.0f;
#pragma unroll
    for (int i = 0; i < 32; ++i) acc = acc + s_x[i] * a.w0[i * 64 + t];
    s_h0[t] = fmaxf(acc + a.b0[t], 0.0f);
  }
  __syncthreads();
  if (t < 64) {
    float acc = 0.0f;
#pragma unroll 16
    for (int i = 0; i < 64; ++i) acc = acc + s_h0[i] * a.w1[i * 64 + t];
    s_h1[t] = fmaxf(acc + a.b1[t], 0.0f);
  }
  __syncthreads();
  {
    // the thread's five outputs side by side: five independent load / FMA chains per step instead of one (each
    // output still adds its 64 products in the same order)
    float acc[5] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll 8
    for (int i = 0; i < 64; ++i) {
      const float hv = s_h1[i];
#pragma unroll
      for (int k = 0; k < 5; ++k) acc[k] = acc[k] + hv * a.w2[i * 640 + t + 128 * k];
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) s_p[t + 128 * k] = acc[k] + a.b2[t + 128 * k];
  }
  __syncthreads();
  // lobe t (128 lobes): get_vmfs (light_sampler.py:135-160) then LightSampler's l2_normalize / softmax
  const float* q = &s_p[t * 5];
  const float px = a.pts[3 * p], py = a.pts[3 * p + 1], pz = a.pts[3 * p + 2];
  const float* nz = a.noise + (p * 128 + t) * 3;
  V3 m = {q[0] * a.vmf_scale + 0.0f + nz[0] * a.vmf_scale / 2.0f - px, q[1] * a.vmf_scale + 0.0f + nz[1] * a.vmf_scale / 2.0f - py,
          q[2] * a.vmf_scale + 0.0f + nz[2] * a.vmf_scale / 2.0f - pz};
  m = l2_normalize(m);
  const float kappa = fminf(softplusf(q[3] + 1.0f), 50.0f);
  const float logit = fmaxf(q[4] + 1.0f, -50.0f);
  // softmax over the 128 lobes
  s_red[t] = logit;
  __syncthreads();
  for (int d = 64; d >= 1; d >>= 1) { if (t < d) s_red[t] = fmaxf(s_red[t], s_red[t + d]); __syncthreads(); }
  const float mx = s_red[0];
  __syncthreads();
  const float e = expf(logit - mx);
  s_red[t] = e;
  __syncthreads();
  for (int d = 64; d >= 1; d >>= 1) { if (t < d) s_red[t] = s_red[t] + s_red[t + d]; __syncthreads(); }
  const float wgt = e / s_red[0];
  float* o = a.vmf + (p * 128 + t) * RC_VMF_CH;
  o[0] = m.x; o[1] = m.y; o[2] = m.z; o[3] = kappa; o[4] = wgt;
  a.vmf_logit[p * 128 + t] = logit;
}
__global__ __launch_bounds__(128) void k_light_head(RcLightHeadArgs a) { light_head_point(a, blockIdx.x); }

// The two heads at the shading points of a material step in ONE launch: blocks [0, mat_blocks) run the material head
// (4 points each), the others the light head (one point each).  Both read only the shading point and feed the BRDF
// sampler; as two launches on two streams the sampler waited 12 us for the cross-stream event behind them.
__global__ __launch_bounds__(128) void k_shading_heads(RcMatHeadArgs m, RcLightHeadArgs l, int mat_blocks) {
  if ((int)blockIdx.x < mat_blocks) material_head_block<4>(m, blockIdx.x);
  else light_head_point(l, (int64_t)blockIdx.x - mat_blocks);
}

// eval_vmf (render_utils.py:1335-1347) with inverse_render.math.safe_exp = exp(min(x, 80))
__device__ __forceinline__ float eval_vmf(V3 x, V3 mean, float kappa) {
  if (kappa <= RC_EPS) return 1.0f / (4.0f * kPi);
  return kappa * expf(fminf(kappa * dot(x, mean), 80.0f)) / (4.0f * kPi * sinhf(kappa));
}

// ---------------------------------------------------------------------------------------------
// BRDF importance sampling: one wave per shading point, lane = secondary sample.
// lanes [0, Ks): GGX microfacet (specular pass); [Ks, Ks+Kc): cosine; [Ks+Kc, Ks+Kd): vMF light.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_brdf_sample(RcBrdfSampleArgs a) {
  __shared__ float s_vmf[4][128 * RC_VMF_CH];
  __shared__ float s_den[4][128];        // 4 pi sinh(kappa) per lobe: the direction-independent part of eval_vmf
  __shared__ float s_qdir[4][128 * 3];   // query directions of the mixture pdf (at most 2 Kd <= 128 per point)
  __shared__ float s_qpdf[4][128];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int64_t r = (int64_t)blockIdx.x * 4 + wave;
  const bool ok = r < a.n;
  if (!ok) r = a.n - 1;
  // Every global read of the point that depends on nothing computed here is issued up front, in one batch: the lobe
  // table (10 values per lane), the lobe logits + noise, the lane's own random inputs.  (As a loop of "load, store to
  // LDS" the table alone was ten dependent round trips -- the loads were not hoisted over the LDS stores -- and the
  // random inputs sat behind the barriers below: ~12 of this kernel's 37 us.)
  const int Ks = a.Ks, Kd = a.Kd, Kc = a.Kc, K = Ks + Kd;
  const int kd = lane - Ks, kl = kd - Kc, Kl = Kd - Kc;
  static_assert(128 * RC_VMF_CH == 10 * 64, "lobe table: ten values per lane");
  float tab[10];
#pragma unroll
  for (int k = 0; k < 10; ++k) tab[k] = a.vmf[r * 128 * RC_VMF_CH + lane + 64 * k];
  float lg[2] = {0.0f, 0.0f}, gm[2] = {0.0f, 0.0f};
  if (!a.vmf_lobe) {
#pragma unroll
    for (int q = 0; q < 2; ++q) { lg[q] = a.vmf_logit[r * 128 + lane + 64 * q]; gm[q] = a.vmf_lobe_gumbel[r * 128 + lane + 64 * q]; }
  }
  float ru1 = 0.0f, ru2 = 0.0f, rtmp = 0.0f;       // (u1, u2) of a GGX / cosine lane, (v0, v1) and tmp of a vMF lane
  if (lane < Ks) { ru1 = a.spec_u1[r * Ks + lane]; ru2 = a.spec_u2[r * Ks + lane]; }
  else if (lane < K && kd < Kc) { ru1 = a.cos_u1[r * Kc + kd]; ru2 = a.cos_u2[r * Kc + kd]; }
  else if (lane < K) { ru1 = a.vmf_v[(r * Kl + kl) * 2]; ru2 = a.vmf_v[(r * Kl + kl) * 2 + 1]; rtmp = a.vmf_tmp[r * Kl + kl]; }
#pragma unroll
  for (int k = 0; k < 10; ++k) s_vmf[wave][lane + 64 * k] = tab[k];
  __syncthreads();
  for (int j = lane; j < 128; j += 64) s_den[wave][j] = 4.0f * kPi * sinhf(s_vmf[wave][j * RC_VMF_CH + 3]);
  __syncthreads();
  const float* vm = s_vmf[wave];
  const float* den = s_den[wave];
  // sample_vmf_vars (render_utils.py:1357-1372): ONE lobe per point, given or drawn here as argmax_j(logit_j + g_j) on
  // the RAW logits of the light head, exactly what jax.random.categorical(key, logits=vars[2]) does (render_utils.py:
  // 1358-1362) -- not on log(softmax), whose rounding and underflow clamp could flip a near-tie
  int lobe;
  if (a.vmf_lobe) {
    lobe = a.vmf_lobe[r];
  } else {
    float key = -INFINITY;
    int best = lane;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int jj = lane + 64 * q;
      const float kq = lg[q] + gm[q];
      if (kq > key) { key = kq; best = jj; }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      const float ok = __shfl_xor(key, d, 64);
      const int ob = __shfl_xor(best, d, 64);
      if (ok > key || (ok == key && ob < best)) { key = ok; best = ob; }
    }
    lobe = best;
  }
  lobe = min(max(lobe, 0), 127);
  const V3 nrm = {a.nrm[3 * r], a.nrm[3 * r + 1], a.nrm[3 * r + 2]};
  const V3 pt = {a.pts[3 * r], a.pts[3 * r + 1], a.pts[3 * r + 2]};
  const V3 gview = {-a.viewdirs[3 * r], -a.viewdirs[3 * r + 1], -a.viewdirs[3 * r + 2]};
  const Frame f = make_frame(nrm);
  const V3 lv = to_local(gview, f);
  const float alpha = a.mat[r * RC_MAT_CH + 3];
  // Mixture pdf of the 128 lobes at a direction: the (direction, lobe) terms of ALL the point's queries are spread over
  // the wave -- every lane takes lobes lane and lane + 64 of each query, a butterfly adds them up -- instead of each
  // diffuse lane walking the 128 lobes once or twice by itself.  Queries: [0, Kd) the MIS light pdf at each diffuse
  // sample's direction, [Kd, Kd + Kl) the sampling pdf of each vMF sample.
  float* qdir = s_qdir[wave];
  float* qpdf = s_qpdf[wave];
  V3 ld = {0.0f, 0.0f, 1.0f};
  float pdf = 0.0f, weight = 0.0f, own_pdf = 0.0f;
  const bool live = lane < K && ok, diffuse = live && lane >= Ks;
  const bool vmf_lane = diffuse && kd >= Kc;
  if (live) {
    if (lane < Ks) {
      // MicrofacetSampler.sample_directions (render_utils.py:501-531); single sampler -> weight 1
      const float u1 = ru1, u2 = ru2;
      const float tan2 = alpha * alpha * u1 / fmaxf(1.0f - u1, RC_EPS);
      const float cost = 1.0f / sqrtf(fmaxf(1.0f + tan2, RC_EPS));
      const float sint = sqrtf(fmaxf(kDenomEps, 1.0f - cost * cost));
      const float phi = u2 * 2.0f * kPi - kPi;
      const V3 h = {sint * cosf(phi), sint * sinf(phi), cost};
      const float npdf = fmaxf(ggx_d(cost, alpha) * fabsf(cost), 0.0f);
      const float wn = dot(lv, h);
      V3 d = {2.0f * wn * h.x - lv.x, 2.0f * wn * h.y - lv.y, 2.0f * wn * h.z - lv.z};
      pdf = npdf * (1.0f / fmaxf(4.0f * wn, RC_EPS));
      if (wn <= 0.0f) pdf = 0.0f;
      pdf = fmaxf(pdf, 0.0f);
      ld = ir_normalize(d);
      weight = 1.0f;
    } else {
      if (kd < Kc) {
        // CosineSampler (render_utils.py:425-433)
        const float u1 = ru1, u2 = ru2;
        const float rr = sqrtf(u1), phi = u2 * 2.0f * kPi - kPi;
        const float x = rr * cosf(phi), y = rr * sinf(phi);
        const float z = sqrtf(fmaxf(kDenomEps, 1.0f - x * x - y * y));
        ld = {x, y, z};
        own_pdf = fmaxf(z / kPi, 0.0f);
      } else {
        // LightSampler -> sample_vmf (render_utils.py:1390-1428): all directions from ONE lobe per point
        const float* q = vm + lobe * RC_VMF_CH;
        const V3 mean = {q[0], q[1], q[2]};
        const float kappa = q[3];
        const V3 tv = l2_normalize(V3{-mean.y, mean.x, 0.0f});
        const V3 bv = l2_normalize(cross(mean, tv));
        float v0 = ru1, v1 = ru2;
        {
          const float dsq = v0 * v0 + v1 * v1, l = sqrtf(fmaxf(RC_TINY, dsq));
          if (dsq < RC_TINY) { v0 = 0.0f; v1 = 0.0f; } else { v0 = v0 / l; v1 = v1 / l; }
        }
        const float tmp = rtmp;
        const float arg = tmp + (1.0f - tmp) * expf(-2.0f * kappa);
        const float w = 1.0f + (1.0f / fmaxf(kappa, RC_EPS)) * logf(fminf(fmaxf(arg, RC_TINY), RC_FMAX));
        const float s = sqrtf(fminf(fmaxf(1.0f - w * w, 0.0f), RC_FMAX));
        const V3 loc = {s * v0, s * v1, w};
        // rotmat = stack([t, b, mean], axis=-1) @ loc
        const V3 g = {tv.x * loc.x + bv.x * loc.y + mean.x * loc.z, tv.y * loc.x + bv.y * loc.y + mean.y * loc.z,
                      tv.z * loc.x + bv.z * loc.y + mean.z * loc.z};
        qdir[3 * (Kd + kl)] = g.x; qdir[3 * (Kd + kl) + 1] = g.y; qdir[3 * (Kd + kl) + 2] = g.z;
        ld = to_local(g, f);
      }
      const V3 gl = to_global(ld, f);
      qdir[3 * kd] = gl.x; qdir[3 * kd + 1] = gl.y; qdir[3 * kd + 2] = gl.z;
    }
  }
  __syncthreads();
  {
    // eval_vmf (render_utils.py:1335-1355) with its denominator 4 pi sinh(kappa) from the per-lobe table
    float m[2][3], kap[2], wgt[2], dn[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const float* v = vm + (lane + 64 * q) * RC_VMF_CH;
      m[q][0] = v[0]; m[q][1] = v[1]; m[q][2] = v[2]; kap[q] = v[3]; wgt[q] = v[4]; dn[q] = den[lane + 64 * q];
    }
    const int NQ = ok ? Kd + Kl : 0;
    for (int qi = 0; qi < NQ; ++qi) {
      const V3 gd = {qdir[3 * qi], qdir[3 * qi + 1], qdir[3 * qi + 2]};
      float sacc = 0.0f;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const float e = kap[q] <= RC_EPS ? 1.0f / (4.0f * kPi)
                                         : kap[q] * expf(fminf(kap[q] * dot(gd, V3{m[q][0], m[q][1], m[q][2]}), 80.0f)) / dn[q];
        sacc = sacc + wgt[q] * e;
      }
      sacc = wsum(sacc);
      if (lane == 0) qpdf[qi] = fmaxf(sacc, 0.0f);
    }
  }
  __syncthreads();
  if (live) {
    if (diffuse) {
      // power heuristic over (cosine, light), one unit of each (render_utils.py:817-853)
      if (vmf_lane) own_pdf = qpdf[Kd + kl];
      float pc = ld.z / kPi;
      if (ld.z < 0.0f) pc = 0.0f;
      pc = fmaxf(pc, 0.0f);
      const float pl = qpdf[kd];
      const float denom = fmaxf(pc * pc + pl * pl, kDenomEps);
      pdf = fmaxf(own_pdf, 0.0f);
      weight = (pdf * pdf) / denom * 2.0f;
    }
    if (!(ld.z > 0.0f)) weight = 0.0f;                         // material.py:1756-1761
    const V3 g = to_global(ld, f);
    // secondary-ray batch: [specular block n*Ks | diffuse block n*Kd], ray-major inside a block
    const int64_t idx = lane < Ks ? r * Ks + lane : a.n * Ks + r * Kd + (lane - Ks);
    a.sec_origins[3 * idx] = pt.x + nrm.x * a.normal_eps;
    a.sec_origins[3 * idx + 1] = pt.y + nrm.y * a.normal_eps;
    a.sec_origins[3 * idx + 2] = pt.z + nrm.z * a.normal_eps;
    a.sec_dirs[3 * idx] = g.x; a.sec_dirs[3 * idx + 1] = g.y; a.sec_dirs[3 * idx + 2] = g.z;
    a.sec_near[idx] = a.near; a.sec_far[idx] = a.far;
    a.sec_lights[3 * idx] = a.lights ? a.lights[3 * r] : 0.0f;
    a.sec_lights[3 * idx + 1] = a.lights ? a.lights[3 * r + 1] : 0.0f;
    a.sec_lights[3 * idx + 2] = a.lights ? a.lights[3 * r + 2] : 0.0f;
    float* sm = a.samples + (r * K + lane) * RC_SMP_CH;
    sm[0] = ld.x; sm[1] = ld.y; sm[2] = ld.z; sm[3] = pdf; sm[4] = weight;
  }
  if (lane == 0 && ok) { a.local_view[3 * r] = lv.x; a.local_view[3 * r + 1] = lv.y; a.local_view[3 * r + 2] = lv.z; }
}

// ---------------------------------------------------------------------------------------------
// Monte-Carlo BRDF integration + composite of the one filtered sample per ray
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_material_integrate(RcMatIntegrateArgs a) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int64_t r = (int64_t)blockIdx.x * 4 + wave;
  const bool ok = r < a.n;
  if (!ok) r = a.n - 1;
  const int Ks = a.Ks, Kd = a.Kd, K = Ks + Kd;
  const float* m = a.mat + r * RC_MAT_CH;
  const float albedo[3] = {m[0], m[1], m[2]};
  const float rough = m[3], metal = m[4];
  const V3 wo = {a.local_view[3 * r], a.local_view[3 * r + 1], a.local_view[3 * r + 2]};
  const bool act = lane < K;
  const bool spec = lane < Ks;
  float ind[3] = {0, 0, 0}, dir[3] = {0, 0, 0}, irr_i[3] = {0, 0, 0}, irr_d[3] = {0, 0, 0};
  float occ = 0.0f;
  if (act) {
    const float* sm = a.samples + (r * K + lane) * RC_SMP_CH;
    const V3 wi = {sm[0], sm[1], sm[2]};
    const float pdf = sm[3];
    float weight = fmaxf(sm[4], 0.0f);
    if (!(wi.z > 0.0f)) weight = 0.0f;
    const float denom = fmaxf(pdf, kDenomEps);
    const int64_t idx = spec ? r * Ks + lane : a.n * Ks + r * Kd + (lane - Ks);
    const float acc = a.sec_acc[idx];
    // get_lobe in the local frame (normal = +z), brdf_correction = 1
    const V3 h = ir_normalize(V3{wi.x + wo.x, wi.y + wo.y, wi.z + wo.z});
    const float n_v = fmaxf(0.0f, wo.z), n_l = fmaxf(0.0f, wi.z), n_h = fmaxf(0.0f, h.z), l_h = fmaxf(0.0f, dot(wi, h));
    const float D = ggx_d(n_h, rough);
    const float k = rough / 2.0f;
    const float G = (n_v / fmaxf(RC_EPS, n_v * (1.0f - k) + k)) * (n_l / fmaxf(RC_EPS, n_l * (1.0f - k) + k));
    // jnp.power(x, 5) with a static integer exponent is lax.integer_pow: x * (x^2)^2, three multiplies (render_utils.py:627)
    const float c1 = fminf(fmaxf(1.0f - l_h, 0.0f), 1.0f), c2 = c1 * c1, c4 = c2 * c2;
    const float c5 = c1 * c4;
    const float dl = fmaxf(0.0f, wi.z) / kPi;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float F0 = albedo[c] * metal + a.f0 * (1.0f - metal);
      const float F = F0 + (1.0f - F0) * c5;
      const float ggx = D * F * G / fmaxf(RC_EPS, 4.0f * n_v);
      const float lambert = n_l * albedo[c] / kPi;
      const float lobe = spec ? ggx * 1.0f * 1.0f : lambert * 1.0f * (1.0f - metal);
      // radiance_cache_fn: max(nan_to_num(rgb), 0); env_map_fn: max(env, 0) * (1 - acc)
      float rin = a.sec_rgb[3 * idx + c];
      if (rin != rin) rin = 0.0f;
      rin = fmaxf(fminf(fmaxf(rin, -RC_FMAX), RC_FMAX), 0.0f);
      float ein = fmaxf(a.sec_env[3 * idx + c], 0.0f) * (1.0f - acc);
      if (ein != ein) ein = 0.0f;
      ein = fminf(fmaxf(ein, -RC_FMAX), RC_FMAX);
      ind[c] = fminf(fmaxf(rin * lobe, 0.0f), a.rgb_max) * weight / denom;
      dir[c] = fminf(fmaxf(ein * lobe, 0.0f), a.rgb_max) * weight / denom;
      irr_i[c] = fminf(fmaxf(rin * dl, 0.0f), a.rgb_max) * weight / denom;
      irr_d[c] = fminf(fmaxf(ein * dl, 0.0f), a.rgb_max) * weight / denom;
    }
    occ = acc;
  }
  // means over the K samples of each pass
  float o_is[3], o_id[3], o_ds[3], o_dd[3], o_irr[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    o_is[c] = wsum(act && spec ? ind[c] : 0.0f) / (float)Ks;
    o_ds[c] = wsum(act && spec ? dir[c] : 0.0f) / (float)Ks;
    o_id[c] = wsum(act && !spec ? ind[c] : 0.0f) / (float)Kd;
    o_dd[c] = wsum(act && !spec ? dir[c] : 0.0f) / (float)Kd;
    const float ii = wsum(act && !spec ? irr_i[c] : 0.0f) / (float)Kd;
    const float id = wsum(act && !spec ? irr_d[c] : 0.0f) / (float)Kd;
    o_irr[c] = (id + ii) * 0.5f;                                           // "irradiance", scale 0.5
  }
  const float occ_s = wsum(act && spec ? occ : 0.0f) / (float)Ks * 0.5f;   // "indirect_occ", scale 0.5
  // acc of the primary ray from the unfiltered weights
  const int S = a.S;
  const float acc_p = wsum(lane < S ? a.weights[r * S + lane] : 0.0f);
  if (lane == 0 && ok) {
    const float w = a.filt_weight[r];
    const float bgw = fmaxf(0.0f, 1.0f - acc_p) * a.bg;
    auto put3 = [&](int id, float x, float y, float z) {
      float* o = a.out.ptr[id];
      if (o) { o[3 * r] = x; o[3 * r + 1] = y; o[3 * r + 2] = z; }
    };
    auto put1 = [&](int id, float x) { if (a.out.ptr[id]) a.out.ptr[id][r] = x; };
    float rgb[3], drgb[3], irgb[3], dif[3], spc[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      rgb[c] = ((o_dd[c] + o_ds[c]) + o_id[c]) + o_is[c];                  // radiance_out sum order (material.py:2727-2735)
      drgb[c] = o_dd[c] + o_ds[c];
      irgb[c] = o_id[c] + o_is[c];
      dif[c] = o_dd[c] + o_id[c];
      spc[c] = o_ds[c] + o_is[c];
    }
    put3(RC_MOUT_RGB, w * rgb[0] + bgw, w * rgb[1] + bgw, w * rgb[2] + bgw);
    put1(RC_MOUT_ACC, acc_p);
    put3(RC_MOUT_DIRECT_RGB, w * drgb[0], w * drgb[1], w * drgb[2]);
    put3(RC_MOUT_INDIRECT_RGB, w * irgb[0], w * irgb[1], w * irgb[2]);
    put3(RC_MOUT_DIFFUSE_RGB, w * dif[0], w * dif[1], w * dif[2]);
    put3(RC_MOUT_SPECULAR_RGB, w * spc[0], w * spc[1], w * spc[2]);
    put3(RC_MOUT_DIRECT_DIFFUSE_RGB, w * o_dd[0], w * o_dd[1], w * o_dd[2]);
    put3(RC_MOUT_DIRECT_SPECULAR_RGB, w * o_ds[0], w * o_ds[1], w * o_ds[2]);
    put3(RC_MOUT_INDIRECT_DIFFUSE_RGB, w * o_id[0], w * o_id[1], w * o_id[2]);
    put3(RC_MOUT_INDIRECT_SPECULAR_RGB, w * o_is[0], w * o_is[1], w * o_is[2]);
    put1(RC_MOUT_INDIRECT_OCC, w * occ_s);
    put3(RC_MOUT_LIGHTING_IRRADIANCE, w * o_irr[0], w * o_irr[1], w * o_irr[2]);
    const float px = a.pts[3 * r], py = a.pts[3 * r + 1], pz = a.pts[3 * r + 2];
    put3(RC_MOUT_MEANS, w * px, w * py, w * pz);
    put3(RC_MOUT_NORMALS_TO_USE, w * a.nrm[3 * r], w * a.nrm[3 * r + 1], w * a.nrm[3 * r + 2]);
    const float ox = a.origins[3 * r], oy = a.origins[3 * r + 1], oz = a.origins[3 * r + 2];
    put1(RC_MOUT_RAY_DISTS, w * sqrtf((ox - px) * (ox - px) + (oy - py) * (oy - py) + (oz - pz) * (oz - pz)));
    if (a.lights) {
      const float lx = a.lights[3 * r], ly = a.lights[3 * r + 1], lz = a.lights[3 * r + 2];
      put1(RC_MOUT_LIGHT_DISTS, w * sqrtf((lx - px) * (lx - px) + (ly - py) * (ly - py) + (lz - pz) * (lz - pz)));
    }
  }
}

}  // namespace

void rc_launch_material_head(const RcMatHeadArgs& a, hipStream_t st) {
  if (a.n <= 0) return;
  if (a.n <= 8192) hipLaunchKernelGGL(k_material_head<4>, dim3((unsigned)((a.n + 3) / 4)), dim3(128), 0, st, a);
  else hipLaunchKernelGGL(k_material_head<16>, dim3((unsigned)((a.n + 15) / 16)), dim3(128), 0, st, a);
}
void rc_launch_material_composite_all(int64_t n, int S, const float* weights, const float* mat, float* out_albedo,
                                      float* out_rough, float* out_metal, float* out_f0, float f0, hipStream_t st) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_material_composite_all, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, S, weights, mat,
                     out_albedo, out_rough, out_metal, out_f0, f0);
}
void rc_launch_shading_heads(const RcMatHeadArgs& m, const RcLightHeadArgs& l, hipStream_t st) {
  if (m.n <= 0 || l.n <= 0) { rc_launch_material_head(m, st); rc_launch_light_head(l, st); return; }
  const int mat_blocks = (int)((m.n + 3) / 4);
  hipLaunchKernelGGL(k_shading_heads, dim3((unsigned)(mat_blocks + l.n)), dim3(128), 0, st, m, l, mat_blocks);
}
void rc_launch_light_head(const RcLightHeadArgs& a, hipStream_t st) {
  if (a.n <= 0) return;
  hipLaunchKernelGGL(k_light_head, dim3((unsigned)a.n), dim3(128), 0, st, a);
}
void rc_launch_brdf_sample(const RcBrdfSampleArgs& a, hipStream_t st) {
  if (a.n <= 0) return;
  hipLaunchKernelGGL(k_brdf_sample, dim3((unsigned)((a.n + 3) / 4)), dim3(256), 0, st, a);
}
void rc_launch_material_integrate(const RcMatIntegrateArgs& a, hipStream_t st) {
  if (a.n <= 0) return;
  hipLaunchKernelGGL(k_material_integrate, dim3((unsigned)((a.n + 3) / 4)), dim3(256), 0, st, a);
}
