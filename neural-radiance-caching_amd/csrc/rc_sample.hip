// Per-ray kernels: proposal resampling, alpha weights, categorical resampling, volume compositing.
// One 64-lane wavefront per ray: lanes are histogram bins / samples, wave scans give the CDF and
// the transmittance, a per-wave LDS slice holds the step function.
//
// Replaces (reference file:line):
//   stepfun.sample_intervals / sample / invert_cdf / integrate_weights   internal/stepfun.py:125-250
//   math.sorted_lookup / sorted_interp                                   internal/math.py:412-457
//   coord.construct_ray_warps, math.power_ladder / inv_power_ladder      internal/coord.py:223-260, math.py:295-341
//   render.cast_rays (means), render.compute_alpha_weights               internal/render.py:26-169
//   the near replacement of secondary rays                               internal/sampling.py:182-205
//   Model.maybe_resample                                                 internal/models.py:193-292
//   VolumeIntegrator / render.volumetric_rendering / weighted_percentile internal/integration.py:112-289,
//                                                                        internal/render.py:172-247, stepfun.py:306-314
#include "rc_dev_sample.h"

using namespace rcdev;

namespace {

__global__ __launch_bounds__(kWavesPerBlock * 64) void k_sample_level(RcSampleArgs a, USpec us, float y_max) {
  __shared__ float lds[kWavesPerBlock][5][kSlots + 3];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int64_t ray = (int64_t)blockIdx.x * kWavesPerBlock + wave;
  const bool ray_ok = ray < a.n_rays;
  if (!ray_ok) ray = a.n_rays - 1;   // keep the wave alive; stores are masked
  float mx, my, mz;
  sample_level_ray(a, us, y_max, ray, ray_ok, &lds[wave][0][0], lane, mx, my, mz);
}

__global__ __launch_bounds__(kWavesPerBlock * 64) void k_sample_intervals(const float* __restrict__ t,
                                                                            const float* __restrict__ logits, int64_t n,
                                                                            int P, int S, const float* jitter,
                                                                            float* __restrict__ out, USpec us) {
  __shared__ float lds[kWavesPerBlock][5][kSlots + 3];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int64_t ray = (int64_t)blockIdx.x * kWavesPerBlock + wave;
  const bool ok = ray < n;
  if (!ok) ray = n - 1;
  float* s_t = lds[wave][0];
  for (int e2 = lane; e2 <= P; e2 += 64) s_t[e2] = t[ray * (P + 1) + e2];
  const float logit = lane < P ? logits[ray * P + lane] : 0.0f;
  const bool hasj = jitter != nullptr;
  const float jit = hasj ? jitter[ray] : 0.0f;
  __syncthreads();
  sample_intervals_wave(logit, P, S, us, hasj, jit, s_t, lds[wave][1], lds[wave][2], lds[wave][3], lds[wave][4], lane);
  if (ok)
    for (int e2 = lane; e2 <= S; e2 += 64) out[ray * (S + 1) + e2] = lds[wave][4][e2];
}

// ---------------------------------------------------------------------------------------------
// Categorical resampling (num_resample == 1)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kWavesPerBlock * 64) void k_resample(RcResampleArgs a) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t ray = (int64_t)blockIdx.x * kWavesPerBlock + wave;
  if (ray >= a.n_rays) return;
  const int S = a.S;
  const bool act = lane < S;
  const float dx = a.directions[3 * ray], dy = a.directions[3 * ray + 1], dz = a.directions[3 * ray + 2];
  const float dnorm = sqrtf(dx * dx + dy * dy + dz * dz);
  const float t0 = act ? a.tdist[ray * (S + 1) + lane] : 0.0f;
  const float t1 = act ? a.tdist[ray * (S + 1) + lane + 1] : 0.0f;
  const float dens = act ? a.density[ray * S + lane] : 0.0f;
  const float w = alpha_weight(dens, t0, t1, dnorm, act, lane);
  if (act) a.weights[ray * S + lane] = w;
  if (a.acc_out) {                                 // render.py:202, as k_composite adds it
    const float acc = wave_sum(w);
    if (lane == 0) a.acc_out[ray] = acc;
  }
  // logits = safe_log(w + 0) * 1 ; probs = softmax (models.py:207-209)
  const float logit = safe_log(w);
  const float m = wave_max(act ? logit : -INFINITY);
  const float e = act ? expf(logit - m) : 0.0f;
  const float p = e / wave_sum(e);
  int ind;
  if (a.inds_in) {
    ind = min(max(a.inds_in[ray], 0), S - 1);     // caller-provided: keep the gathers behind it in range
  } else {
    // jax.random.categorical == argmax(logits + gumbel); first index on ties
    float key = act ? logit + a.gumbel[ray * S + lane] : -INFINITY;
    int best = lane;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      const float ok = __shfl_xor(key, d, 64);
      const int ob = __shfl_xor(best, d, 64);
      if (ok > key || (ok == key && ob < best)) { key = ok; best = ob; }
    }
    ind = best;
  }
  const float wsel = __shfl(w, ind, 64), psel = __shfl(p, ind, 64);
  if (lane == 0) {
    a.inds_out[ray] = ind;
    a.filt_weight[ray] = wsel / (1.0f * psel + 1e-8f);     // models.py:287-289, num_resample = 1
    if (a.src_out) a.src_out[ray] = (int32_t)(ray * S + ind);
  }
  if (a.pts_out && lane < 6) {
    // lanes 0-2: position, lanes 3-5: predicted normal of the picked sample
    const int64_t np = a.n_rays * S, q = ray * S + ind;
    const int c = lane < 3 ? lane : lane - 3;
    const float v = (lane < 3 ? a.means : a.normals)[c * np + q];
    (lane < 3 ? a.pts_out : a.nrm_out)[3 * ray + c] = v;
  }
}

// ---------------------------------------------------------------------------------------------
// Volume compositing
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kWavesPerBlock * 64) void k_composite(RcCompositeArgs a) {
  __shared__ float lds[kWavesPerBlock][2][kSlots + 3];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int64_t ray = (int64_t)blockIdx.x * kWavesPerBlock + wave;
  const bool ray_ok = ray < a.n_rays;
  if (!ray_ok) ray = a.n_rays - 1;
  float* s_t = lds[wave][0];
  float* s_cw = lds[wave][1];
  const int S = a.S;
  const bool act = lane < S;
  const int64_t np = a.n_rays * S, pidx = ray * S + (act ? lane : 0);
  const float dx = a.directions[3 * ray], dy = a.directions[3 * ray + 1], dz = a.directions[3 * ray + 2];
  const float dnorm = sqrtf(dx * dx + dy * dy + dz * dz);
  for (int e2 = lane; e2 <= S; e2 += 64) s_t[e2] = a.tdist[ray * (S + 1) + e2];
  lds_sync<false>();
  const float t0 = act ? s_t[lane] : 1.0f, t1 = act ? s_t[lane + 1] : 1.0f;
  // (loads from always-valid addresses, selects behind them: a `condition ? load : 0` becomes a branch around the load
  // with a wait behind it, one dependent round trip per input)
  const float dens_ld = a.density[pidx];
  const float dens = act ? dens_ld : 0.0f;
  const float wnf = alpha_weight(dens, t0, t1, dnorm, act, lane);     // weights_no_filter
  if (act && ray_ok && a.weights) a.weights[pidx] = wnf;
  const float acc = wave_sum(wnf);                                    // render.py:202

  // Filtered weights: identical to wnf without resampling; with one resampled sample the only
  // non-zero filtered weight sits on the selected index.
  const bool resampled = a.Sf == 1 && a.inds != nullptr;
  int sel = 0;
  float w = wnf;
  if (resampled) {
    sel = a.inds[ray];
    w = (lane == sel) ? a.filt_weight[ray] : 0.0f;
  }
  const bool contrib = act && (!resampled || lane == sel);
  const int64_t nsh = resampled ? a.n_rays : np;
  const int64_t sidx = resampled ? ray : pidx;

  auto shade = [&](int ch) -> float { const float v = a.shade[(int64_t)ch * nsh + sidx]; return contrib ? v : 0.0f; };
  auto store3 = [&](int id, float x, float y, float z) {
    if (lane == 0 && ray_ok && a.out.ptr[id]) {
      a.out.ptr[id][3 * ray] = x; a.out.ptr[id][3 * ray + 1] = y; a.out.ptr[id][3 * ray + 2] = z;
    }
  };
  auto store1 = [&](int id, float x) {
    if (lane == 0 && ray_ok && a.out.ptr[id]) a.out.ptr[id][ray] = x;
  };

  // Every output is a wave reduction.  A caller that asks for the colour components (the plain cache pass) gets all of
  // them from ONE batched butterfly (the exchanges of the 33 sums in flight together); a caller that asks for rgb / acc
  // and perhaps a geometry extra (the batched secondary trace of the material stage) runs only those reductions.  Either
  // way each sum is added up in the same order: same bits.  The branches are uniform (kernel arguments).
  auto want = [&](int id) { return a.out.ptr[id] != nullptr; };
  const float bgw = fmaxf(0.0f, 1.0f - acc) * a.bg;
  const float wc = contrib ? w : 0.0f;
  const bool components = want(RC_OUT_DIRECT_RGB) || want(RC_OUT_INDIRECT_DIFFUSE_RGB) || want(RC_OUT_INDIRECT_SPECULAR_RGB) ||
                          want(RC_OUT_SPECULAR_RGB) || want(RC_OUT_ALBEDO_RGB) || want(RC_OUT_DIFFUSE_RGB) || want(RC_OUT_INDIRECT_RGB);
  const bool want_pos = components || want(RC_OUT_MEANS) || want(RC_OUT_RAY_DISTS) || want(RC_OUT_LIGHT_DISTS);      // uniform
  const bool pos = act && want_pos;
  float mx = 0.0f, my = 0.0f, mz = 0.0f;
  if (want_pos) {
    const float lx = a.means[pidx], ly = a.means[np + pidx], lz = a.means[2 * np + pidx];
    mx = pos ? lx : 0.0f; my = pos ? ly : 0.0f; mz = pos ? lz : 0.0f;
  }
  auto ray_dist = [&]() {
    const float ox = a.origins[3 * ray], oy = a.origins[3 * ray + 1], oz = a.origins[3 * ray + 2];
    return sqrtf((ox - mx) * (ox - mx) + (oy - my) * (oy - my) + (oz - mz) * (oz - mz));
  };
  auto light_dist = [&]() {
    const float lx = a.lights[3 * ray], ly = a.lights[3 * ray + 1], lz = a.lights[3 * ray + 2];
    return sqrtf((lx - mx) * (lx - mx) + (ly - my) * (ly - my) + (lz - mz) * (lz - mz));
  };
  if (components) {
    enum { V_RGB = 0, V_AD = 3, V_IDF = 6, V_IS = 9, V_TINT = 12, V_DIF = 15, V_IND = 18, V_OCC = 21, V_MEAN = 22, V_RD = 25,
           V_LD = 26, V_NP = 27, V_NG = 30, V_COUNT = 33 };
    float v[V_COUNT];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float v_rgb = shade(RC_SH_RGB + c), v_ad = shade(RC_SH_AD + c), v_id = shade(RC_SH_ID + c);
      const float v_is = shade(RC_SH_IS + c), v_t = shade(RC_SH_TINT + c);
      v[V_RGB + c] = w * v_rgb;
      v[V_AD + c] = w * v_ad;
      v[V_IDF + c] = w * v_id;
      v[V_IS + c] = w * v_is;
      v[V_TINT + c] = w * v_t;
      v[V_DIF + c] = w * (v_ad + v_id);       // diffuse_rgb = ambient_diffuse + indirect_diffuse
      v[V_IND + c] = w * (v_id + v_is);       // indirect_rgb = indirect_diffuse + indirect_specular
    }
    v[V_OCC] = wc;                             // indirect_occ = sum w * 1
    v[V_MEAN] = wc * mx; v[V_MEAN + 1] = wc * my; v[V_MEAN + 2] = wc * mz;
    v[V_RD] = wc * ray_dist();
    v[V_LD] = a.lights ? wc * light_dist() : 0.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float np_v = 0.0f, ng_v = 0.0f;
      if (a.normals_pred) { const float l = a.normals_pred[c * np + pidx]; np_v = act ? l : 0.0f; }
      if (a.normals_grad) { const float l = a.normals_grad[c * np + pidx]; ng_v = act ? l : 0.0f; }
      v[V_NP + c] = a.normals_pred ? wc * np_v : 0.0f;
      v[V_NG + c] = a.normals_grad ? wc * ng_v : 0.0f;
    }
    wave_sum_n<V_COUNT>(v);
    store3(RC_OUT_RGB, v[V_RGB] + bgw, v[V_RGB + 1] + bgw, v[V_RGB + 2] + bgw);
    store3(RC_OUT_DIRECT_RGB, v[V_AD], v[V_AD + 1], v[V_AD + 2]);
    store3(RC_OUT_INDIRECT_DIFFUSE_RGB, v[V_IDF], v[V_IDF + 1], v[V_IDF + 2]);
    store3(RC_OUT_INDIRECT_SPECULAR_RGB, v[V_IS], v[V_IS + 1], v[V_IS + 2]);
    store3(RC_OUT_SPECULAR_RGB, v[V_IS], v[V_IS + 1], v[V_IS + 2]);             // ambient_specular == 0 exactly
    store3(RC_OUT_ALBEDO_RGB, v[V_TINT], v[V_TINT + 1], v[V_TINT + 2]);
    store3(RC_OUT_DIFFUSE_RGB, v[V_DIF], v[V_DIF + 1], v[V_DIF + 2]);
    store3(RC_OUT_INDIRECT_RGB, v[V_IND], v[V_IND + 1], v[V_IND + 2]);
    store3(RC_OUT_INDIRECT_OCC, v[V_OCC], v[V_OCC], v[V_OCC]);
    store3(RC_OUT_MEANS, v[V_MEAN], v[V_MEAN + 1], v[V_MEAN + 2]);
    store1(RC_OUT_RAY_DISTS, v[V_RD]);
    if (a.lights) store1(RC_OUT_LIGHT_DISTS, v[V_LD]);
    if (a.normals_pred) store3(RC_OUT_NORMALS_PRED, v[V_NP], v[V_NP + 1], v[V_NP + 2]);
    if (a.normals_grad) store3(RC_OUT_NORMALS, v[V_NG], v[V_NG + 1], v[V_NG + 2]);
  } else {
    if (want(RC_OUT_RGB)) {
      float c3[3] = {w * shade(RC_SH_RGB), w * shade(RC_SH_RGB + 1), w * shade(RC_SH_RGB + 2)};    // three loads in flight
      wave_sum_n<3>(c3);
      store3(RC_OUT_RGB, c3[0] + bgw, c3[1] + bgw, c3[2] + bgw);
    }
    if (want(RC_OUT_INDIRECT_OCC)) {
      const float wsum = wave_sum(wc);
      store3(RC_OUT_INDIRECT_OCC, wsum, wsum, wsum);
    }
    if (want(RC_OUT_MEANS)) store3(RC_OUT_MEANS, wave_sum(wc * mx), wave_sum(wc * my), wave_sum(wc * mz));
    if (want(RC_OUT_RAY_DISTS)) store1(RC_OUT_RAY_DISTS, wave_sum(wc * ray_dist()));
    if (a.lights && want(RC_OUT_LIGHT_DISTS)) store1(RC_OUT_LIGHT_DISTS, wave_sum(wc * light_dist()));
    if (a.normals_pred && want(RC_OUT_NORMALS_PRED)) {
      const float nx = act ? a.normals_pred[pidx] : 0.0f, ny = act ? a.normals_pred[np + pidx] : 0.0f,
                  nz = act ? a.normals_pred[2 * np + pidx] : 0.0f;
      store3(RC_OUT_NORMALS_PRED, wave_sum(wc * nx), wave_sum(wc * ny), wave_sum(wc * nz));
    }
    if (a.normals_grad && want(RC_OUT_NORMALS)) {
      const float nx = act ? a.normals_grad[pidx] : 0.0f, ny = act ? a.normals_grad[np + pidx] : 0.0f,
                  nz = act ? a.normals_grad[2 * np + pidx] : 0.0f;
      store3(RC_OUT_NORMALS, wave_sum(wc * nx), wave_sum(wc * ny), wave_sum(wc * nz));
    }
  }
  store1(RC_OUT_ACC, acc);

  // distances (render.py:227-245) always use weights_no_filter
  if (want(RC_OUT_DISTANCE_MEAN) || want(RC_OUT_DISTANCE_PERCENTILE_5) || want(RC_OUT_DISTANCE_MEDIAN) ||
      want(RC_OUT_DISTANCE_PERCENTILE_95)) {
    const float tmid = 0.5f * (t0 + t1);
    const float e = wave_sum(act ? wnf * logf(tmid) : 0.0f) / fmaxf(RC_EPS, acc);
    float dm = expf(e);
    // render.py:233-237 `jnp.nan_to_num(x, jnp.inf)`: the second positional parameter of jax 0.4.16's nan_to_num is
    // `copy`, so nan keeps its default 0.0 (then the clip lifts it to tdist[0]); +inf -> finfo.max (oracle/JAX_CALLS.md)
    if (dm != dm) dm = 0.0f;
    dm = fminf(dm, RC_FMAX);
    dm = fminf(fmaxf(dm, s_t[0]), s_t[S]);
    store1(RC_OUT_DISTANCE_MEAN, dm);
    const float wn = wnf / fmaxf(RC_EPS, acc);
    const float incl = wave_scan_incl(wn, lane);
    if (lane == 0) s_cw[0] = 0.0f;
    if (lane < S - 1) s_cw[lane + 1] = fminf(1.0f, incl);
    if (lane == 0) s_cw[S] = 1.0f;
    lds_sync<false>();                           // this wave's own slice
    if (lane < 3 && ray_ok) {
      const float ps = a.pct[lane] / 100.0f;
      const float v = interp1(ps, s_cw, s_t, S + 1);
      const int id = lane == 0 ? RC_OUT_DISTANCE_PERCENTILE_5 : (lane == 1 ? RC_OUT_DISTANCE_MEDIAN : RC_OUT_DISTANCE_PERCENTILE_95);
      if (a.out.ptr[id]) a.out.ptr[id][ray] = v;
    }
  }
}

}  // namespace

// see rc_internal.h.  k_composite in this case: rgb[c] = sum over the lanes of w * shade with w = filt_weight on the
// picked lane and 0 elsewhere (adding +0 is exact, the product is never -0: both factors are >= 0), + max(0, 1 - acc) * bg.
__global__ void k_composite_pick(int64_t n, const float* __restrict__ shade_rgb, const float* __restrict__ filt_weight,
                                 const float* __restrict__ acc_in, float bg, float* __restrict__ out_rgb,
                                 float* __restrict__ out_acc) {
  const int64_t ray = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (ray >= n) return;
  const float acc = acc_in[ray];
  if (out_rgb) {
    const float w = filt_weight[ray];
    const float bgw = fmaxf(0.0f, 1.0f - acc) * bg;
#pragma unroll
    for (int c = 0; c < 3; ++c) out_rgb[3 * ray + c] = w * shade_rgb[(int64_t)c * n + ray] + bgw;
  }
  if (out_acc) out_acc[ray] = acc;
}

__global__ void k_ladder_bounds(float near, float far, float far_clamp, float p, float premult, float* out) {
  far = fminf(far, far_clamp);                                      // models.py:670-673 (sample_level_ray, secondary)
  out[0] = power_ladder(near, p, premult);
  out[1] = power_ladder(far, p, premult);
}
void rc_launch_ladder_bounds(float near, float far, float far_clamp, float p, float premult, float* out, hipStream_t stream) {
  hipLaunchKernelGGL(k_ladder_bounds, dim3(1), dim3(1), 0, stream, near, far, far_clamp, p, premult, out);
}

void rc_launch_sample(const RcSampleArgs& a, hipStream_t stream) {
  if (a.n_rays <= 0) return;
  const USpec us = make_uspec(a.S, a.jitter != nullptr);
  float y_max = 0.0f;
  if (a.raydist_p < 0.0f) y_max = nextafterf((a.raydist_p - 1.0f) / a.raydist_p, -INFINITY);
  dim3 grid((unsigned)((a.n_rays + kWavesPerBlock - 1) / kWavesPerBlock)), block(kWavesPerBlock * 64);
  hipLaunchKernelGGL(k_sample_level, grid, block, 0, stream, a, us, y_max);
}

void rc_launch_sample_intervals(const float* t, const float* logits, int64_t n, int P, int S, const float* jitter,
                                float* out, hipStream_t stream) {
  if (n <= 0) return;
  const USpec us = make_uspec(S, jitter != nullptr);
  dim3 grid((unsigned)((n + kWavesPerBlock - 1) / kWavesPerBlock)), block(kWavesPerBlock * 64);
  hipLaunchKernelGGL(k_sample_intervals, grid, block, 0, stream, t, logits, n, P, S, jitter, out, us);
}

void rc_launch_resample(const RcResampleArgs& a, hipStream_t stream) {
  if (a.n_rays <= 0) return;
  dim3 grid((unsigned)((a.n_rays + kWavesPerBlock - 1) / kWavesPerBlock)), block(kWavesPerBlock * 64);
  hipLaunchKernelGGL(k_resample, grid, block, 0, stream, a);
}

void rc_launch_composite(const RcCompositeArgs& a, hipStream_t stream) {
  if (a.n_rays <= 0) return;
  dim3 grid((unsigned)((a.n_rays + kWavesPerBlock - 1) / kWavesPerBlock)), block(kWavesPerBlock * 64);
  hipLaunchKernelGGL(k_composite, grid, block, 0, stream, a);
}

void rc_launch_composite_pick(int64_t n, const float* shade_rgb, const float* filt_weight, const float* acc, float bg,
                              float* out_rgb, float* out_acc, hipStream_t stream) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_composite_pick, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, n, shade_rgb, filt_weight, acc, bg,
                     out_rgb, out_acc);
}
