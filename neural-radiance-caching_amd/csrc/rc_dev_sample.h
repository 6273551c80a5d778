// Device-side per-ray primitives: wave scans, step-function resampling, alpha weights, interp
// (shared by rc_sample.hip and rc_fused.hip).  See rc_sample.hip for the reference mapping.
#pragma once
#include "rc_internal.h"

namespace rcdev {

// LDS hand-off between lanes: a workgroup barrier when several waves share the data flow
// (stand-alone kernels), a wave-local fence when every wave works on its own LDS slice.
template <bool BLOCK_SYNC>
__device__ __forceinline__ void lds_sync() {
  if constexpr (BLOCK_SYNC) {
    __syncthreads();
  } else {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

__device__ __forceinline__ void lds_sync_wave() { lds_sync<false>(); }

constexpr int kWavesPerBlock = 4;
constexpr int kMaxBins = 64;          // P, S <= 64
constexpr int kSlots = kMaxBins + 1;  // fence posts

// Cross-lane data movement on the VALU (DPP), no LDS round trip.  CTRL: row_shr:n = 0x110 + n,
// row_bcast:15 = 0x142, row_bcast:31 = 0x143.  Lanes without a source (or masked off) receive `old`.
template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf>
__device__ __forceinline__ float dpp_mov(float old, float src) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src),
                                                               CTRL, ROW_MASK, BANK_MASK, false));
}
// Inclusive scan of a wave with an associative op (identity `id`): 3 row shifts of the input, 2 of the
// partial result (row prefix), then the row totals are passed on with the two row broadcasts.
template <typename Op>
__device__ __forceinline__ float wave_scan_op(float x, float id, Op op) {
  float r = op(x, dpp_mov<0x111>(id, x));
  r = op(r, dpp_mov<0x112>(id, x));
  r = op(r, dpp_mov<0x113>(id, x));
  r = op(r, dpp_mov<0x114, 0xf, 0xe>(id, r));
  r = op(r, dpp_mov<0x118, 0xf, 0xc>(id, r));
  r = op(r, dpp_mov<0x142, 0xa, 0xf>(id, r));
  r = op(r, dpp_mov<0x143, 0xc, 0xf>(id, r));
  return r;
}
__device__ __forceinline__ float lane63(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_scan_incl(float v, int /*lane*/) {
  return wave_scan_op(v, 0.0f, [](float a, float b) { return a + b; });
}
// total of the wave in scan order (lane 63 of the inclusive scan), broadcast
__device__ __forceinline__ float wave_sum_scan(float v) { return lane63(wave_scan_incl(v, 0)); }
// Sum of a wave, broadcast to every lane: butterfly inside each row of 16 lanes on DPP moves (quad_perm xor 1, xor 2,
// row_half_mirror, row_mirror), the four row totals passed on with the two row broadcasts, lane 63 read back -- six
// vector adds per value and no LDS crossbar (ds_bpermute) round trips.  wave_sum and wave_sum_n add in the same order.
__device__ __forceinline__ float wave_sum_step(float v, int step) {
  switch (step) {
    case 0: return v + dpp_mov<0xB1>(0.0f, v);                 // quad_perm [1, 0, 3, 2]
    case 1: return v + dpp_mov<0x4E>(0.0f, v);                 // quad_perm [2, 3, 0, 1]
    case 2: return v + dpp_mov<0x141>(0.0f, v);                // row_half_mirror
    case 3: return v + dpp_mov<0x140>(0.0f, v);                // row_mirror
    case 4: return v + dpp_mov<0x142, 0xa, 0xf>(0.0f, v);      // row_bcast:15 into rows 1 and 3
    default: return v + dpp_mov<0x143, 0xc, 0xf>(0.0f, v);     // row_bcast:31 into rows 2 and 3
  }
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int st = 0; st < 6; ++st) v = wave_sum_step(v, st);
  return lane63(v);
}
// N independent wave sums at once: the N reductions advance step by step (independent adds back to back; same add
// order per sum as wave_sum).
template <int N>
__device__ __forceinline__ void wave_sum_n(float (&v)[N]) {
#pragma unroll
  for (int st = 0; st < 6; ++st) {
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] = wave_sum_step(v[k], st);
  }
#pragma unroll
  for (int k = 0; k < N; ++k) v[k] = lane63(v[k]);
}
__device__ __forceinline__ float wave_max(float v) {
  return lane63(wave_scan_op(v, -INFINITY, [](float a, float b) { return fmaxf(a, b); }));
}

__device__ __forceinline__ float safe_log(float x) { return logf(fminf(fmaxf(x, RC_TINY), RC_FMAX)); }

// math.power_ladder for finite p not in {0,1} (math.py:295-316)
__device__ __forceinline__ float power_ladder(float x, float p, float premult) {
  x = x * premult;
  const float xp = fabsf(x);
  const float xs = xp / fmaxf(RC_TINY, fabsf(p - 1.0f));
  float y = fabsf(p - 1.0f) / p * (powf(xs + 1.0f, p) - 1.0f);
  y = fminf(fmaxf(y, -RC_FMAX), RC_FMAX);
  return x < 0.0f ? -y : y;
}
// x^c for a positive normal x: 2^(c log2 x) on the hardware's log2 / exp2 with the exponent of x split off (|log2 m| <=
// 0.5), the product c * e carried with its rounding residual and the integer part of the result's exponent applied by
// ldexp -- about 1.5 ulp in 16 instructions, where the library's powf (<= 1 ulp, every special case) takes ~200: the
// s -> t mapping of the secondary rays' samples (one power per fence post, 65 posts on 64 lanes = two passes per ray and
// level) was 40 % of the vector instructions of the sampler in front of every level of the secondary trace.  Anything
// else (zero, subnormal, negative, inf, NaN) goes to powf.
__device__ __forceinline__ float pow_pos(float x, float c) {
  if (!(x >= 1.17549435e-38f && x < INFINITY)) return powf(x, c);
  int e = __builtin_amdgcn_frexp_expf(x);              // x = m 2^e, m in [0.5, 1)
  float m = __builtin_amdgcn_frexp_mantf(x);
  const bool low = m < 0.70710678f;
  m = low ? m + m : m;
  e = low ? e - 1 : e;                                 // m in [0.707, 1.414)
  const float l = __builtin_amdgcn_logf(m);            // v_log_f32: log2
  const float fe = (float)e;
  const float ce = c * fe;
  const float r = __builtin_fmaf(c, fe, -ce);          // what the rounding of c * e dropped
  const float rest = __builtin_fmaf(c, l, r);
  const float n = rintf(ce);
  const float f = (ce - n) + rest;                     // |f| < 1
  return ldexpf(__builtin_amdgcn_exp2f(f), (int)n);
}

// math.inv_power_ladder (math.py:319-341); y_max = minus_eps((p-1)/p) for p < 0.
__device__ __forceinline__ float inv_power_ladder(float y, float p, float premult, float y_max) {
  float yp = fabsf(y);
  if (p < 0.0f) yp = fminf(fmaxf(yp, -y_max), y_max);
  const float pm1 = fabsf(p - 1.0f);
  float x = pm1 * (pow_pos((p / pm1) * yp + 1.0f, 1.0f / p) - 1.0f);
  x = y < 0.0f ? -x : x;
  return x / premult;
}

// Alpha-compositing weights of one ray; lane i < S owns interval i.  Returns w, and (by ref) dd.
__device__ __forceinline__ float alpha_weight(float density, float t0, float t1, float dnorm, bool active, int lane) {
  // render.py:143-168: delta = (t1-t0)*||d||; dd = density*|delta|; alpha = 1-exp(-dd); T = exp(-excl cumsum)
  const float dd = active ? density * fabsf((t1 - t0) * dnorm) : 0.0f;
  const float incl = wave_scan_incl(dd, lane);
  float excl = dpp_mov<0x138>(0.0f, incl);      // wave_shr:1, lane 0 keeps 0
  const float alpha = 1.0f - expf(-dd);
  const float trans = expf(-excl);
  return active ? alpha * trans : 0.0f;
}

// Number of entries of s[0..m-1] (sorted ascending) that are <= x  (searchsorted side='right').
__device__ __forceinline__ int upper_bound(const float* s, int m, float x) {
  int lo = 0, hi = m;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (s[mid] <= x) lo = mid + 1; else hi = mid;
  }
  return lo;
}

struct USpec { float start, stop, max_jitter; };

// stepfun.sample_intervals on one wave.  s_t/s_cw: LDS [P+1]; writes sorted samples to s_out[0..S].
// w_logits: lane p < P holds the logit of bin p.
template <bool BLOCK_SYNC = true>
__device__ __forceinline__ void sample_intervals_wave(float logit, int P, int S, USpec us, bool has_jitter,
                                                      float jitter, const float* s_t, float* s_cw, float* s_c,
                                                      float* s_v, float* s_out, int lane) {
  // softmax (jax.nn.softmax) + integrate_weights (stepfun.py:125-144).  One bin (the first level of every sampler call,
  // wave-uniform): softmax of one logit is exp(0) / exp(0) = 1 and the CDF is [0, 1] whatever the logit -- no scans.
  if (P > 1) {
    const bool binact = lane < P;
    const float m = wave_max(binact ? logit : -INFINITY);
    const float e = binact ? expf(logit - m) : 0.0f;
    const float ssum = wave_sum_scan(e);
    const float wn = e / ssum;
    const float incl = wave_scan_incl(wn, lane);
    if (lane == 0) s_cw[0] = 0.0f;
    if (lane < P - 1) s_cw[lane + 1] = fminf(1.0f, incl);
    if (lane == 0) s_cw[P] = 1.0f;
    lds_sync<BLOCK_SYNC>();
  }
  // u (stepfun.py:186-202); linspace = start*(1-step) + stop*step, last == stop
  if (lane < S) {
    float u;
    if (lane == S - 1) {
      u = us.stop;
    } else {
      const float step = (float)lane / (float)(S - 1);
      u = us.start * (1.0f - step) + us.stop * step;
    }
    if (has_jitter) u = u + jitter * us.max_jitter;
    // sorted_interp (math.py:448-457)
    if (P > 1) {
      const int idx = upper_bound(s_cw, P + 1, u);
      const int i1 = min(idx, P), i0 = max(idx - 1, 0);
      const float c0 = s_cw[i0], c1 = s_cw[i1], t0 = s_t[i0], t1 = s_t[i1];
      const float off = fminf(fmaxf((u - c0) / fmaxf(RC_EPS * RC_EPS, c1 - c0), 0.0f), 1.0f);
      s_c[lane] = t0 + off * (t1 - t0);
    } else {
      // the same statements on cw = [0, 1]: searchsorted(side='right') gives 1 for u < 1 (interval [0, 1]) and 2 for
      // u >= 1 (clamped to i0 = i1 = 1, offset clip(0 / eps^2) = 0) -- same operations on the same values, same bits
      const bool in = u < 1.0f;
      const float c0 = in ? 0.0f : 1.0f, t0 = in ? s_t[0] : s_t[1], t1 = s_t[1];
      const float off = fminf(fmaxf((u - c0) / fmaxf(RC_EPS * RC_EPS, 1.0f - c0), 0.0f), 1.0f);
      s_c[lane] = t0 + off * (t1 - t0);
    }
  }
  lds_sync<BLOCK_SYNC>();
  // midpoints + reflected end posts, clip to [0,1] (stepfun.py:239-248)
  for (int e2 = lane; e2 <= S; e2 += 64) {
    float v;
    if (e2 == 0) {
      const float mid0 = (s_c[1] + s_c[0]) / 2.0f;
      v = 2.0f * s_c[0] - mid0;
    } else if (e2 == S) {
      const float midl = (s_c[S - 1] + s_c[S - 2]) / 2.0f;
      v = 2.0f * s_c[S - 1] - midl;
    } else {
      v = (s_c[e2] + s_c[e2 - 1]) / 2.0f;
    }
    s_v[e2] = fminf(fmaxf(v, 0.0f), 1.0f);
  }
  lds_sync<BLOCK_SYNC>();
  // jnp.sort: exact rank sort (ties broken by index).  The input is sorted already unless a rounding
  // step at a bin boundary produced an inversion: a wave that owns its arrays checks and copies.
  bool need_sort = true;
  if constexpr (!BLOCK_SYNC) {
    const bool inv = lane < S && s_v[lane] > s_v[lane + 1];
    need_sort = __ballot(inv) != 0ull;
  }
  if (need_sort) {
    for (int e2 = lane; e2 <= S; e2 += 64) {
      const float v = s_v[e2];
      int rank = 0;
      for (int k = 0; k <= S; ++k) {
        const float o = s_v[k];
        rank += (o < v) || (o == v && k < e2);
      }
      s_out[rank] = v;
    }
  } else {
    for (int e2 = lane; e2 <= S; e2 += 64) s_out[e2] = s_v[e2];
  }
  lds_sync<BLOCK_SYNC>();
}

// One level of the proposal sampler for one ray on one wave (the body of k_sample_level; also the first half of
// rc_level.hip's per-ray level kernel): weights of the previous level -> resampling logits -> sample_intervals ->
// s -> t -> cast.  `lds`: this wave's 5 x (kSlots + 3) floats.  Stores sdist / tdist / means (and the previous level's
// weights) when ray_ok; returns the sample mean of interval `lane` (lanes < S).
// Replaces sampling.py:284-639 per level, coord.py:223-260, render.py:49-59, 106-131, sampling.py:182-205.
__device__ __forceinline__ void sample_level_ray(const RcSampleArgs& a, const USpec& us, float y_max, int64_t ray, bool ray_ok,
                                                 float* lds, int lane, float& mean_x, float& mean_y, float& mean_z) {
  float* s_t = lds;
  float* s_cw = lds + (kSlots + 3);
  float* s_c = lds + 2 * (kSlots + 3);
  float* s_v = lds + 3 * (kSlots + 3);
  float* s_out = lds + 4 * (kSlots + 3);

  const float ox = a.origins[3 * ray], oy = a.origins[3 * ray + 1], oz = a.origins[3 * ray + 2];
  const float dx = a.directions[3 * ray], dy = a.directions[3 * ray + 1], dz = a.directions[3 * ray + 2];
  float near = a.near[ray], far = a.far[ray];
  if (a.secondary) {
    far = fminf(far, a.far_clamp);                                  // models.py:670-673
    if (a.normals) {                                                // sampling.py:182-205
      const float dp = a.viewdirs[3 * ray] * a.normals[3 * ray] + a.viewdirs[3 * ray + 1] * a.normals[3 * ray + 1] +
                       a.viewdirs[3 * ray + 2] * a.normals[3 * ray + 2];
      float off = fminf(fmaxf(a.eps_dot_min / fmaxf(dp, 1e-5f), near), far);
      off = dp > 0.0f ? off : near;
      near = fmaxf(near, off);
      near = fminf(fmaxf(near, 1e-5f), far - 1e-5f);
    }
  }
  const int P = a.P, S = a.S;

  // --- weights of the previous level -> resampling logits (sampling.py:339)
  float w;
  if (a.prev_sdist == nullptr) {
    w = 1.0f;
    if (lane == 0) { s_t[0] = 0.0f; s_t[1] = 1.0f; }
  } else {
    const bool act = lane < P;
    const float t0 = act ? a.prev_tdist[ray * (P + 1) + lane] : 0.0f;
    const float t1 = act ? a.prev_tdist[ray * (P + 1) + lane + 1] : 0.0f;
    const float dens = act ? a.prev_density[ray * P + lane] : 0.0f;
    const float dnorm = sqrtf(dx * dx + dy * dy + dz * dz);
    w = alpha_weight(dens, t0, t1, dnorm, act, lane);
    if (act && ray_ok && a.prev_weights) a.prev_weights[ray * P + lane] = w;
    {
      // P + 1 <= 65 fence posts on 64 lanes: both loads of a lane issued before the LDS stores (as a loop the second pass,
      // one lane, was a dependent round trip of its own)
      const bool h0 = lane <= P, h1 = lane + 64 <= P;
      float v0 = 0.0f, v1 = 0.0f;
      if (h0) v0 = a.prev_sdist[ray * (P + 1) + lane];
      if (h1) v1 = a.prev_sdist[ray * (P + 1) + lane + 64];
      if (h0) s_t[lane] = v0;
      if (h1) s_t[lane + 64] = v1;
    }
  }
  const float logit = a.anneal * safe_log(w + a.padding);
  const bool hasj = a.jitter != nullptr;
  const float jit = hasj ? a.jitter[ray] : 0.0f;
  // every wave works on its own LDS slice: wave-local hand-offs, and the exact rank sort of jnp.sort only when the
  // clipped fence posts really contain an inversion (same values as the unconditional sort)
  lds_sync<false>();
  sample_intervals_wave<false>(logit, P, S, us, hasj, jit, s_t, s_cw, s_c, s_v, s_out, lane);

  // --- s -> t (coord.py:259-260), cast (render.py:49-59, 106-131)
  float s_near = 0.0f, s_far = 0.0f;
  if (a.use_raydist) {
    if (a.s_bounds) {           // one (near, far) for the whole batch: computed once by the same function
      s_near = a.s_bounds[0];
      s_far = a.s_bounds[1];
    } else {
      s_near = power_ladder(near, a.raydist_p, a.raydist_premult);
      s_far = power_ladder(far, a.raydist_p, a.raydist_premult);
    }
  }
  for (int e2 = lane; e2 <= S; e2 += 64) {
    const float s = s_out[e2];
    float t;
    if (a.use_raydist) t = inv_power_ladder(s * s_far + (1.0f - s) * s_near, a.raydist_p, a.raydist_premult, y_max);
    else t = s * far + (1.0f - s) * near;
    s_v[e2] = t;
    if (ray_ok) {
      a.sdist[ray * (S + 1) + e2] = s;
      a.tdist[ray * (S + 1) + e2] = t;
    }
  }
  lds_sync<false>();
  mean_x = 0.0f; mean_y = 0.0f; mean_z = 0.0f;
  if (lane < S) {
    const float t0 = s_v[lane], t1 = s_v[lane + 1];
    const float sm = t0 + t1, d = t1 - t0;
    const float ratio = (d * d) / fmaxf(RC_EPS * RC_EPS, 3.0f * (sm * sm) + d * d);
    const float tm = sm * (0.5f + ratio);
    mean_x = dx * tm + ox; mean_y = dy * tm + oy; mean_z = dz * tm + oz;
    if (ray_ok && a.means) {
      const int64_t np = a.n_rays * S, pidx = ray * S + lane;
      a.means[pidx] = mean_x;
      a.means[np + pidx] = mean_y;
      a.means[2 * np + pidx] = mean_z;
    }
  }
}

// jnp.interp(x, xp, fp) on LDS arrays of length m (stepfun.weighted_percentile).
__device__ __forceinline__ float interp1(float x, const float* xp, const float* fp, int m) {
  int i = upper_bound(xp, m, x);
  i = min(max(i, 1), m - 1);
  const float df = fp[i] - fp[i - 1];
  const float dxx = xp[i] - xp[i - 1];
  const float delta = x - xp[i - 1];
  const float epsilon = 1.4210855e-14f;   // np.spacing(np.finfo(np.float32).eps)
  const bool dx0 = fabsf(dxx) <= epsilon;
  float f = dx0 ? fp[i - 1] : fp[i - 1] + (delta / (dx0 ? 1.0f : dxx)) * df;
  if (x < xp[0]) f = fp[0];
  if (x > xp[m - 1]) f = fp[m - 1];
  return f;
}

inline USpec make_uspec(int S, bool has_jitter) {
  // stepfun.py:186-202: Python-float (double) arithmetic rounded to float32 on use.
  const double eps = (double)RC_EPS;
  USpec u;
  if (!has_jitter) {
    const double pad = 1.0 / (2.0 * S);
    u.start = (float)pad;
    u.stop = (float)(1.0 - pad - eps);
    u.max_jitter = 0.0f;
  } else {
    const double u_max = eps + (1.0 - eps) / S;
    u.start = 0.0f;
    u.stop = (float)(1.0 - u_max);
    u.max_jitter = (float)((1.0 - u_max) / (S - 1) - eps);
  }
  return u;
}


}  // namespace rcdev
