// Fused cache forward for primary rays: ONE launch per ray batch, one wavefront per ray.
//
//   3 x [ resample intervals -> cast -> contract -> grid lookup -> density MLP -> alpha weights ]
//   -> appearance grid -> cache shader -> volume compositing
//
// Everything between the ray record and the rendered pixel stays on chip: the step functions and
// sample distances live in a per-wave LDS slice, grid features are written straight into the MFMA
// B-operand slots of the density MLP / shader, the 64-wide hidden feature of the last density MLP is
// already parked where the shader reads it, per-sample colours are composited out of registers.
// The four waves of a workgroup (four rays) march in lockstep through ONE weight stream
// [density MLP 0 | density MLP 1 | density MLP 2 (+ backward fragments) | shader] pulled through the
// LDS ring by LDS-DMA.  Levels 0/1 have 64 samples per ray = two 32-point MFMA tiles per wave, so every
// weight fragment feeds two MFMAs there.
//
// Same arithmetic, in the same order, as the stand-alone kernels (rc_sample.hip, rc_hashgrid.hip,
// rc_mlp.hip), whose reference citations apply; this file only changes where the data lives.
// Used for the cache pass on primary rays without resampling (BASELINE configs 1, 2, 4); secondary
// rays, resampling and the material stage use the stand-alone kernels.
#include "rc_fused_common.h"
#include <type_traits>

using namespace rcdev;
using namespace rcfused;

namespace {

// Density MLP of a proposal level on 64 samples (two point-tiles); returns the raw density of
// sample `lane`.  K grid features of this lane's sample are in f[].
template <int K, int FB, int NF, class WS>
__device__ __forceinline__ float density_level64(const WS& ws, float* act_wave, int lane, const float (&f)[K]) {
  constexpr int KS0 = (K + 1) / 2 + 1;
  using FR = DensFrags<KS0>;
  const int tile = lane >> 5, j = lane & 31, h = lane >> 5;
  // scatter the features into B-operand slots: feature k of point (tile, j) -> step k/2, lane j + 32 (k & 1)
#pragma unroll
  for (int k = 0; k < 2 * (KS0 - 1); ++k)
    act_wave[tile * kTileStride + (k >> 1) * 64 + j + 32 * (k & 1)] = k < K ? f[k < K ? k : 0] : 0.0f;
#pragma unroll
  for (int p = 0; p < 2; ++p) act_wave[p * kTileStride + (KS0 - 1) * 64 + lane] = h == 0 ? 1.0f : 0.0f;
  lds_sync<false>();
  float* act = act_wave + lane;
  f32x16 acc[2][2];
#pragma unroll
  for (int p = 0; p < 2; ++p) { acc[p][0] = zero16(); acc[p][1] = zero16(); }
  mlp_layer_pt<2, 2, KS0, FB + FR::D0, NF>(ws, act, kTileStride, acc);
#pragma unroll
  for (int p = 0; p < 2; ++p) { park<2, true>(acc[p], act + p * kTileStride, 0); act[p * kTileStride + 32 * 64] = h == 0 ? 1.0f : 0.0f; }
#pragma unroll
  for (int p = 0; p < 2; ++p) { acc[p][0] = zero16(); acc[p][1] = zero16(); }
  mlp_layer_pt<2, 2, 33, FB + FR::D1, NF>(ws, act, kTileStride, acc);
  float out[2][1], nokeep[1];
  dot_out<2, 2, 1, FB + FR::DO, NF>(ws, acc, out, nokeep);       // output_density_layer on relu(acc), both point-tiles
  // the dot product is complete on both half-waves: sample `lane` = (tile = lane >> 5, point lane & 31)
  return tile == 0 ? out[0][0] : out[1][0];
}

// FRONT: stop behind the last proposal level (density MLP + appearance lookup) and hand the per-sample results to the
// stages of the time-resolved cache; the stream then ends at F_SH.
// DIRECT (experiment, -DRC_FUSED_DIRECT_EXPERIMENT + RC_FUSED_DIRECT=1): the weight fragments come straight from global
// memory (WDirect: L2-resident, one coalesced 256-byte load per fragment) instead of through the workgroup's LDS ring:
// no ring, no workgroup barrier anywhere in the kernel, so the four rays of a workgroup drift apart and their gather
// phases stop queueing behind each other in the CU's memory pipe.  Bitwise equal results; measured 145 us per 1024 rays
// against 132 us through the ring: 2165 extra vector loads per ray and their L2 latency cost more than the 40 barriers
// and the lockstep.  Kept as a template parameter, not instantiated in the product build.
// EXPORT: the full kernel that also leaves the last level's per-sample results (fence posts, density, sample means,
// predicted normals) in the workspace buffers of the launch-per-stage plan, for a caller that goes on working per sample
// behind the cache pass (the material stage: its shading-point pick and the material-only composite).
template <bool GRAD, bool FRONT = false, bool DIRECT = false, bool EXPORT = false>
__global__ __launch_bounds__(kWaves * 64) void k_cache_fused(RcFusedArgs a) {
  split_exclusive_simd();
  constexpr int NF = FRONT ? F_SH : NF_FUSED;
  extern __shared__ __attribute__((aligned(16))) float lds_dyn[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int64_t ray = (int64_t)blockIdx.x * kWaves + wave;
  const bool ray_ok = ray < a.n;
  if (!ray_ok) ray = a.n - 1;                      // keep the wave in the workgroup's lockstep
  float* ring = lds_dyn;
  float* act_wave = lds_dyn + kRingFloats + wave * (kShActSteps * 64);
  float* scr = lds_dyn + kRingFloats + kWaves * (kShActSteps * 64) + wave * kScratch;
  float* s_sd[2] = {scr, scr + 68};
  float* s_td = scr + 2 * 68; float* s_cw = scr + 3 * 68; float* s_c = scr + 4 * 68; float* s_v = scr + 5 * 68;
  float* s_out = scr + 6 * 68;
  typename std::conditional<DIRECT, WDirect, WStream>::type ws;
  if constexpr (DIRECT) { ws.g = a.wstream; ws.lane = lane; }
  else { ws.g = a.wstream; ws.ring = ring; ws.lane = lane; ws.wave = wave; }
#ifdef RC_STAMPS
  unsigned long long stamps[16];
  const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
#endif
  RC_FSTAMP(0);
  if constexpr (!DIRECT) ws_begin<NF>(ws);

  const float ox = a.origins[3 * ray], oy = a.origins[3 * ray + 1], oz = a.origins[3 * ray + 2];
  const float dx = a.directions[3 * ray], dy = a.directions[3 * ray + 1], dz = a.directions[3 * ray + 2];
  const float near = a.near[ray], far = a.far[ray];
  const float dnorm = sqrtf(dx * dx + dy * dy + dz * dz);

  // resample S intervals from (prev sdist in s_prev [P+1], logit per bin), produce tdist in s_td and
  // the sample mean of interval `idx` (all lanes)
  auto resample = [&](int level, int P, int S, float logit, const float* s_prev) {
    const bool hasj = a.jitter[level] != nullptr;
    const float jit = hasj ? a.jitter[level][ray] : 0.0f;
#if defined(RC_ABL) && RC_ABL == 14
    for (int e2 = lane; e2 <= S; e2 += 64) s_out[e2] = (float)e2 / (float)S + 1e-9f * logit;
    lds_sync<false>();
#else
    sample_intervals_wave<false>(logit, P, S, a.us[level], hasj, jit, s_prev, s_cw, s_c, s_v, s_out, lane);
#endif
    if constexpr (FRONT) {
      if (a.use_raydist) {      // TransientNeRFModel samples its primary rays in power-ladder distance (models.py:122, 183-191)
        const float s_near = power_ladder(near, a.raydist_p, a.raydist_premult), s_far = power_ladder(far, a.raydist_p, a.raydist_premult);
        for (int e2 = lane; e2 <= S; e2 += 64)
          s_td[e2] = inv_power_ladder(s_out[e2] * s_far + (1.0f - s_out[e2]) * s_near, a.raydist_p, a.raydist_premult, a.y_max);
        lds_sync<false>();
        return;
      }
    }
    for (int e2 = lane; e2 <= S; e2 += 64) s_td[e2] = s_out[e2] * far + (1.0f - s_out[e2]) * near;   // coord.py:259-260
    lds_sync<false>();
  };
  auto mean_of = [&](int idx, float& mx, float& my, float& mz, float& t0, float& t1) {
    t0 = s_td[idx]; t1 = s_td[idx + 1];
    const float sm = t0 + t1, d = t1 - t0;
    const float ratio = (d * d) / fmaxf(RC_EPS * RC_EPS, 3.0f * (sm * sm) + d * d);
    const float tm = sm * (0.5f + ratio);
    mx = dx * tm + ox; my = dy * tm + oy; mz = dz * tm + oz;
  };
  auto density_of = [&](float raw, float cx, float cy, float cz, float bbox) {
    const bool inside = (cx > -bbox) & (cx < bbox) & (cy > -bbox) & (cy < bbox) & (cz > -bbox) & (cz < bbox);
    const float d = rc_safe_exp(raw + a.density_bias);
    return inside ? d : 0.0f;
  };

  // ------------------------------------------------------------------ level 0 (P = 1, S = 64)
  if (lane == 0) { s_sd[0][0] = 0.0f; s_sd[0][1] = 1.0f; }
  lds_sync<false>();
  resample(0, 1, 64, a.anneal * safe_log(1.0f + a.padding), s_sd[0]);
  RC_FSTAMP(1);
  float w;
  {
    float mx, my, mz, t0, t1;
    mean_of(lane, mx, my, mz, t0, t1);
    float cx = mx, cy = my, cz = mz;
    contract3(cx, cy, cz, a.contract_radius);
    float f[6];
    {
      const float ux = unit_box(a.grid[0].bbox, cx), uy = unit_box(a.grid[0].bbox, cy), uz = unit_box(a.grid[0].bbox, cz);
#if defined(RC_ABL) && RC_ABL == 11
      for (int l = 0; l < 6; ++l) f[l] = ux * (float)l + uy - uz;
#else
      Corners<1> C[6];          // all 48 corner loads in flight before the first combine
      RC_FSTAMP(12);
#pragma unroll
      for (int l = 0; l < 6; ++l) {
        const RcGridLevel& L = a.grid[0].lvl[l];
        if (L.dense) grid_fetch_cell(a.cell_table[0][l], L.size, ux, uy, uz, C[l]);
        else grid_fetch<1, true>(L.table, L.size, L.mask, 0u, false, ux, uy, uz, C[l]);
      }
      RC_FSTAMP_NOWAIT(13);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int l = 0; l < 6; ++l) {
        float v[1], jd[1];
        grid_combine<1, false>(C[l], v, jd);
        f[l] = v[0] * a.grid[0].precondition;
      }
#endif
    }
    RC_FSTAMP(2);
#if defined(RC_ABL) && RC_ABL == 13
    const float raw = f[0] + f[5];
#else
    const float raw = density_level64<6, F_L0, NF>(ws, act_wave, lane, f);
#endif
    w = alpha_weight(density_of(raw, cx, cy, cz, a.grid[0].bbox), t0, t1, dnorm, true, lane);
  }
  RC_FSTAMP(3);
  for (int e2 = lane; e2 <= 64; e2 += 64) s_sd[1][e2] = s_out[e2];
  lds_sync<false>();
  // ------------------------------------------------------------------ level 1 (P = 64, S = 64)
  resample(1, 64, 64, a.anneal * safe_log(w + a.padding), s_sd[1]);
  RC_FSTAMP(4);
  {
    float mx, my, mz, t0, t1;
    mean_of(lane, mx, my, mz, t0, t1);
    float cx = mx, cy = my, cz = mz;
    contract3(cx, cy, cz, a.contract_radius);
    float f[7];
    {
      const float ux = unit_box(a.grid[1].bbox, cx), uy = unit_box(a.grid[1].bbox, cy), uz = unit_box(a.grid[1].bbox, cz);
#if defined(RC_ABL) && RC_ABL == 11
      for (int l = 0; l < 7; ++l) f[l] = ux * (float)l + uy - uz;
#else
      Corners<1> C[7];          // all 56 corner loads in flight before the first combine
#pragma unroll
      for (int l = 0; l < 7; ++l) {
        const RcGridLevel& L = a.grid[1].lvl[l];
        if (L.dense) grid_fetch_cell(a.cell_table[1][l], L.size, ux, uy, uz, C[l]);
        else grid_fetch<1, true>(L.table, L.size, L.mask, 0u, false, ux, uy, uz, C[l]);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int l = 0; l < 7; ++l) {
        float v[1], jd[1];
        grid_combine<1, false>(C[l], v, jd);
        f[l] = v[0] * a.grid[1].precondition;
      }
#endif
    }
    RC_FSTAMP(5);
#if defined(RC_ABL) && RC_ABL == 13
    const float raw = f[0] + f[6];
#else
    const float raw = density_level64<7, F_L1, NF>(ws, act_wave, lane, f);
#endif
    w = alpha_weight(density_of(raw, cx, cy, cz, a.grid[1].bbox), t0, t1, dnorm, true, lane);
  }
  RC_FSTAMP(6);
  for (int e2 = lane; e2 <= 64; e2 += 64) s_sd[0][e2] = s_out[e2];
  lds_sync<false>();
  // ------------------------------------------------------------------ level 2 (P = 64, S = 32)
  resample(2, 64, 32, a.anneal * safe_log(w + a.padding), s_sd[0]);
  RC_FSTAMP(7);
  const int j = lane & 31, h = lane >> 5;
  float mx, my, mz, t0, t1;
  mean_of(j, mx, my, mz, t0, t1);
  float cx = mx, cy = my, cz = mz;
  const float zx = rc_div(mx, a.contract_radius), zy = rc_div(my, a.contract_radius), zz = rc_div(mz, a.contract_radius);
  contract3(cx, cy, cz, a.contract_radius);
  float* act = act_wave + lane;
  // (GRAD) d feature / d contracted coordinate of the density grid, 96 values per point, goes to LDS: element e of
  // point j at step kJac + e / 2, lane j + 32 (e & 1); written and read back by the lane (j, 0) that owns the point
  auto jac_at = [&](int e) -> float& { return act_wave[(kJac + (e >> 1)) * 64 + j + 32 * (e & 1)]; };
  {
    // half-wave 0 looks up the level-2 density grid, half-wave 1 the appearance grid (same level sizes)
    const RcGridDev& g = a.grid[2];      // bbox / precondition: same for both grids (checked on the host)
    float f[32];
    const float ux = unit_box(g.bbox, cx), uy = unit_box(g.bbox, cy), uz = unit_box(g.bbox, cz);
    // two rounds of four levels: 32 corner loads of 16 bytes in flight per lane
#if defined(RC_ABL) && RC_ABL == 11
    for (int k = 0; k < 32; ++k) f[k] = ux * (float)k + uy - uz;
    if (false)
#endif
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      Corners<4> C[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int l = half * 4 + q;
        // level geometry is identical for the two grids (checked on the host) and so is the entry index: the host keeps
        // a copy of the two tables interleaved entry by entry ([density 16 B | appearance 16 B]), so the two half-waves
        // of a point read the two halves of ONE 32-byte pair -- half the cache-line requests and sector traffic
        const RcGridLevel& L = a.grid[2].lvl[l];
        grid_fetch<4, true, 2, true>(a.pair_table[l] + 4 * h, L.size, L.mask, 0u, L.dense != 0, ux, uy, uz, C[q]);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int l = half * 4 + q;
        const int size = a.grid[2].lvl[l].size;
        const bool dense = a.grid[2].lvl[l].dense != 0;
        float v[4], jd[GRAD ? 12 : 1];
        grid_combine<4, GRAD>(C[q], v, jd);
#pragma unroll
        for (int c = 0; c < 4; ++c) f[4 * l + c] = v[c] * g.precondition;
        if constexpr (GRAD) {
          const float s = g.precondition * (float)size / (2.0f * g.bbox);
          if (h == 0) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              jac_at(0 * 32 + 4 * l + c) = (dense ? jd[2 * 4 + c] : jd[0 * 4 + c]) * s;
              jac_at(1 * 32 + 4 * l + c) = jd[1 * 4 + c] * s;
              jac_at(2 * 32 + 4 * l + c) = (dense ? jd[0 * 4 + c] : jd[2 * 4 + c]) * s;
            }
          }
        }
      }
    }
    // feature k of point j -> step base + k/2, lane j + 32 (k & 1): density features feed D0 at [0,16),
    // appearance features wait at [kAppTmp, kAppTmp + 16) until the density MLP is through
    const int base = h == 0 ? 0 : kAppTmp;
#pragma unroll
    for (int k = 0; k < 32; ++k) act_wave[(base + (k >> 1)) * 64 + j + 32 * (k & 1)] = f[k];
    act[16 * 64] = h == 0 ? 1.0f : 0.0f;
    lds_sync<false>();
  }
  RC_FSTAMP(8);
  // density MLP of the last level on one tile (same code as k_density_mlp<17, GRAD>)
  using FR = DensFrags<17>;
  float density, npx, npy, npz, ngx = 0.0f, ngy = 0.0f, ngz = 0.0f;
  {
    f32x16 acc[2];
    acc[0] = zero16(); acc[1] = zero16();
    mlp_layer_d<2, 17, F_L2 + FR::D0, NF>(ws, act, acc);
    uint32_t m0 = 0, m1 = 0;
    if constexpr (GRAD) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) m0 |= (acc[t][r] > 0.0f ? 1u : 0u) << (t * 16 + r);
    }
    park<2, true>(acc, act, 0);
    act[32 * 64] = h == 0 ? 1.0f : 0.0f;
    acc[0] = zero16(); acc[1] = zero16();
    mlp_layer_d<2, 33, F_L2 + FR::D1, NF>(ws, act, acc);
    if constexpr (GRAD) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) m1 |= (acc[t][r] > 0.0f ? 1u : 0u) << (t * 16 + r);
    }
    park<2, true>(acc, act, 0);               // hidden feature: stays at [0,32) for the shader
    float out[4], wout[GRAD ? 32 : 1];
    dot_out1<2, 4, F_L2 + FR::DO, NF, GRAD>(ws, acc, out, wout);      // density + predicted normals on relu(acc)
    density = density_of(out[0], cx, cy, cz, a.grid[2].bbox);
    npx = out[1]; npy = out[2]; npz = out[3];
    neg_normalize(npx, npy, npz);
    if constexpr (GRAD) {
      // the backward pass borrows [0, 32): the hidden feature waits in registers meanwhile
      float hid[32];
#pragma unroll
      for (int s = 0; s < 32; ++s) hid[s] = act[s * 64];
      float* bw = act;
#pragma unroll
      for (int s = 0; s < 32; ++s) bw[s * 64] = ((m1 >> s) & 1u) ? wout[s] : 0.0f;
      f32x16 g[2];
      g[0] = zero16(); g[1] = zero16();
      mlp_layer_d<2, 32, F_L2 + FR::B1, NF>(ws, bw, g);
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) bw[(t * 16 + r) * 64] = ((m0 >> (t * 16 + r)) & 1u) ? g[t][r] : 0.0f;
      f32x16 gf[1];
      gf[0] = zero16();
      mlp_layer_d<1, 32, F_L2 + FR::B0, NF>(ws, bw, gf);
      // d raw / d feature i = acc_feat(0, r, h) sits on lane (j, h): each half-wave contracts ITS 16 features with the
      // Jacobian rows parked in LDS, the two partial sums are added last (the order k_density_mlp uses)
      float gp[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * h;
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) gp[ax] += gf[0][r] * jac_at(ax * 32 + i);
      }
      float gw[3];
#pragma unroll
      for (int ax = 0; ax < 3; ++ax) {
        const float oth = __shfl_xor(gp[ax], 32, 64);
        gw[ax] = h == 0 ? gp[ax] + oth : oth + gp[ax];        // half 0's partial first on both halves
      }
#pragma unroll
      for (int s = 0; s < 32; ++s) act[s * 64] = hid[s];
      const float msq = zx * zx + zy * zy + zz * zz;
      float gzx = gw[0], gzy = gw[1], gzz = gw[2];
      if (msq > 1.0f) {
        const float rt = sqrtf(msq);
        const float s = (2.0f * rt - 1.0f) / msq;
        const float ds = (1.0f - rt) / (msq * msq);
        const float gz_dot = gw[0] * zx + gw[1] * zy + gw[2] * zz;
        gzx = s * gw[0] + 2.0f * ds * gz_dot * zx;
        gzy = s * gw[1] + 2.0f * ds * gz_dot * zy;
        gzz = s * gw[2] + 2.0f * ds * gz_dot * zz;
      }
      ngx = rc_div(gzx, a.contract_radius); ngy = rc_div(gzy, a.contract_radius); ngz = rc_div(gzz, a.contract_radius);
      neg_normalize(ngx, ngy, ngz);
    } else if constexpr (!FRONT && !DIRECT) {
      // the stream is consumed strictly in order: step the ring over the unused backward fragments
#pragma unroll
      for (int f = F_L2 + FR::B1; f < F_SH; ++f)
        if (f % kChunk == 0) ws_advance<NF>(ws, f / kChunk);
    }
  }
  RC_FSTAMP(9);
  if constexpr (FRONT) {
    // hand-over to the stages behind the sampler, in the layouts of the launch-per-stage front end
    if (ray_ok) {
      const int64_t np = a.n * 32, p = ray * 32 + j;
      for (int e2 = lane; e2 <= 32; e2 += 64) a.f_tdist[ray * 33 + e2] = s_td[e2];
      if (h == 0) {
        a.f_density[p] = density;
        a.f_means[p] = mx; a.f_means[np + p] = my; a.f_means[2 * np + p] = mz;
        a.f_normals_pred[p] = npx; a.f_normals_pred[np + p] = npy; a.f_normals_pred[2 * np + p] = npz;
        if constexpr (GRAD) {
          if (a.f_normals_grad) { a.f_normals_grad[p] = ngx; a.f_normals_grad[np + p] = ngy; a.f_normals_grad[2 * np + p] = ngz; }
        }
      }
      float* hb = a.f_hbuf + ray * (32 * 64) + lane;
#pragma unroll
      for (int s = 0; s < 32; ++s) hb[s * 64] = act[s * 64];                 // hidden feature, accumulator layout
#pragma unroll
      for (int s = 0; s < 16; ++s) a.f_app[(int64_t)(2 * s + h) * np + p] = act[(kAppTmp + s) * 64];   // feature 2 s + h of point j
    }
    return;
  }
  if constexpr (EXPORT) {
    if (ray_ok) {
      const int64_t np = a.n * 32, p = ray * 32 + j;
      for (int e2 = lane; e2 <= 32; e2 += 64) a.f_tdist[ray * 33 + e2] = s_td[e2];
      if (h == 0) {
        a.f_density[p] = density;
        a.f_means[p] = mx; a.f_means[np + p] = my; a.f_means[2 * np + p] = mz;
        a.f_normals_pred[p] = npx; a.f_normals_pred[np + p] = npy; a.f_normals_pred[2 * np + p] = npz;
      }
    }
  }
  // ------------------------------------------------------------------ shader on the 32 samples
#pragma unroll
  for (int s = 0; s < 16; ++s) act[(32 + s) * 64] = act[(kAppTmp + s) * 64];
  act[48 * 64] = h == 0 ? 1.0f : 0.0f;
#if defined(RC_ABL) && RC_ABL == 12
  ShadeOut so;
  for (int c = 0; c < 3; ++c) { so.rgb[c] = act[c * 64]; so.ad[c] = npx; so.idf[c] = npy; so.is[c] = npz; so.tint[c] = density; }
#else
  const ShadeOut so = shader_tile<F_SH, NF>(ws, act, lane, h, npx, npy, npz, a.viewdirs[3 * ray], a.viewdirs[3 * ray + 1],
                                            a.viewdirs[3 * ray + 2], reinterpret_cast<const RcIdeTable*>(a.ide_coef), a.sh);
#endif

  RC_FSTAMP(10);
  // ------------------------------------------------------------------ volume compositing (k_composite)
  const bool act_s = lane < 32;
  const float wnf = alpha_weight(density, t0, t1, dnorm, act_s, lane);
  auto store3 = [&](int id, float x, float y, float z) {
    if (lane == 0 && ray_ok && a.out.ptr[id]) { a.out.ptr[id][3 * ray] = x; a.out.ptr[id][3 * ray + 1] = y; a.out.ptr[id][3 * ray + 2] = z; }
  };
  auto store1 = [&](int id, float x) { if (lane == 0 && ray_ok && a.out.ptr[id]) a.out.ptr[id][ray] = x; };
  // every weighted sum of the ray in one batched butterfly
  enum { V_ACC = 0, V_RGB = 1, V_AD = 4, V_IDF = 7, V_IS = 10, V_TINT = 13, V_DIF = 16, V_IND = 19, V_MEAN = 22, V_RD = 25,
         V_LD = 26, V_NP = 27, V_NG = 30, V_LOGT = 33, V_COUNT = 34 };
  float v[V_COUNT];
  v[V_ACC] = wnf;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float v_rgb = act_s ? so.rgb[c] : 0.0f, v_ad = act_s ? so.ad[c] : 0.0f, v_id = act_s ? so.idf[c] : 0.0f;
    const float v_is = act_s ? so.is[c] : 0.0f, v_t = act_s ? so.tint[c] : 0.0f;
    v[V_RGB + c] = wnf * v_rgb;
    v[V_AD + c] = wnf * v_ad;
    v[V_IDF + c] = wnf * v_id;
    v[V_IS + c] = wnf * v_is;
    v[V_TINT + c] = wnf * v_t;
    v[V_DIF + c] = wnf * (v_ad + v_id);
    v[V_IND + c] = wnf * (v_id + v_is);
  }
  v[V_MEAN] = wnf * mx; v[V_MEAN + 1] = wnf * my; v[V_MEAN + 2] = wnf * mz;
  v[V_RD] = wnf * sqrtf((ox - mx) * (ox - mx) + (oy - my) * (oy - my) + (oz - mz) * (oz - mz));
  v[V_LD] = 0.0f;
  if (a.lights) {
    const float lx = a.lights[3 * ray], ly = a.lights[3 * ray + 1], lz = a.lights[3 * ray + 2];
    v[V_LD] = wnf * sqrtf((lx - mx) * (lx - mx) + (ly - my) * (ly - my) + (lz - mz) * (lz - mz));
  }
  v[V_NP] = wnf * npx; v[V_NP + 1] = wnf * npy; v[V_NP + 2] = wnf * npz;
  v[V_NG] = wnf * ngx; v[V_NG + 1] = wnf * ngy; v[V_NG + 2] = wnf * ngz;
  v[V_LOGT] = act_s ? wnf * logf(0.5f * (t0 + t1)) : 0.0f;
  wave_sum_n<V_COUNT>(v);
  const float accw = v[V_ACC];
  const float bgw = fmaxf(0.0f, 1.0f - accw) * a.bg;
  store3(RC_OUT_RGB, v[V_RGB] + bgw, v[V_RGB + 1] + bgw, v[V_RGB + 2] + bgw);
  store3(RC_OUT_DIRECT_RGB, v[V_AD], v[V_AD + 1], v[V_AD + 2]);
  store3(RC_OUT_INDIRECT_DIFFUSE_RGB, v[V_IDF], v[V_IDF + 1], v[V_IDF + 2]);
  store3(RC_OUT_INDIRECT_SPECULAR_RGB, v[V_IS], v[V_IS + 1], v[V_IS + 2]);
  store3(RC_OUT_SPECULAR_RGB, v[V_IS], v[V_IS + 1], v[V_IS + 2]);
  store3(RC_OUT_ALBEDO_RGB, v[V_TINT], v[V_TINT + 1], v[V_TINT + 2]);
  store3(RC_OUT_DIFFUSE_RGB, v[V_DIF], v[V_DIF + 1], v[V_DIF + 2]);
  store3(RC_OUT_INDIRECT_RGB, v[V_IND], v[V_IND + 1], v[V_IND + 2]);
  store3(RC_OUT_INDIRECT_OCC, accw, accw, accw);
  store1(RC_OUT_ACC, accw);
  store3(RC_OUT_MEANS, v[V_MEAN], v[V_MEAN + 1], v[V_MEAN + 2]);
  store1(RC_OUT_RAY_DISTS, v[V_RD]);
  if (a.lights) store1(RC_OUT_LIGHT_DISTS, v[V_LD]);
  store3(RC_OUT_NORMALS_PRED, v[V_NP], v[V_NP + 1], v[V_NP + 2]);
  if constexpr (GRAD) store3(RC_OUT_NORMALS, v[V_NG], v[V_NG + 1], v[V_NG + 2]);
  {
    const float e = v[V_LOGT] / fmaxf(RC_EPS, accw);
    float dm = expf(e);
    if (dm != dm) dm = 0.0f;                     // nan_to_num(x, copy=inf): nan -> 0.0 (rc_sample.hip, k_composite)
    dm = fminf(dm, RC_FMAX);
    dm = fminf(fmaxf(dm, s_td[0]), s_td[32]);
    store1(RC_OUT_DISTANCE_MEAN, dm);
    const float wn = wnf / fmaxf(RC_EPS, accw);
    const float incl = wave_scan_incl(wn, lane);
    if (lane == 0) s_cw[0] = 0.0f;
    if (lane < 31) s_cw[lane + 1] = fminf(1.0f, incl);
    if (lane == 0) s_cw[32] = 1.0f;
    lds_sync<false>();
    if (lane < 3 && ray_ok) {
      const float ps = a.pct[lane] / 100.0f;
      const float v = interp1(ps, s_cw, s_td, 33);
      const int id = lane == 0 ? RC_OUT_DISTANCE_PERCENTILE_5 : (lane == 1 ? RC_OUT_DISTANCE_MEDIAN : RC_OUT_DISTANCE_PERCENTILE_95);
      if (a.out.ptr[id]) a.out.ptr[id][ray] = v;
    }
  }
  RC_FSTAMP(11);
#ifdef RC_STAMPS
  if (lane == 0 && a.stamps && ray_ok) {
    unsigned long long* d = a.stamps + ray * 16;
    for (int i = 0; i < 12; ++i) d[i] = stamps[i];
    d[14] = rt0; d[15] = __builtin_amdgcn_s_memrealtime();
    d[12] = stamps[12]; d[13] = stamps[13];
  }
#endif
}

}  // namespace

#ifdef RC_STAMPS
static unsigned long long* g_fused_stamps = nullptr;
extern "C" void* rc_debug_fused_stamps() { return g_fused_stamps; }
#endif

__global__ void k_build_cells(const float* __restrict__ src, int N, int F, float* __restrict__ dst, int dst_stride, int dst_off) {
  const int M = N + 3;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)M * M * M * 8;
  if (i >= total) return;
  int64_t e;
  const bool inside = rc_cell_corner(N, i, &e);      // rc_pack_host.h (the same rule runs under the CPU sanitizers)
  for (int f = 0; f < F; ++f) dst[i * dst_stride + dst_off + f] = inside ? src[e * F + f] : 0.0f;
}

void rc_launch_build_cells(const float* src, int N, int F, float* dst, int dst_stride, int dst_off, hipStream_t stream) {
  const int64_t total = (int64_t)(N + 3) * (N + 3) * (N + 3) * 8;
  hipLaunchKernelGGL(k_build_cells, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, src, N, F, dst, dst_stride, dst_off);
}

int rc_fused_stream_offsets(int* l0, int* l1, int* l2, int* sh) {
  *l0 = F_L0; *l1 = F_L1; *l2 = F_L2; *sh = F_SH;
  return NF_FUSED;
}

void rc_launch_fused(const RcFusedLaunch& L, hipStream_t stream) {
  if (L.n <= 0) return;
  static std::atomic<uint64_t> prepared{0};
  const int lds = (kRingFloats + kWaves * (kShActSteps * 64 + kScratch)) * (int)sizeof(float);
  if (rc_first_use_on_device(prepared)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_cache_fused<true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_cache_fused<false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_cache_fused<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_cache_fused<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_cache_fused<true, false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_cache_fused<false, false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
#ifdef RC_FUSED_DIRECT_EXPERIMENT
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_cache_fused<true, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_cache_fused<false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
#endif
  }
  RcFusedArgs a{};
  a.origins = L.rays.origins; a.directions = L.rays.directions; a.viewdirs = L.rays.viewdirs; a.near = L.rays.near;
  a.far = L.rays.far; a.lights = L.rays.lights; a.n = L.n;
  for (int l = 0; l < 3; ++l) { a.jitter[l] = L.jitter[l]; a.us[l] = make_uspec(L.num_samples[l], L.jitter[l] != nullptr); }
  for (int g = 0; g < 4; ++g) a.grid[g] = *L.grid[g];
  for (int l = 0; l < RC_MAX_GRID_LEVELS; ++l) {
    a.pair_table[l] = L.pair_table[l];
    a.cell_table[0][l] = L.cell_table[0][l]; a.cell_table[1][l] = L.cell_table[1][l];
  }
  a.wstream = L.wstream; a.ide_coef = L.ide_coef;
  a.anneal = L.anneal; a.padding = L.padding; a.density_bias = L.density_bias; a.contract_radius = L.contract_radius; a.bg = L.bg;
  for (int i = 0; i < 3; ++i) a.pct[i] = L.pct[i];
  a.sh = ShaderConsts{L.roughness_bias, L.irradiance_bias, L.ambient_bias, L.rgb_max, L.slf_ambient_bias};
  a.out = L.out;
#ifdef RC_STAMPS
  {
    static unsigned long long* buf = nullptr; static int64_t cap = 0;
    // 2 x: the two-wave kernel keeps the stamps of a ray's second wave at [n + ray]
    if (cap < L.n) { if (buf) (void)hipFree(buf); (void)hipMalloc((void**)&buf, (size_t)L.n * 2 * 16 * 8); cap = L.n; }
    a.stamps = buf; g_fused_stamps = buf;
  }
#endif
  dim3 grid((unsigned)((L.n + kWaves - 1) / kWaves)), block(kWaves * 64);
  if (L.front) {
    a.use_raydist = L.use_raydist; a.raydist_p = L.raydist_p; a.raydist_premult = L.raydist_premult;
    a.y_max = L.raydist_p < 0.0f ? nextafterf((L.raydist_p - 1.0f) / L.raydist_p, -INFINITY) : 0.0f;       // as rc_launch_sample
    a.f_tdist = L.f_tdist; a.f_density = L.f_density; a.f_means = L.f_means; a.f_normals_pred = L.f_normals_pred;
    a.f_normals_grad = L.want_grad ? L.f_normals_grad : nullptr; a.f_hbuf = L.f_hbuf; a.f_app = L.f_app;
    if (L.want_grad) hipLaunchKernelGGL((k_cache_fused<true, true>), grid, block, lds, stream, a);
    else hipLaunchKernelGGL((k_cache_fused<false, true>), grid, block, lds, stream, a);
    return;
  }
  if (L.export_samples && L.team) {
    a.f_tdist = L.f_tdist; a.f_density = L.f_density; a.f_means = L.f_means; a.f_normals_pred = L.f_normals_pred;
  } else if (L.export_samples) {
    a.f_tdist = L.f_tdist; a.f_density = L.f_density; a.f_means = L.f_means; a.f_normals_pred = L.f_normals_pred;
    if (L.out.ptr[RC_OUT_NORMALS]) hipLaunchKernelGGL((k_cache_fused<true, false, false, true>), grid, block, lds, stream, a);
    else hipLaunchKernelGGL((k_cache_fused<false, false, false, true>), grid, block, lds, stream, a);
    return;
  }
#ifdef RC_FUSED_DIRECT_EXPERIMENT      // measured 145.3 us against 132.2 us through the ring (same box, bitwise equal results): not built by default
  if (L.direct) {
    if (L.out.ptr[RC_OUT_NORMALS]) hipLaunchKernelGGL((k_cache_fused<true, false, true>), grid, block, lds, stream, a);
    else hipLaunchKernelGGL((k_cache_fused<false, false, true>), grid, block, lds, stream, a);
    return;
  }
#endif
  if (L.team) {
    a.stagger_cycles = L.stagger_cycles;
    a.prio_mode = L.prio_mode;
    rc_launch_fused_team(a, L.out.ptr[RC_OUT_NORMALS] != nullptr, stream);
    return;
  }
  if (L.out.ptr[RC_OUT_NORMALS]) hipLaunchKernelGGL(k_cache_fused<true>, grid, block, lds, stream, a);
  else hipLaunchKernelGGL(k_cache_fused<false>, grid, block, lds, stream, a);
}
