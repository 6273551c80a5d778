// Dense cache MLPs on the gfx950 matrix cores, exact fp32 (v_mfma_f32_32x32x2_f32).
//
// Replaces (reference file:line):
//   DensityMLP.run_network / convert_raw_density / pred normals   internal/geometry.py:155-168, 318-341, 467-471
//   NeRFMLP heads, get_integrated_brdf, _predict_appearance_passive internal/nerf.py:461-482, 628-634, 940-1090
//   SurfaceLightFieldMLP.run_surface_lightfield_network + heads    internal/surface_light_field.py:480-499, 1011-1059
//   ref_utils.generate_ide_fn, reflect                              internal/ref_utils.py:25-42, 131-192
//
// Formulation.  Every layer is computed transposed, Y^T = W^T X^T, so that the 32 points of a wave
// sit on the MFMA's N/lane axis and the features live along registers:
//   A operand = weight fragment  (lane l: W[k = step row of half l>>5][n = 32 t + (l & 31)])
//   B operand = activations      (lane l: X[point l & 31][k = step row of half l>>5])
//   D         = 32 features x 32 points, lane l reg r holds feature (r&3) + 8 (r>>2) + 4 (l>>5).
// A layer's output accumulators are therefore *already* in B-operand form for the next layer: no
// transposes, no cross-lane traffic.  The k order inside a layer follows the accumulator layout;
// the host packs every weight matrix into exactly that fragment order once at load time
// (rc_api.hip), so a layer is one linear stream of 256-byte fragments.  Bias is one extra k-step
// (B = 1 on the low half-wave), ReLU is a VALU max on the accumulators.
// Activations are parked in a per-wave LDS slice between layers (each lane only ever re-reads what
// it wrote itself, so no barriers are needed) which keeps the register file for accumulators and
// the prefetched weight fragments.
#include "rc_dev_mlp.h"

using namespace rcdev;

namespace {

// ---------------------------------------------------------------------------------------------
// Density MLP: K -> 64 -> 64 -> 1 (+ 64 -> 3 predicted normals on the last level)
// ---------------------------------------------------------------------------------------------
constexpr int kDensActSteps = 33;

// GRAD (last level only): also the analytic normals -normalize(d raw / d x) of geometry.py:421-460:
// backward through the two hidden layers on the matrix cores (transposed weights, ReLU masks kept as
// one bit per accumulator register), then the trilinear Jacobian emitted by k_hashgrid_fwd<F, true>
// and the Jacobian of the contraction.
template <int KS0, bool GRAD>   // KS0: k-steps of layer 0 including the bias step
__global__ __launch_bounds__(kWaves * 64) void k_density_mlp(RcDensityMlpArgs a) {
  __shared__ __attribute__((aligned(16))) float ring[kRingFloats];
  __shared__ float lds[kWaves][kDensActSteps * 64];
  // the last level (32 grid features) also predicts the normals: 4 output rows, otherwise 1
  constexpr int NO = KS0 == 17 ? 4 : 1;
  constexpr int F_D0 = 0, F_D1 = rc_lfr32(KS0, 2), F_DO = F_D1 + rc_lfr32(33, 2), F_B1 = F_DO + rc_dfr32(NO, 2), F_B0 = F_B1 + rc_lfr32(32, 2),
                NF = GRAD ? F_B0 + rc_lfr32(32, 1) : F_B1;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t tile = (int64_t)blockIdx.x * kWaves + wave;
  const int64_t p0 = tile * 32;
  const int j = lane & 31, h = lane >> 5;
  const int64_t p = p0 + j;
  const bool valid = p < a.n;        // waves past the end stay alive for the workgroup barriers
  float* act = &lds[wave][lane];
  WStream ws{a.wstream, ring, lane, wave};
  ws_begin<NF>(ws);

  // stage the grid features (natural k pairs) + bias step
  // (all loads first, from clamped -- always valid -- addresses, the selects behind them: as "condition ? load : 0" each
  // of the up to 16 loads sat in its own branch with an s_waitcnt vmcnt(0) behind it, 16 dependent round trips)
  {
    float fv[KS0 - 1];
    const int64_t pp = valid ? p : 0;
#pragma unroll
    for (int s = 0; s < KS0 - 1; ++s) {
      const int k = 2 * s + h;
      fv[s] = a.feat[(int64_t)(k < a.K ? k : a.K - 1) * a.ld + pp];
    }
#pragma unroll
    for (int s = 0; s < KS0 - 1; ++s) {
      const int k = 2 * s + h;
      act[s * 64] = (valid && k < a.K) ? fv[s] : 0.0f;
    }
  }
  act[(KS0 - 1) * 64] = h == 0 ? 1.0f : 0.0f;

  f32x16 acc[2];
  acc[0] = zero16(); acc[1] = zero16();
  mlp_layer_d<2, KS0, F_D0, NF>(ws, act, acc);
  uint32_t m0 = 0, m1 = 0;           // ReLU masks, bit t*16+r
  if constexpr (GRAD) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) m0 |= (acc[t][r] > 0.0f ? 1u : 0u) << (t * 16 + r);
  }
  park<2, true>(acc, act, 0);
  act[32 * 64] = h == 0 ? 1.0f : 0.0f;

  acc[0] = zero16(); acc[1] = zero16();
  mlp_layer_d<2, 33, F_D1, NF>(ws, act, acc);
  if constexpr (GRAD) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) m1 |= (acc[t][r] > 0.0f ? 1u : 0u) << (t * 16 + r);
  }
  if (a.last && a.hbuf && p0 < a.n) {
    // hidden feature handed to the shader in accumulator (= B operand) layout
    float* hb = a.hbuf + tile * (32 * 64) + lane;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) hb[(t * 16 + r) * 64] = relu0(acc[t][r]);
  }

  float out[NO], wout[GRAD ? 32 : 1];
  dot_out1<2, NO, F_DO, NF, GRAD>(ws, acc, out, wout);     // output_density_layer (+ pred_normals_layer) on relu(acc)

  float cx = 0.0f, cy = 0.0f, cz = 0.0f;   // contracted position
  float zx = 0.0f, zy = 0.0f, zz = 0.0f;   // x / radius
  if (valid) {
    const int64_t q = a.src ? (int64_t)a.src[p] : p, ms = a.src ? a.n_src : a.n;
    zx = rc_div(a.means[q], a.contract_radius); zy = rc_div(a.means[ms + q], a.contract_radius); zz = rc_div(a.means[2 * ms + q], a.contract_radius);
    cx = a.means[q]; cy = a.means[ms + q]; cz = a.means[2 * ms + q];
    contract3(cx, cy, cz, a.contract_radius);
  }
  if (h == 0 && valid) {
    // convert_raw_density (geometry.py:318-341)
    const float raw = out[0];
    const bool inside = (cx > -a.bbox) & (cx < a.bbox) & (cy > -a.bbox) & (cy < a.bbox) & (cz > -a.bbox) & (cz < a.bbox);
    const float d = rc_safe_exp(raw + a.density_bias);
    a.density[p] = inside ? d : 0.0f;
    if (NO == 4 && a.last && a.normals_pred) {
      float gx = out[NO > 1 ? 1 : 0], gy = out[NO > 2 ? 2 : 0], gz = out[NO > 3 ? 3 : 0];
      neg_normalize(gx, gy, gz);
      a.normals_pred[p] = gx; a.normals_pred[a.n + p] = gy; a.normals_pred[2 * a.n + p] = gz;
    }
  }

  if constexpr (GRAD) {
    // d raw / d h1 = w_out (the accumulator-layout weights the forward dot product just used), masked by ReLU'(h1)
#pragma unroll
    for (int s = 0; s < 32; ++s) act[s * 64] = ((m1 >> s) & 1u) ? wout[s] : 0.0f;
    f32x16 g[2];
    g[0] = zero16(); g[1] = zero16();
    mlp_layer_d<2, 32, F_B1, NF>(ws, act, g);            // W1 . (.)   (transposed layer, no bias)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) act[(t * 16 + r) * 64] = ((m0 >> (t * 16 + r)) & 1u) ? g[t][r] : 0.0f;
    f32x16 gf[1];
    gf[0] = zero16();
    mlp_layer_d<1, 32, F_B0, NF>(ws, act, gf);           // W0 . (.) -> d raw / d feature (32, accumulator layout)
    // chain through the trilinear Jacobian: this lane owns features acc_feat(0, r, h)
    float gw[3] = {0.0f, 0.0f, 0.0f};
    if (valid) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * h;
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) gw[ax] += gf[0][r] * a.jac[(int64_t)(ax * 32 + i) * a.ld + p];
      }
    }
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) gw[ax] += __shfl_xor(gw[ax], 32, 64);
    if (h == 0 && valid) {
      // Jacobian of contract at z = x / radius: w = s(m) z, s = (2 sqrt(m) - 1) / m, m = max(1, |z|^2)
      const float msq = zx * zx + zy * zy + zz * zz;
      float gzx = gw[0], gzy = gw[1], gzz = gw[2];
      if (msq > 1.0f) {
        const float rt = sqrtf(msq);
        const float s = (2.0f * rt - 1.0f) / msq;
        const float ds = (1.0f - rt) / (msq * msq);          // ds/dm
        const float gz_dot = gw[0] * zx + gw[1] * zy + gw[2] * zz;
        gzx = s * gw[0] + 2.0f * ds * gz_dot * zx;
        gzy = s * gw[1] + 2.0f * ds * gz_dot * zy;
        gzz = s * gw[2] + 2.0f * ds * gz_dot * zz;
      }
      float nx = rc_div(gzx, a.contract_radius), ny = rc_div(gzy, a.contract_radius), nz = rc_div(gzz, a.contract_radius);
      neg_normalize(nx, ny, nz);
      a.normals_grad[p] = nx; a.normals_grad[a.n + p] = ny; a.normals_grad[2 * a.n + p] = nz;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Cache shader
// ---------------------------------------------------------------------------------------------
#ifdef RC_STAMPS
#define RC_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); stamps[i] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define RC_STAMP(i) do { } while (0)
#endif

__global__ __launch_bounds__(kWaves * 64) void k_cache_shader(RcShaderArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds_dyn[];
  split_exclusive_simd();
  constexpr int NF = ShaderFrags::COUNT;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t tile = (int64_t)blockIdx.x * kWaves + wave;
  const int64_t p0 = tile * 32;
  const int j = lane & 31, h = lane >> 5;
  const int64_t p = p0 + j;
  const bool valid = p < a.n;        // waves past the end stay alive for the workgroup barriers
  const int64_t pc = valid ? p : a.n - 1;
  const int64_t q = a.src ? (int64_t)a.src[pc] : pc;       // source point of the last level
  const int64_t ray = pc / a.samples_per_ray;
  float* ring = lds_dyn;
  float* act = lds_dyn + kRingFloats + wave * (kShActSteps * 64) + lane;
  WStream ws{a.wstream, ring, lane, wave};
#ifdef RC_STAMPS
  unsigned long long stamps[12];
  const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
#endif
  RC_STAMP(0);
  ws_begin<NF>(ws);

  // ---- stage feature = [density feature (64, accumulator order) | appearance grid (32)] + bias
  {
    const float* hb = a.hbuf + (q >> 5) * (32 * 64) + (q & 31) + 32 * h;
#pragma unroll
    for (int s = 0; s < 32; ++s) act[s * 64] = hb[s * 64];
#pragma unroll
    for (int s = 0; s < 16; ++s) act[(32 + s) * 64] = a.app[(int64_t)(2 * s + h) * a.n + pc];
    act[48 * 64] = h == 0 ? 1.0f : 0.0f;
  }
  RC_STAMP(1);
  const ShaderConsts kc{a.roughness_bias, a.irradiance_bias, a.ambient_bias, a.rgb_max, a.slf_ambient_bias};
  const ShadeOut so = shader_tile<0, NF>(ws, act, lane, h, a.normals_pred[q], a.normals_pred[a.n_src + q],
                                        a.normals_pred[2 * a.n_src + q], a.viewdirs[3 * ray], a.viewdirs[3 * ray + 1],
                                        a.viewdirs[3 * ray + 2], reinterpret_cast<const RcIdeTable*>(a.ide_coef), kc
#ifdef RC_STAMPS
                                        , stamps + 2
#endif
                                        );
  RC_STAMP(11);
#ifdef RC_STAMPS
  if (lane == 0 && a.debug) {
    const unsigned long long rt1 = __builtin_amdgcn_s_memrealtime();
    unsigned long long* d = reinterpret_cast<unsigned long long*>(a.debug) + tile * 16;
    for (int i = 0; i < 12; ++i) d[i] = stamps[i];
    d[12] = rt0; d[13] = rt1;
  }
#endif
  if (h == 0 && valid) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      a.shade[(int64_t)(RC_SH_RGB + c) * a.n + p] = so.rgb[c];
      a.shade[(int64_t)(RC_SH_AD + c) * a.n + p] = so.ad[c];
      a.shade[(int64_t)(RC_SH_ID + c) * a.n + p] = so.idf[c];
      a.shade[(int64_t)(RC_SH_IS + c) * a.n + p] = so.is[c];
      a.shade[(int64_t)(RC_SH_TINT + c) * a.n + p] = so.tint[c];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Model-level EnvMap (background of secondary rays): pos_enc(dir, 0, 4) [27] -> 256 -> 256 -> 256
// -> concat input (283) -> 128 -> rgba; rgb = clip(softplus(raw + rgb_bias), 0, inf).
// Replaces Model._handle_env_map -> SurfaceLightFieldMLP.__call__ as configured by
// NeRFModel.env_map_params (internal/models.py:360-421, internal/surface_light_field.py:480-499,
// 1011-1058, internal/coord.py:298-312).  One wave = 32 rays.  The 256-wide activations take 128 LDS steps per wave
// (32 KiB; the bias step of those layers takes its B operand from a register, mlp_bias_step): four waves (one per
// SIMD) sit next to a ring of 2 x 8 KiB chunks (144 KiB; a ring of 16-KiB chunks measures the same).
// The 256 -> 256 layers run as two 64-step halves: as ONE 128-step loop of 8 tiles the body exceeds the compiler's
// full-unroll budget, the loop stays rolled, the three operand register sets turn into runtime-indexed registers
// (s_set_gpr_idx) with one s_waitcnt per LDS read, and the kernel takes 160 us instead of 110 (tools/isa_scan.py
// looks for exactly that).
// ---------------------------------------------------------------------------------------------
constexpr int kEnvWaves = 4;
constexpr int kEnvActSteps = 128;
constexpr int kEnvChunk = 32;                        // fragments per chunk of this kernel's ring
constexpr int kEnvRingFloats = 2 * kEnvChunk * 64;

__global__ __launch_bounds__(kEnvWaves * 64) void k_envmap(RcEnvMapArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds_dyn[];
  split_exclusive_simd();
  constexpr int KS_IN = 15;   // 14 natural pairs of the 27 inputs (+1 zero pad) + bias
  constexpr int F_E0 = 0, F_E1 = F_E0 + rc_lfr(KS_IN, 8), F_E2 = F_E1 + rc_lfr(129, 8), F_EB = F_E2 + rc_lfr(129, 8),
                F_EI = F_EB + rc_lfr(128, 4), F_EO = F_EI + rc_lfr(KS_IN, 4), NF = F_EO + rc_lfr(65, 1);
  constexpr int H8 = rc_lfr(64, 8);          // fragments of half a 256 -> 256 layer (64 k-steps, 8 tiles)
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t tile = (int64_t)blockIdx.x * kEnvWaves + wave;
  const int j = lane & 31, h = lane >> 5;
  const int64_t p = tile * 32 + j;
  const bool valid = p < a.n;
  const int64_t pc = valid ? p : a.n - 1;
  float* ring = lds_dyn;
  float* act = lds_dyn + kEnvRingFloats + wave * (kEnvActSteps * 64) + lane;
  WStream ws{a.wstream, ring, lane, wave};
  ws_begin<NF, kEnvWaves, kEnvChunk>(ws);

  // pos_enc(x, 0, 4, append_identity): [x(3), sin(2^j x)(12), sin(2^j x + pi/2)(12)]
  const float d[3] = {a.viewdirs[3 * pc], a.viewdirs[3 * pc + 1], a.viewdirs[3 * pc + 2]};
  auto enc = [&](int k) -> float {
    if (k < 3) return d[k];
    if (k >= 27) return 0.0f;
    const int q = (k - 3) % 12, second = (k - 3) / 12;
    const float sx = d[q % 3] * (float)(1 << (q / 3));
    return sinf(second ? sx + 1.5707963267948966f : sx);
  };
  auto stage_inputs = [&]() {
#pragma unroll
    for (int s = 0; s < KS_IN - 1; ++s) act[s * 64] = enc(2 * s + h);
    act[(KS_IN - 1) * 64] = h == 0 ? 1.0f : 0.0f;
  };
  stage_inputs();
  f32x16 acc[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) acc[t] = zero16();
  mlp_layer<8, KS_IN, F_E0, NF, 1, kEnvWaves, kEnvChunk>(ws, act, acc);
  park<8, true>(acc, act, 0);
#pragma unroll
  for (int t = 0; t < 8; ++t) acc[t] = zero16();
  mlp_layer<8, 64, F_E1, NF, 1, kEnvWaves, kEnvChunk>(ws, act, acc);
  mlp_layer<8, 64, F_E1 + H8, NF, 1, kEnvWaves, kEnvChunk>(ws, act + 64 * 64, acc);
  mlp_bias_step<8, F_E1 + 2 * H8, NF, kEnvWaves, kEnvChunk>(ws, acc);
  park<8, true>(acc, act, 0);
#pragma unroll
  for (int t = 0; t < 8; ++t) acc[t] = zero16();
  mlp_layer<8, 64, F_E2, NF, 1, kEnvWaves, kEnvChunk>(ws, act, acc);
  mlp_layer<8, 64, F_E2 + H8, NF, 1, kEnvWaves, kEnvChunk>(ws, act + 64 * 64, acc);
  mlp_bias_step<8, F_E2 + 2 * H8, NF, kEnvWaves, kEnvChunk>(ws, acc);
  park<8, true>(acc, act, 0);
  // layer_bottleneck on concat([x2 (256), inputs (27)]): x part, then the re-staged input part (+bias)
  f32x16 bt[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) bt[t] = zero16();
  mlp_layer<4, 128, F_EB, NF, 2, kEnvWaves, kEnvChunk>(ws, act, bt);
  stage_inputs();
  mlp_layer<4, KS_IN, F_EI, NF, 2, kEnvWaves, kEnvChunk>(ws, act, bt);
  park<4, true>(bt, act, 0);
  act[64 * 64] = h == 0 ? 1.0f : 0.0f;
  f32x16 o[1];
  o[0] = zero16();
  mlp_layer<1, 65, F_EO, NF, 8, kEnvWaves, kEnvChunk>(ws, act, o);
  if (h == 0 && valid) {
#pragma unroll
    for (int c = 0; c < 3; ++c) a.env_rgb[3 * p + c] = fmaxf(softplus(o[0][c] + a.rgb_bias), 0.0f);
  }
}

}  // namespace

void rc_launch_density_mlp(const RcDensityMlpArgs& a, hipStream_t stream) {
  if (a.n <= 0) return;
  const int64_t tiles = (a.n + 31) / 32;
  dim3 grid((unsigned)((tiles + kWaves - 1) / kWaves)), block(kWaves * 64);
  const int ks0 = (a.K + 1) / 2 + 1;
  switch (ks0) {
    case 4: hipLaunchKernelGGL((k_density_mlp<4, false>), grid, block, 0, stream, a); break;
    case 5: hipLaunchKernelGGL((k_density_mlp<5, false>), grid, block, 0, stream, a); break;
    case 17:
      if (a.jac && a.normals_grad) hipLaunchKernelGGL((k_density_mlp<17, true>), grid, block, 0, stream, a);
      else hipLaunchKernelGGL((k_density_mlp<17, false>), grid, block, 0, stream, a);
      break;
    default: break;   // rejected by the host before getting here
  }
}

int rc_shader_lds_bytes() { return (kRingFloats + kWaves * kShActSteps * 64) * (int)sizeof(float); }
int rc_weight_chunk_floats() { return kChunk * 64; }

// The shader's per-wave activation slices exceed the default 64 KiB dynamic-LDS limit.
void rc_shader_prepare() {
  static std::atomic<uint64_t> done{0};
  if (!rc_first_use_on_device(done)) return;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_cache_shader),
                            hipFuncAttributeMaxDynamicSharedMemorySize, rc_shader_lds_bytes());
}

void rc_launch_shader(const RcShaderArgs& a, hipStream_t stream) {
  if (a.n <= 0) return;
  const int64_t tiles = (a.n + 31) / 32;
  dim3 grid((unsigned)((tiles + kWaves - 1) / kWaves)), block(kWaves * 64);
  hipLaunchKernelGGL(k_cache_shader, grid, block, rc_shader_lds_bytes(), stream, a);
}

void rc_launch_envmap(const RcEnvMapArgs& a, hipStream_t stream) {
  if (a.n <= 0) return;
  static std::atomic<uint64_t> prepared{0};
  const int lds = (kEnvRingFloats + kEnvWaves * kEnvActSteps * 64) * (int)sizeof(float);
  if (rc_first_use_on_device(prepared)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_envmap), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  }
  const int64_t tiles = (a.n + 31) / 32;
  dim3 grid((unsigned)((tiles + kEnvWaves - 1) / kEnvWaves)), block(kEnvWaves * 64);
  hipLaunchKernelGGL(k_envmap, grid, block, lds, stream, a);
}
