// Time-resolved radiance cache (cornell configuration): the TransientNeRFMLP shader and the per-bin
// compositing of the TransientVolumeIntegrator, fp32 MFMA, gfx950.
//
//   k_transient_shader   internal/nerf.py:561-689 (predict_appearance), :691-938 (_predict_appearance_active),
//                        :1097-1191 (_compute_light_radiance), :1422-1497 (_compute_direct_lighting),
//                        :484-538 (get_brdf_light), :461-482 (get_integrated_brdf), :1775-1797 (get_indirect trunk),
//                        internal/surface_light_field.py:480-499, 845-1035 (TransientSurfaceLightFieldMLP trunk)
//   k_transient_bins     the two 2100-wide heads (nerf.py:1795, surface_light_field.py:1036-1058),
//                        nerf.py:1692-1771 (_compute_indirect_lighting), render_utils.py:1699-1767 (zero_invalid_bins),
//                        internal/render.py:250-507 (volumetric_transient_rendering, shift_direct,
//                        shift_map_coordinates), internal/integration.py:343-551
//
// The [rays, samples, 700, 3] per-sample histograms (8.4 KB per sample and key) are never written: the
// second kernel gives one wavefront to each ray, evaluates the wide heads tile by tile as X W (samples in
// the MFMA rows, 32 histogram entries in the columns, so a lane owns ONE entry and 16 of the 32 samples),
// applies activation / travel-time masks / clip, and folds the tile straight into the ray's time-shifted
// histogram in LDS and into the unshifted per-bin composites.
#include "rc_dev_mlp.h"
#include "rc_dev_sample.h"

using namespace rcdev;

namespace {

// ---------------------------------------------------------------------------------------------
// shader: activation steps of a wave (32 samples)
// ---------------------------------------------------------------------------------------------
// (the linear shader bottleneck is folded into its consumers on the host, as in the steady-state shader:
// SLF layer_0 / layer_bottleneck input part, integrated_brdf_layers_0, brdf_layers_0 read the 96-wide feature)
constexpr int kTAct = 101;
constexpr int T_BIAS0 = 48;    // heads / irradiance-net bias step (1 | 0)
constexpr int T_LENC = 49;     // [49, 57)  pos_enc(lights): 15 values + an unused slot
constexpr int T_IRR = 57;      // [57, 90)  irradiance-net hidden layer (32 steps + bias)
constexpr int T_IDE = 48;      // [48, 84)  IDE of the reflection direction (after the irradiance net is done)
constexpr int T_WENC = 84;     // [84, 92)  pos_enc(contract(lights)): 15 values + bias slot (= 1)
constexpr int T_DOT = 92;      // (n.(-v) | 1)
constexpr int T_BENC = 93;     // [93, 101) BRDF encoding: 15 values + bias slot (= 1)
constexpr int T_SCR = 48;      // [48, 81)  scratch of the IBRDF / BRDF tails (IDE and light encoding are dead then)

struct TFrags {
  static constexpr int F_H = 0, F_IR0 = F_H + rc_lfr(49, 1), F_IR1 = F_IR0 + rc_lfr(57, 2), F_S0 = F_IR1 + rc_lfr(33, 2), F_I0 = F_S0 + rc_lfr(92, 8),
                       F_I1 = F_I0 + rc_lfr(49, 2), F_IO = F_I1 + rc_lfr(33, 2), F_B0 = F_IO + rc_lfr(33, 1), F_B1 = F_B0 + rc_lfr(56, 2), F_BO = F_B1 + rc_lfr(33, 2),
                       F_S1 = F_BO + rc_lfr(33, 1), F_S2 = F_S1 + rc_lfr(65, 4), F_SB = F_S2 + rc_lfr(65, 4), COUNT = F_SB + rc_lfr(64, 4);
};

// coord.pos_enc(x, 0, 2, append_identity=True) of a 3-vector: [x | sin(x) | sin(2x) | sin(x + pi/2) | sin(2x + pi/2)]
__device__ __forceinline__ void pos_enc2(const float (&x)[3], float (&e)[16], float slot15) {
  const float hp = 1.5707963267948966f;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float s1 = x[c] * 1.0f, s2 = x[c] * 2.0f;
    e[c] = x[c];
    e[3 + c] = sinf(s1);
    e[6 + c] = sinf(s2);
    e[9 + c] = sinf(s1 + hp);
    e[12 + c] = sinf(s2 + hp);
  }
  e[15] = slot15;
}
// 16 slots as 8 natural activation steps: slot q at step base + q / 2, half-wave q & 1
__device__ __forceinline__ void stage16(float* act, int base, int h, const float (&e)[16]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) act[(base + i) * 64] = h == 0 ? e[2 * i] : e[2 * i + 1];
}

__global__ __launch_bounds__(kWaves * 64) void k_transient_shader(RcTransShaderArgs a) {
  split_exclusive_simd();
  extern __shared__ __attribute__((aligned(16))) float lds_dyn[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int j = lane & 31, h = lane >> 5;
  const int64_t ntiles = (a.n + 31) / 32;
  int64_t tile = (int64_t)blockIdx.x * kWaves + wave;
  const bool tile_ok = tile < ntiles;
  if (!tile_ok) tile = ntiles - 1;
  float* ring = lds_dyn;
  float* act = lds_dyn + kRingFloats + wave * (kTAct * 64) + lane;
  constexpr int NF = TFrags::COUNT;
  WStream ws{a.wstream, ring, lane, wave};
  ws_begin<NF>(ws);

  const int64_t p = tile * 32 + j;
  const bool valid = p < a.n;
  const int64_t pc = valid ? p : a.n - 1;
  const int64_t ray = pc / a.samples_per_ray;
  // ---- inputs: [hidden density feature (accumulator order) | appearance features | bias | pos_enc(lights)]
  {
    const float* hb = a.hbuf + (pc >> 5) * (32 * 64) + (pc & 31) + 32 * h;     // [tile][32 steps][64 lanes] (k_density_mlp)
#pragma unroll
    for (int s = 0; s < 32; ++s) act[s * 64] = hb[s * 64];
  }
#pragma unroll
  for (int s = 0; s < 16; ++s) act[(32 + s) * 64] = a.app[(int64_t)(2 * s + h) * a.n + pc];
  act[T_BIAS0 * 64] = h == 0 ? 1.0f : 0.0f;
  const float lx = a.lights[3 * ray], ly = a.lights[3 * ray + 1], lz = a.lights[3 * ray + 2];
  {
    const float l3[3] = {lx, ly, lz};
    float e[16];
    pos_enc2(l3, e, 0.0f);
    stage16(act, T_LENC, h, e);
  }
  // ---- heads tile (0 roughness, 1-3 tint, 4-6 direct tint, 7-9 albedo)
  f32x16 hd[1];
  hd[0] = zero16();
  mlp_layer<1, 49, TFrags::F_H, NF>(ws, act, hd);
  // ---- irradiance trunk on [feature | pos_enc(lights)] (nerf.py:1781-1791): the inputs are still in place
  {
    f32x16 ir[2];
    ir[0] = zero16(); ir[1] = zero16();
    mlp_layer<2, 57, TFrags::F_IR0, NF>(ws, act, ir);
    park<2, true>(ir, act, T_IRR);
    act[(T_IRR + 32) * 64] = h == 0 ? 1.0f : 0.0f;
    ir[0] = zero16(); ir[1] = zero16();
    mlp_layer<2, 33, TFrags::F_IR1, NF>(ws, act + T_IRR * 64, ir);
    if (tile_ok) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) a.irr_feat[(tile * 32 + t * 16 + r) * 64 + lane] = relu0(ir[t][r]);
    }
  }
  const float rough = softplus(hd[0][0] + a.roughness_bias);                   // nerf.py:633-634
  const float tint[3] = {sigmoidf(hd[0][1]), sigmoidf(hd[0][2]), sigmoidf(hd[0][3])};          // nerf.py:1697
  const float dtint[3] = {sigmoidf(hd[0][4]), sigmoidf(hd[0][5]), sigmoidf(hd[0][6])};         // nerf.py:1469
  const float albedo[3] = {softplus(hd[0][7] + a.albedo_bias), softplus(hd[0][8] + a.albedo_bias),
                           softplus(hd[0][9] + a.albedo_bias)};                                // nerf.py:1466-1468
  // ---- geometry of this lane's sample
  const float mx = a.means[pc], my = a.means[a.n + pc], mz = a.means[2 * a.n + pc];
  const float nx = a.normals[pc], ny = a.normals[a.n + pc], nz = a.normals[2 * a.n + pc];
  const float vx = a.viewdirs[3 * ray], vy = a.viewdirs[3 * ray + 1], vz = a.viewdirs[3 * ray + 2];
  const float ox = a.origins[3 * ray], oy = a.origins[3 * ray + 1], oz = a.origins[3 * ray + 2];
  const float lox = lx - mx, loy = ly - my, loz = lz - mz;                    // nerf.py:717-722
  const float ldist = sqrtf(lox * lox + loy * loy + loz * loz);
  const float ldn = fmaxf(ldist, 1e-5f);
  const float ldx = lox / ldn, ldy = loy / ldn, ldz = loz / ldn;
  const float rdist = sqrtf((ox - mx) * (ox - mx) + (oy - my) * (oy - my) + (oz - mz) * (oz - mz));
  const float cox = ox - a.cam_origins[3 * ray], coy = oy - a.cam_origins[3 * ray + 1], coz = oz - a.cam_origins[3 * ray + 2];
  const float camdist = rdist + sqrtf(cox * cox + coy * coy + coz * coz);     // render_utils.py:1731-1740
  // light radiance (nerf.py:1142-1171): power = safe_exp(light_power), 1/d^2, light_zero
  float radiance = 1.0f * expf(fminf(a.light_power, 70.0f));
  if (a.use_falloff) radiance = radiance * (1.0f / fmaxf(ldist * ldist, 1e-5f));
  if (a.light_zero && ldist < a.light_near) radiance = 0.0f;
  const float radiance_before_occ = radiance;
  const float n_dot_l = fmaxf(0.0f, nx * ldx + ny * ldy + nz * ldz);          // nerf.py:744
  float occ = 0.0f;
  if (a.occ) {
    // nerf.py:1300-1340: acc of the shadow ray; no occlusion when the light sits on the camera; thresholded
    occ = a.occ[pc];
    const float bx = lx - ox, by = ly - oy, bz = lz - oz;
    if (sqrtf(bx * bx + by * by + bz * bz) < 1e-3f) occ = 0.0f;
    if (occ <= a.occ_threshold) occ = 0.0f;
  }
  if (n_dot_l <= 0.0f) occ = 1.0f;                                            // nerf.py:764
  radiance = radiance * (1.0f - occ);
  const float dotp = nx * (-vx) + ny * (-vy) + nz * (-vz);
  {
    // BRDF-light input (nerf.py:495-523): sort(n.v, n.l), n.h with h = normalize(-v + l)
    const float hx = -vx + ldx, hy = -vy + ldy, hz = -vz + ldz;
    const float hn = sqrtf(hx * hx + hy * hy + hz * hz);
    const float ndl = nx * ldx + ny * ldy + nz * ldz;
    const float ndh = nx * (hx / hn) + ny * (hy / hn) + nz * (hz / hn);
    const float b3[3] = {fminf(dotp, ndl), fmaxf(dotp, ndl), ndh};
    float e[16];
    pos_enc2(b3, e, 1.0f);
    stage16(act, T_BENC, h, e);
    // lights for the surface light field: warp_fn = contract (surface_light_field.py:1022-1024)
    float w3[3] = {lx, ly, lz};
    contract3(w3[0], w3[1], w3[2], a.contract_radius);
    pos_enc2(w3, e, 1.0f);
    stage16(act, T_WENC, h, e);
    act[T_DOT * 64] = h == 0 ? dotp : 1.0f;
  }
  {
    // IDE of reflect(-v, n) (ref_utils.py:25-42, 155-190), as in shader_tile
    const float rx = 2.0f * dotp * nx - (-vx), ry = 2.0f * dotp * ny - (-vy), rz = 2.0f * dotp * nz - (-vz);
    const RcIdeTable* tb = reinterpret_cast<const RcIdeTable*>(a.ide_coef);
    float zp[RC_IDE_ZPOW];
    zp[0] = 1.0f;
#pragma unroll
    for (int k = 1; k < RC_IDE_ZPOW; ++k) zp[k] = zp[k - 1] * rz;
    float cpw[RC_IDE_ZPOW];
    {
      float cre = 1.0f, cim = 0.0f;
      cpw[0] = h == 0 ? cre : cim;
#pragma unroll
      for (int m = 1; m < RC_IDE_ZPOW; ++m) {
        const float nre = cre * rx - cim * ry;
        const float nim = cre * ry + cim * rx;
        cre = nre; cim = nim;
        cpw[m] = h == 0 ? cre : cim;
      }
    }
#pragma unroll
    for (int i = 0; i < RC_IDE_TERMS; ++i) {
      const int l = ide_l(i), m = ide_m(i);
      float poly = 0.0f;
#pragma unroll
      for (int k = 0; k < RC_IDE_ZPOW; ++k)
        if (k <= l - m && ((l - m - k) & 1) == 0) poly = poly + zp[k] * tb->coef[i][k];
      const float att = expf(-(0.5f * (float)(l * (l + 1))) * rough);
      act[(T_IDE + i) * 64] = (cpw[m] * poly) * att;
    }
  }
  // ---- SLF layer_0 (tiles 0-3) + input part of layer_bottleneck (tiles 4-7) over [feature | IDE | lights]
  f32x16 s0[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) s0[t] = zero16();
  mlp_layer<8, 92, TFrags::F_S0, NF>(ws, act, s0);
  // ---- integrated BRDF (nerf.py:461-482)
  float ibrdf;
  {
    f32x16 ib[2];
    ib[0] = zero16(); ib[1] = zero16();
    mlp_layer<2, 48, TFrags::F_I0, NF>(ws, act, ib);
    mlp_layer<2, 1, TFrags::F_I0 + rc_lfr(48, 2), NF>(ws, act + T_DOT * 64, ib);
    park<2, true>(ib, act, T_SCR);
    act[(T_SCR + 32) * 64] = h == 0 ? 1.0f : 0.0f;
    ib[0] = zero16(); ib[1] = zero16();
    mlp_layer<2, 33, TFrags::F_I1, NF>(ws, act + T_SCR * 64, ib);
    park<2, true>(ib, act, T_SCR);
    f32x16 o[1];
    o[0] = zero16();
    mlp_layer<1, 33, TFrags::F_IO, NF>(ws, act + T_SCR * 64, o);
    ibrdf = sigmoidf(o[0][0] + 1.0986123f);
  }
  // ---- BRDF towards the light (nerf.py:484-538): [bottleneck | enc(sorted dots, n.h)] -> 64 -> 64 -> 1 (first layer folded)
  float lbrdf;
  {
    f32x16 b[2];
    b[0] = zero16(); b[1] = zero16();
    mlp_layer<2, 48, TFrags::F_B0, NF>(ws, act, b);
    mlp_layer<2, 8, TFrags::F_B0 + rc_lfr(48, 2), NF>(ws, act + T_BENC * 64, b);
    park<2, true>(b, act, T_SCR);
    act[(T_SCR + 32) * 64] = h == 0 ? 1.0f : 0.0f;
    b[0] = zero16(); b[1] = zero16();
    mlp_layer<2, 33, TFrags::F_B1, NF>(ws, act + T_SCR * 64, b);
    park<2, true>(b, act, T_SCR);
    f32x16 o[1];
    o[0] = zero16();
    mlp_layer<1, 33, TFrags::F_BO, NF>(ws, act + T_SCR * 64, o);
    lbrdf = softplus(o[0][0] + a.brdf_bias);
    if (n_dot_l == 0.0f) lbrdf = 0.0f;                                        // nerf.py:1478-1481
  }
  // ---- SLF trunk: layer_1, layer_2, layer_bottleneck -> the 128-wide feature of the wide output layer
  {
    f32x16 acc[4] = {s0[0], s0[1], s0[2], s0[3]};
    f32x16 skip[4] = {s0[4], s0[5], s0[6], s0[7]};
    park<4, true>(acc, act, 0);
    act[64 * 64] = h == 0 ? 1.0f : 0.0f;
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = zero16();
    mlp_layer<4, 65, TFrags::F_S1, NF>(ws, act, acc);
    park<4, true>(acc, act, 0);
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = zero16();
    mlp_layer<4, 65, TFrags::F_S2, NF>(ws, act, acc);
    park<4, true>(acc, act, 0);
    mlp_layer<4, 64, TFrags::F_SB, NF>(ws, act, skip);
    if (tile_ok) {
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) a.slf_feat[(tile * 64 + t * 16 + r) * 64 + lane] = relu0(skip[t][r]);
    }
  }
  if (valid && tile_ok && h == 0) {
    float* o = a.tshade + p;
    const int64_t n = a.n;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      // nerf.py:1483-1488 (direct_diffuse, direct_specular clipped to [0, rgb_max])
      o[(RC_TS_DD + c) * n] = fminf(fmaxf(((albedo[c] * n_dot_l) * radiance) / 3.14159265358979323846f, 0.0f), a.rgb_max);
      o[(RC_TS_DS + c) * n] = fminf(fmaxf((dtint[c] * lbrdf) * radiance, 0.0f), a.rgb_max);
      o[(RC_TS_ALBEDO + c) * n] = albedo[c];
      o[(RC_TS_TIB + c) * n] = tint[c] * ibrdf;
    }
    o[RC_TS_ROUGH * n] = rough;
    o[RC_TS_NDOTL * n] = n_dot_l;
    o[RC_TS_IRRAD * n] = (n_dot_l * radiance_before_occ) / 3.14159265358979323846f;      // nerf.py:921-925
    o[RC_TS_OCC * n] = occ;
    o[RC_TS_LDIST * n] = ldist;
    o[RC_TS_RDIST * n] = rdist;
    o[RC_TS_CAMDIST * n] = camdist;
  }
}

// Shadow rays of _compute_occlusions (nerf.py:1233-1288; get_secondary_rays with the ActiveSampler,
// render_utils.py:462-478, 927-1056): one ray per shaded sample towards the light.
__global__ void k_shadow_rays(RcShadowRayArgs a) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= a.n) return;
  const int64_t ray = p / a.samples_per_ray;
  const float mx = a.means[p], my = a.means[a.n + p], mz = a.means[2 * a.n + p];
  const float nx = a.normals[p], ny = a.normals[a.n + p], nz = a.normals[2 * a.n + p];
  const float lx = a.lights[3 * ray], ly = a.lights[3 * ray + 1], lz = a.lights[3 * ray + 2];
  const float ox = lx - mx, oy = ly - my, oz = lz - mz;
  const float dist = sqrtf(ox * ox + oy * oy + oz * oz);
  const float dn = fmaxf(dist, 1e-5f);
  a.origins[3 * p] = mx + nx * a.normal_eps; a.origins[3 * p + 1] = my + ny * a.normal_eps; a.origins[3 * p + 2] = mz + nz * a.normal_eps;
  a.dirs[3 * p] = ox / dn; a.dirs[3 * p + 1] = oy / dn; a.dirs[3 * p + 2] = oz / dn;
  a.near[p] = a.shadow_near;
  a.far[p] = fminf(fmaxf(dist - a.light_near, a.shadow_near), a.shadow_far);       // nerf.py:1276-1281
  a.out_normals[3 * p] = nx; a.out_normals[3 * p + 1] = ny; a.out_normals[3 * p + 2] = nz;
  a.out_lights[3 * p] = lx; a.out_lights[3 * p + 1] = ly; a.out_lights[3 * p + 2] = lz;
}

// ---------------------------------------------------------------------------------------------
// per-bin heads + compositing: one wavefront per ray
// ---------------------------------------------------------------------------------------------
constexpr int kBins = 700;
constexpr int kHist = kBins * 3;                 // 2100 histogram entries per ray, entry = bin * 3 + channel
constexpr int kTilesB = (kHist + 31) / 32;       // 66 column tiles
constexpr int kTileFrags = 65 + 33;              // SLF output layer (128 + bias) | transient_indirect_layer (64 + bias)
// split form (rc_pack_host.h): 9 + 5 blocks of 8 k-steps x 3 pieces x 4 fragments = 168 fragments per tile
constexpr int kTileFragsSplit = (rc_lfr(65, 1) + rc_lfr(33, 1));
constexpr int kChunksPerTile = kRcSplit ? 3 : 2;
constexpr int kFragsPerTile = kChunksPerTile * kChunk;   // padded to whole chunks of the LDS ring: the chunk seams sit at
                                                 // compile-time positions of a tile (no per-fragment seam test)
static_assert(!kRcSplit || kTileFragsSplit <= kFragsPerTile, "a tile's pieces fit its chunks");
static_assert(kTileFrags <= kFragsPerTile, "a tile's fragments fit its two chunks");
constexpr int kBinFrags = kTilesB * kFragsPerTile;
constexpr int kSP = 10;                          // per-sample parameters kept in LDS
constexpr int kHistPad = kHist + 32;             // + one dummy slot per lane (out-of-range targets read-add-write there)
constexpr int kMaxTaps = 32;                     // temporal filter taps handled by the unrolled window
constexpr int kPadD = 3 * (kMaxTaps / 2);        // zeros on both sides of the direct histogram: the filter window needs no range test
constexpr int kWaveLds = 2 * kHistPad + (kHist + 2 * kPadD) + kSP * 32 + 6 * 32;   // floats: indirect hist (one per half-wave), padded direct hist, params, bin sums

// softplus on the hardware transcendentals (v_exp_f32 / v_log_f32, about 1 ulp each): the per-bin heads evaluate
// 2 x 2100 of them per sample, which is what bounds k_transient_bins.  log1p(e) for small e by its series.
// The per-bin heads' form: max(x, 0) + log(1 + exp(-|x|)) on the two hardware transcendentals, WITHOUT the series branch
// for tiny exp(-|x|).  1 + e rounds e away below 6e-8 and carries an absolute error of <= 6e-8 below 1e-3: 4e-8 on a
// softplus that is then scaled by indirect_scale (0.05) into per-bin radiance compared at 2e-6 -- three orders of
// magnitude inside the budget, for 4 vector instructions less per evaluation (2 x 16 x 2100 evaluations per ray).
__device__ __forceinline__ float softplus_bins(float x) {
  const float e = __builtin_amdgcn_exp2f(-fabsf(x) * 1.44269504088896341f);
  return fmaxf(x, 0.0f) + __builtin_amdgcn_logf(1.0f + e) * 0.693147180559945309f;
}
__device__ __forceinline__ float softplus_hw(float x) {
  // straight-line: v_exp_f32 / v_log_f32 are base 2; both forms of log1p are evaluated and selected
  const float e = __builtin_amdgcn_exp2f(-fabsf(x) * 1.44269504088896341f);
  const float l = __builtin_amdgcn_logf(1.0f + e) * 0.693147180559945309f;
  const float s = e * (1.0f - 0.5f * e);
  return fmaxf(x, 0.0f) + (e < 1.0e-3f ? s : l);
}

__device__ __forceinline__ void ws_issue_rt(const WStream& w, int c, int nf) {
#pragma unroll
  for (int k = 0; k < kChunk / 4 / kWaves; ++k) {
    const int i = w.wave + kWaves * k;
    const int frag0 = c * kChunk + 4 * i;
    if (frag0 < nf) {
      const float* src = w.g + (size_t)frag0 * 64 + w.lane * 4;
      float* dst = w.ring + ((c & 1) * kChunk + 4 * i) * 64;
      __builtin_amdgcn_global_load_lds((const void*)src, (lds_void_ptr)dst, 16, 0, 0);
    }
  }
}
// fragment J of column tile T of the stream.  T is a run-time value, J a constant once the callers' loops are
// unrolled: a tile is exactly two chunks of the ring, so the seam test folds away and the waits / barriers sit at
// J = 0 and J = kChunk of every tile
__device__ __forceinline__ void ws_tile_seam(const WStream& w, int T, int J, int nf, int c0) {
  if (J % kChunk == 0) {
    const int c = kChunksPerTile * T + J / kChunk;
    if (c > c0) {         // c0: the first chunk of this launch's tile range (in flight since the prologue)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if ((c + 1) * kChunk < nf) ws_issue_rt(w, c + 1, nf);
    }
  }
}
__device__ __forceinline__ float ws_tile_frag(const WStream& w, int T, int J, int nf, int c0) {
  static_assert(kRcSplit || kFragsPerTile == 2 * kChunk, "fp32 form: tile = the two chunks of the ring, J is the ring slot");
  ws_tile_seam(w, T, J, nf, c0);
  return w.ring[J * 64 + w.lane];
}
// split form: the 1-KiB piece at fragments [J, J + 4) of tile T (three chunks per tile: the ring half of a chunk follows
// the parity of its number, a run-time value)
__device__ __forceinline__ u32x4 ws_tile_piece(const WStream& w, int T, int J, int nf, int c0) {
  ws_tile_seam(w, T, J, nf, c0);
  const int c = kChunksPerTile * T + J / kChunk;
  return lds_piece(w.ring + ((c & 1) * kChunk + J % kChunk) * 64 + w.lane * 4);
}

// Split form of tile_xw2 below: the activations of the two heads were split once per ray (xsp: 9 blocks, xip: 5), the
// weights of the tile come as pieces in the order [SLF head blocks 0-8 | irradiance head blocks 0-4]; six products per
// block and head, one accumulation chain per head (a chain of this MFMA needs no second one beside it).
__device__ __forceinline__ void tile_xw2_split(const WStream& w, int T, int nf, int c0, const u32x4 (&xsp)[9][3], const u32x4 (&xip)[5][3],
                                               f32x16& as, f32x16& ai) {
  constexpr int NCELL = 14;
  u32x4 b[2][3];
  auto load = [&](int cell, int buf) {
#pragma unroll
    for (int p = 0; p < 3; ++p) b[buf][p] = ws_tile_piece(w, T, (cell * 3 + p) * 4, nf, c0);
  };
  load(0, 0);
#pragma unroll
  for (int cell = 0; cell < NCELL; ++cell) {
    if (cell + 1 < NCELL) load(cell + 1, (cell + 1) & 1);
    if (cell < 9) mfma_split6(xsp[cell], b[cell & 1], as);
    else mfma_split6(xip[cell - 9], b[cell & 1], ai);
  }
}


// X W of one column tile for both heads: 65 k-steps of the SLF head (xs -> as) and 33 of the irradiance head
// (xi -> ai).  The two accumulator chains are interleaved 4 : 2 (a chain's next MFMA waits for its previous one; the
// other chain fills the gap), and the host packs the tile's fragments in exactly this order:
//   16 groups of [as 4g .. 4g+3 | ai 2g, 2g+1], then [as 64 | ai 32].
__device__ __forceinline__ void tile_xw2(const WStream& w, int T, int nf, int c0, const float (&xs)[65], const float (&xi)[33],
                                         f32x16& as, f32x16& ai) {
  constexpr int NG = 17;
  float b[3][6];                 // operands two groups ahead (three register sets, like mlp_layer)
  auto load = [&](int g, int buf) {
#pragma unroll
    for (int q = 0; q < 6; ++q)
      if (g < 16 || q < 2) b[buf][q] = ws_tile_frag(w, T, g * 6 + q, nf, c0);
  };
  load(0, 0);
  load(1, 1);
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    if (g + 2 < NG) load(g + 2, (g + 2) % 3);
    __builtin_amdgcn_sched_barrier(0);
    const float (&bb)[6] = b[g % 3];
    if (g < 16) {
      as = __builtin_amdgcn_mfma_f32_32x32x2f32(xs[4 * g + 0], bb[0], as, 0, 0, 0);
      ai = __builtin_amdgcn_mfma_f32_32x32x2f32(xi[2 * g + 0], bb[4], ai, 0, 0, 0);
      as = __builtin_amdgcn_mfma_f32_32x32x2f32(xs[4 * g + 1], bb[1], as, 0, 0, 0);
      as = __builtin_amdgcn_mfma_f32_32x32x2f32(xs[4 * g + 2], bb[2], as, 0, 0, 0);
      ai = __builtin_amdgcn_mfma_f32_32x32x2f32(xi[2 * g + 1], bb[5], ai, 0, 0, 0);
      as = __builtin_amdgcn_mfma_f32_32x32x2f32(xs[4 * g + 3], bb[3], as, 0, 0, 0);
    } else {
      as = __builtin_amdgcn_mfma_f32_32x32x2f32(xs[64], bb[0], as, 0, 0, 0);
      ai = __builtin_amdgcn_mfma_f32_32x32x2f32(xi[32], bb[1], ai, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

#ifdef RC_STAMPS
#define RC_BSTAMP(v) do { __builtin_amdgcn_sched_barrier(0); v = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define RC_BSTAMP(v) do { } while (0)
#endif

__global__ __launch_bounds__(kWaves * 64) void k_transient_bins(RcTransBinsArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds_dyn[];
  split_exclusive_simd();
#ifdef RC_STAMPS
  const unsigned long long st_kernel_begin = __builtin_amdgcn_s_memtime();
#endif
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int fl = lane & 31, h = lane >> 5;
  int64_t ray = (int64_t)blockIdx.x * kWaves + wave;
  const bool ray_ok = ray < a.n_rays;
  if (!ray_ok) ray = a.n_rays - 1;
  const int64_t n = a.n_rays * 32;                 // samples
  float* ring = lds_dyn;
  float* wl = lds_dyn + kRingFloats + wave * kWaveLds;
  float* hist_i = wl;                              // time-shifted indirect histogram, one per half-wave (added at the end)
  float* hist_d = wl + 2 * kHistPad + kPadD;       // direct histogram (before the temporal filter), zero padded
  float* sp = hist_d + kHist + kPadD;              // [kSP][32] per-sample parameters
  float* bsum = sp + kSP * 32;                     // [6][32] per-sample sums over bins (diffuse rgb, specular rgb)
  enum { P_W = 0, P_LDIST, P_CAMDIST, P_TIB0, P_TIB1, P_TIB2, P_DIND, P_KILL, P_LO, P_HI };
  WStream ws{a.wstream, ring, lane, wave};
  __shared__ int s_win[2 * kWaves];               // per ray of the workgroup: first / last bin any of its samples keeps
  for (int e = lane; e < 2 * kHistPad + kHist + 2 * kPadD; e += 64) wl[e] = 0.0f;
  int win_lo = kBins, win_hi = -1;
  if (lane < 32) {
    const int64_t p = ray * 32 + lane;
    // the eight per-sample inputs in one batch of loads (as "LDS slot = load" they were eight dependent round trips: the
    // loads were not moved over the LDS stores between them)
    const float ld = a.tshade[RC_TS_LDIST * n + p], in_w = a.weights[p], in_cam = a.tshade[RC_TS_CAMDIST * n + p];
    const float in_t0 = a.tshade[(RC_TS_TIB + 0) * n + p], in_t1 = a.tshade[(RC_TS_TIB + 1) * n + p],
                in_t2 = a.tshade[(RC_TS_TIB + 2) * n + p], in_rd = a.tshade[RC_TS_RDIST * n + p];
    sp[P_W * 32 + lane] = in_w;
    sp[P_LDIST * 32 + lane] = ld;
    sp[P_CAMDIST * 32 + lane] = in_cam;
    sp[P_TIB0 * 32 + lane] = in_t0;
    sp[P_TIB1 * 32 + lane] = in_t1;
    sp[P_TIB2 * 32 + lane] = in_t2;
    // bins_move / exposure_time (render.py:483): ray_dist + shift, divided by the exposure
    sp[P_DIND * 32 + lane] = (in_rd + a.shift) / a.exposure;
    const bool kill = a.light_zero && ld < a.light_near;                               // render_utils.py:1750-1760
    sp[P_KILL * 32 + lane] = kill ? 1.0f : 0.0f;
    // Window of bins that survive zero_invalid_bins (render_utils.py:1699-1767), once per sample (lane = sample).
    // Both travel-time tests are monotone in the bin index, so each is a bound: bins >= lo pass
    // "(b + thr) * e < light_dist" (too close), bins <= hi pass "b * e + cam_dist > max_dists" (too far); the bounds
    // are settled with the very comparisons of the reference.
    const float cdist = in_cam;
    auto close = [&](int b) { return (float)(b + a.bin_zero_threshold_light) * a.exposure < ld; };
    auto far = [&](int b) { return ((float)b * a.exposure + cdist) > a.max_dists; };
    int lo = (int)ceilf(ld / a.exposure) - a.bin_zero_threshold_light;
    lo = min(max(lo, 0), kBins);
    while (lo > 0 && !close(lo - 1)) --lo;
    while (lo < kBins && close(lo)) ++lo;
    int hi = (int)floorf((a.max_dists - cdist) / a.exposure);
    hi = min(max(hi, -1), kBins - 1);
    while (hi < kBins - 1 && !far(hi + 1)) ++hi;
    while (hi >= 0 && far(hi)) --hi;
    if (kill) { lo = kBins; hi = -1; }
    sp[P_LO * 32 + lane] = __int_as_float(lo);
    sp[P_HI * 32 + lane] = __int_as_float(hi);
    if (lo <= hi) { win_lo = lo; win_hi = hi; }
  }
  // Column tiles outside [first, last] bin that ANY sample of the workgroup's rays keeps hold exact zeros on every lane
  // (diff = spec = 0 there): their 98 MFMAs and their epilogue are not run, the weight stream starts at the first tile
  // of the range.  The tile behind the range still runs: it places what the range's last three entries carry over.
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    win_lo = min(win_lo, __shfl_xor(win_lo, d, 64));
    win_hi = max(win_hi, __shfl_xor(win_hi, d, 64));
  }
  if (lane == 0) { s_win[2 * wave] = win_lo; s_win[2 * wave + 1] = win_hi; }
  __syncthreads();
  int blo = kBins, bhi = -1;
#pragma unroll
  for (int q = 0; q < kWaves; ++q) { blo = min(blo, s_win[2 * q]); bhi = max(bhi, s_win[2 * q + 1]); }
  // tile of entry f = 3 b + c is f / 32; groups of three tiles (T3) keep the channel phase of a lane a constant
  const int T3_lo = blo <= bhi ? ((3 * blo) / 32) / 3 : 0;
  const int T3_hi = blo <= bhi ? min(kTilesB / 3 - 1, ((3 * bhi + 2) / 32 + 1) / 3) : -1;
  const int c0 = kChunksPerTile * (3 * T3_lo);
  if (T3_hi >= T3_lo) ws_issue_rt(ws, c0, kBinFrags);
  // activations of the two output layers (this ray's 32 samples), with the bias step
  float xs[65], xi[33];
#pragma unroll
  for (int s = 0; s < 64; ++s) xs[s] = a.slf_feat[(ray * 64 + s) * 64 + lane];
  xs[64] = h == 0 ? 1.0f : 0.0f;
#pragma unroll
  for (int s = 0; s < 32; ++s) xi[s] = a.irr_feat[(ray * 32 + s) * 64 + lane];
  xi[32] = h == 0 ? 1.0f : 0.0f;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (T3_hi >= T3_lo && (c0 + 1) * kChunk < kBinFrags) ws_issue_rt(ws, c0 + 1, kBinFrags);
  // split form: the three bf16 pieces of every activation, once for all 66 column tiles
  u32x4 xsp[kRcSplit ? 9 : 1][3], xip[kRcSplit ? 5 : 1][3];
  if constexpr (kRcSplit) {
#pragma unroll
    for (int q = 0; q < 9; ++q) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = 8 * q + j < 65 ? xs[8 * q + j] : 0.0f;
      split8(v, xsp[q]);
    }
#pragma unroll
    for (int q = 0; q < 5; ++q) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = 8 * q + j < 33 ? xi[8 * q + j] : 0.0f;
      split8(v, xip[q]);
    }
  }

  // per-sample sums over the bins: by tile phase u = T % 3 (the channel of a lane's entry is (2 u + fl) % 3)
  float sd[3][16], ss[3][16];
#pragma unroll
  for (int u = 0; u < 3; ++u)
#pragma unroll
    for (int r = 0; r < 16; ++r) { sd[u][r] = 0.0f; ss[u][r] = 0.0f; }

  // Per-sample terms of this lane's 16 samples, kept in registers for all 66 tiles: weight, time shift, bin window
  float rw[16], rdm[16];
  int rlo[16], rspan[16], rfl3[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int i = (r & 3) + 8 * (r >> 2) + 4 * h;
    rw[r] = sp[P_W * 32 + i];
    rdm[r] = sp[P_DIND * 32 + i];
    rfl3[r] = 3 * (int)floorf(rdm[r]);                               // entry offset of the time shift (3 entries per bin)
    // window [lo, hi] as ONE unsigned compare in the epilogue: (unsigned)(b - lo) <= (unsigned)(hi - lo); an empty window
    // (hi < lo) is stored as lo = 2^30, span = 0 -- no bin passes
    const int lo_i = __float_as_int(sp[P_LO * 32 + i]), hi_i = __float_as_int(sp[P_HI * 32 + i]);
    rlo[r] = hi_i >= lo_i ? lo_i : (1 << 30);
    rspan[r] = hi_i >= lo_i ? hi_i - lo_i : 0;
  }
  // value of the tile before, per sample (see the epilogue)
  float cval[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) cval[r] = 0.0f;
  float* hist_h = hist_i + h * kHistPad;
  const int nb_addr = 4 * (fl < 3 ? lane + 29 : lane - 3);     // ds_bpermute byte address of the neighbour entry's lane
#ifdef RC_STAMPS
  unsigned long long st_mfma = 0, st_epi = 0;
  const unsigned long long st_begin = __builtin_amdgcn_s_memtime();
#endif
  // A tile = its 98 MFMAs, then its epilogue.  Running the epilogue of tile T next to the MFMAs of tile T + 1 in one
  // straight-line block was tried and is slower: at one wave per SIMD a wave's own VALU work does not run in the shadow
  // of its MFMAs on gfx950 (tools/micro/mfma_valu_overlap.hip: interleaved = 85 % of the sum), it takes a second wave.
  auto tile_body = [&](const int T, const int u, float (&sdu)[16], float (&ssu)[16]) __attribute__((always_inline)) {
    {
#ifdef RC_STAMPS
      unsigned long long q0, q1, q2;
#endif
      RC_BSTAMP(q0);
      f32x16 as = zero16(), ai = zero16();
      if constexpr (kRcSplit) tile_xw2_split(ws, T, kBinFrags, c0, reinterpret_cast<const u32x4 (&)[9][3]>(xsp), reinterpret_cast<const u32x4 (&)[5][3]>(xip), as, ai);
      else tile_xw2(ws, T, kBinFrags, c0, xs, xi, as, ai);
      RC_BSTAMP(q1);
      const int f = T * 32 + fl;                   // histogram entry of this lane
      const bool fok = f < kHist;
      const int b = f / 3, c = f - 3 * b;
      const float b_f = (float)b, bm1_f = (float)(b - 1);
      const int b_live = fok ? b : -(1 << 29);     // an entry beyond the histogram passes no sample's window test
      // this lane's column of the per-sample tint table, kept in a register for the tile's 16 samples (left to itself the
      // compiler re-derives it with a 64-bit multiply-add per sample)
      int tib_off = (P_TIB0 + c) * 32 + 4 * h;     // (the OFFSET is pinned, not a pointer: a pinned pointer loses its LDS address space)
      asm volatile("" : "+v"(tib_off));
      float cd = 0.0f, cs = 0.0f;                  // unshifted composites over this half-wave's 16 samples
#pragma unroll
      for (int r = 0; r < 16; ++r) {
#if defined(RC_ABL) && RC_ABL == 1
        cd += as[r] + ai[r];      // ablation: MFMAs only
        continue;
#endif
        // A sample whose window misses every bin of this tile contributes exact zeros: skip it for the whole wave
        // (unless the tile before left it a value to place, see below).
        const bool live = (unsigned)(b_live - rlo[r]) <= (unsigned)rspan[r];
        const bool carry_in = cval[r] != 0.0f;       // cval is kept on the lanes that serve the next tile only (fl >= 29)
        if (__builtin_amdgcn_ballot_w64(live | carry_in) == 0ull) continue;
        const float w = rw[r];
        // shift_map_coordinates (render.py:480-496): out[y] = in(y - d), linear, zero outside.  Entry b of this sample
        // reaches y0 = b + floor(d) and y0 + 1, and y0 also receives the second part of entry b - 1 -- the lane three
        // entries down (same channel).  Each lane therefore owns ONE target, y0, and adds both parts to it in one
        // read-add-write of its half-wave's histogram (32 different targets per half-wave and sample; plain
        // read-add-write measured 2x faster than ds_add_f32).  The read is issued first and the transcendentals
        // below cover its latency.  The lanes of the first three entries take the neighbour's value from the tile
        // before (cval); a target outside the histogram goes to the lane's dummy slot.
        const float dmove = rdm[r];
        // target entry 3 y0 + c = f + 3 floor(d); inside the histogram exactly when 0 <= y0 < kBins (c < 3)
        const int e0 = f + rfl3[r];
        const bool yok = (unsigned)e0 < (unsigned)kHist;
        float* slot = hist_h + (yok ? e0 : kHist + fl);
#if !(defined(RC_ABL) && RC_ABL == 2)
        const float old = *slot;
#endif
        const float tib = sp[tib_off + (r & 3) + 8 * (r >> 2)];
        // nerf.py:1795-1797 and :1712-1719: softplus(. + irradiance_bias) * indirect_scale;
        // surface_light_field.py:1037-1058 and nerf.py:1721-1723: tint * ibrdf * clip(softplus(. + rgb_bias), 0) * scale
#if defined(RC_ABL) && RC_ABL == 3
        float diff = (ai[r] + a.irradiance_bias) * a.indirect_scale;       // ablation: no transcendentals
        const float ref = fmaxf(1.0f * as[r] + a.slf_rgb_bias, 0.0f);
#else
        float diff = softplus_bins(ai[r] + a.irradiance_bias) * a.indirect_scale;
        const float ref = fmaxf(softplus_bins(1.0f * as[r] + a.slf_rgb_bias), 0.0f);
        // keep the value in front of the `live` select below: the optimizer otherwise sinks this chain into a divergent
        // branch on `live` (s_and_saveexec / s_cbranch / s_or exec per iteration) where one v_cndmask does
        asm volatile("" : "+v"(diff));
#endif
        float spec = (tib * ref) * a.indirect_scale;
        // jnp.clip(x, 0, rgb_max) (nerf.py:1757-1758) as ONE v_med3_f32: the median of (x, 0, rgb_max) is the clamp for
        // 0 <= rgb_max and an ordered x (softplus >= 0 is never NaN here); fmaxf + fminf were two instructions plus a
        // canonicalising v_max of the scalar bound each
        diff = live ? __builtin_amdgcn_fmed3f(diff, 0.0f, a.rgb_max) : 0.0f;
        spec = live ? __builtin_amdgcn_fmed3f(spec, 0.0f, a.rgb_max) : 0.0f;
        sdu[r] += diff; ssu[r] += spec;
        cd += w * diff; cs += w * spec;
        const float val = w * (diff + spec);
        // entry f - 3: three lanes down in this tile (fl >= 3), or lanes 29-31 of the tile before (fl < 3) -- one
        // bpermute: the lanes that serve the second case (fl >= 29) are never a source of the first
        const float nb = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(
            nb_addr, __builtin_bit_cast(int, fl >= 29 ? cval[r] : val)));
        cval[r] = fl >= 29 ? val : 0.0f;
        // The weights are those the target bin computes (coordinate t = y0 - d, i0 = floor(t), fw = t - i0; weight
        // 1 - fw to source bin i0 and fw to i0 + 1): wa for source b, wb for source b - 1.  t lies in [b - 1, b], so
        // i0 is b - 1 (wa = fw, wb = 1 - fw) or, when d is integral or t rounds up to b, b (wa = 1 - fw, wb = 0);
        // the reference's other terms are exact zeros.
        // (t in [b - 1, b]: floor(t) is b exactly when t == b, else b - 1; both are constants of the lane for the tile)
        const float t = (b_f + floorf(dmove)) - dmove;        // (float)y0, y0 = b + floor(d): both are small integers, the sum is exact
        const bool at_b = t == b_f;
        const float fw = at_b ? 0.0f : t - bm1_f;
        const float wa = at_b ? 1.0f : fw;
        const float wb = at_b ? 0.0f : 1.0f - fw;
#if defined(RC_ABL) && RC_ABL == 2
        cd += yok ? val * wa + nb * wb : 0.0f;      // ablation: no histogram update
#else
        *slot = old + (val * wa + nb * wb);
#endif
      }
      cd += __shfl_xor(cd, 32, 64);
      cs += __shfl_xor(cs, 32, 64);
      if (h == 0 && fok && ray_ok) {
        if (a.out_ti_diffuse) a.out_ti_diffuse[ray * kHist + f] = cd;
        if (a.out_ti_specular) a.out_ti_specular[ray * kHist + f] = cs;
      }
      RC_BSTAMP(q2);
#ifdef RC_STAMPS
      st_mfma += q1 - q0; st_epi += q2 - q1;
#endif
    }
  };
  // the unshifted composites of the tiles that are not run: exact zeros
  if (ray_ok && (a.out_ti_diffuse || a.out_ti_specular)) {
    const int f_lo = 96 * T3_lo, f_hi = min(kHist, 96 * (T3_hi + 1));
    for (int f = lane; f < kHist; f += 64) {
      if (f >= f_lo && f < f_hi) continue;
      if (a.out_ti_diffuse) a.out_ti_diffuse[ray * kHist + f] = 0.0f;
      if (a.out_ti_specular) a.out_ti_specular[ray * kHist + f] = 0.0f;
    }
  }
  for (int T3 = T3_lo; T3 <= T3_hi; ++T3) {
    tile_body(T3 * 3 + 0, 0, sd[0], ss[0]);
    tile_body(T3 * 3 + 1, 1, sd[1], ss[1]);
    tile_body(T3 * 3 + 2, 2, sd[2], ss[2]);
  }
#ifdef RC_STAMPS
  const unsigned long long st_tiles_end = __builtin_amdgcn_s_memtime();
#endif
  // ---- per-sample sums over the bins: pick the channel of each tile phase, add up the 32 entry lanes
  {
    float v[6][16];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float d = 0.0f, s = 0.0f;
#pragma unroll
        for (int u = 0; u < 3; ++u) {
          const bool mine = (2 * u + fl) % 3 == c;
          d += mine ? sd[u][r] : 0.0f;
          s += mine ? ss[u][r] : 0.0f;
        }
        v[c][r] = d; v[3 + c][r] = s;
      }
    // sums over the 32 entry lanes of each half-wave on DPP moves (rc_dev_sample.h wave_sum_step: butterfly inside the
    // rows of 16, then row_bcast:15 adds row 0 into row 1 and row 2 into row 3): the totals sit on lanes 16-31 / 48-63
#pragma unroll
    for (int st = 0; st < 5; ++st)
#pragma unroll
      for (int q = 0; q < 6; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) v[q][r] = wave_sum_step(v[q][r], st);
    if (fl == 16) {
#pragma unroll
      for (int q = 0; q < 6; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) bsum[q * 32 + (r & 3) + 8 * (r >> 2) + 4 * h] = v[q][r];
    }
  }
  lds_sync<false>();
#ifdef RC_STAMPS
  unsigned long long st_t1, st_t2;
  RC_BSTAMP(st_t1);
#endif
  // ---- direct light: scatter at (ray_dist + light_dist) / exposure with floor / ceil weights (render.py:436-477).
  //      The reference indexes the flattened [rays * bins] array: bins >= 700 of the previous ray of the batch
  //      land at the start of this ray's histogram.
  {
    // lanes 0-31 take the samples of the previous ray (their bins >= 700), lanes 32-63 those of this ray
    const int64_t sr = ray - 1 + h;
    int il = -1, ih = -1;
    float vl[3] = {0.0f, 0.0f, 0.0f}, vh[3] = {0.0f, 0.0f, 0.0f};
    if (sr >= 0) {
      const int64_t p = sr * 32 + fl;
      const float d = (a.tshade[RC_TS_LDIST * n + p] + a.tshade[RC_TS_RDIST * n + p]) / a.exposure + a.shift / a.exposure;
      const float low = fmaxf(floorf(d), 0.0f), high = ceilf(d);
      const float w_high = d - low, w_low = 1.0f - w_high;
      const int off = h == 0 ? kBins : 0;
      il = (int)low - off; ih = (int)high - off;
      if (il < 0 || il >= kBins) il = -1;
      if (ih < 0 || ih >= kBins) ih = -1;
      const float w = a.weights[p];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float val = w * (a.tshade[(RC_TS_DD + c) * n + p] + a.tshade[(RC_TS_DS + c) * n + p]);
        vl[c] = val * w_low; vh[c] = val * w_high;
      }
    }
    // One LDS float add per (target kind, channel) for all 64 samples at once: lane = sample, all the low targets first,
    // then all the high ones -- the order of the reference's two scatter-adds (render.py:453-477: `.at[indices_low].add`,
    // then `.at[indices_high].add`).  Lanes that hit one address are served one after the other by the LDS.  (Rounds 1-3
    // walked the 64 samples in order, eight v_readlane broadcasts and two three-lane adds each: 512 + 128 instructions.)
#pragma unroll
    for (int c = 0; c < 3; ++c)
      if (il >= 0) atomicAdd(&hist_d[il * 3 + c], vl[c]);
#pragma unroll
    for (int c = 0; c < 3; ++c)
      if (ih >= 0) atomicAdd(&hist_d[ih * 3 + c], vh[c]);
  }
  lds_sync<false>();
#ifdef RC_STAMPS
  RC_BSTAMP(st_t2);
#endif
  // ---- temporal filter on the direct part (render.py:406-417), outputs, sums over bins
  float sum_d[3] = {0.0f, 0.0f, 0.0f}, sum_i[3] = {0.0f, 0.0f, 0.0f};
  // the taps in registers (wave-uniform loads); the window of an entry is n_taps LDS reads 3 floats apart around it,
  // the zero padding of hist_d stands in for the bins outside [0, 700) (adding tap * 0 leaves the sum unchanged)
  float tp[kMaxTaps];
#pragma unroll
  for (int k = 0; k < kMaxTaps; ++k) tp[k] = k < a.n_taps ? a.taps[k] : 0.0f;
  const int half_taps = (a.n_taps - 1) / 2;
  for (int e = lane; e < kHist; e += 64) {
    const int b = e / 3, c = e - 3 * b;
    float dv;
    if (a.n_taps > kMaxTaps) {
      dv = 0.0f;
      for (int k = 0; k < a.n_taps; ++k) {
        const int yb = b - (k - half_taps);
        if (yb >= 0 && yb < kBins) dv += a.taps[k] * hist_d[yb * 3 + c];
      }
#if defined(RC_ABL) && RC_ABL == 6
    } else if (a.n_taps < 0) {                   // ablation: no filter
#else
    } else if (a.n_taps > 0) {
#endif
      dv = 0.0f;
      const float* win = hist_d + e + 3 * half_taps;        // tap k reads bin b - (k - half)
      if (a.n_taps == 25) {
#pragma unroll
        for (int k = 0; k < 25; ++k) dv += tp[k] * win[-3 * k];
      } else {
#pragma unroll
        for (int k = 0; k < kMaxTaps; ++k)
          if (k < a.n_taps) dv += tp[k] * win[-3 * k];
      }
    } else {
      dv = hist_d[e];
    }
    const float iv = hist_i[e] + hist_i[kHistPad + e];
#if defined(RC_ABL) && RC_ABL == 5
    if (ray_ok && dv + iv == 12345.678f) {       // ablation: no output stores
#else
    if (ray_ok) {
#endif
      if (a.out_rgb) a.out_rgb[ray * kHist + e] = dv + iv;
      if (a.out_direct) a.out_direct[ray * kHist + e] = dv;
      if (a.out_indirect) a.out_indirect[ray * kHist + e] = iv;
    }
    sum_d[c] += dv; sum_i[c] += iv;
  }
  {
    float v[6] = {sum_d[0], sum_d[1], sum_d[2], sum_i[0], sum_i[1], sum_i[2]};
    wave_sum_n<6>(v);
    if (lane == 0 && ray_ok) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        if (a.out_direct_rgb) a.out_direct_rgb[3 * ray + c] = v[c];
        if (a.out_indirect_rgb) a.out_indirect_rgb[3 * ray + c] = v[3 + c];
        if (a.out_integrated_rgb) a.out_integrated_rgb[3 * ray + c] = v[c] + v[3 + c];
      }
    }
  }
  // ---- per-sample extras composited with the sample weights (integration.py:420-470, render.py:283-299)
  {
    const bool act_s = lane < 32;
    const int64_t p = ray * 32 + (act_s ? lane : 0);
    const float w = act_s ? a.weights[p] : 0.0f;
    auto ts = [&](int ch) { return a.tshade[(int64_t)ch * n + p]; };
    enum { E_DIFF = 0, E_SPEC = 3, E_ALB = 6, E_OCC = 9, E_IRR = 10, E_NDL = 11, E_DD = 12, E_DS = 15, E_ID = 18, E_IS = 21,
           E_DVIZ = 24, E_W = 27, E_COUNT = 28 };
    float v[E_COUNT];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float dd = ts(RC_TS_DD + c), ds = ts(RC_TS_DS + c);
      const float idf = bsum[c * 32 + (lane & 31)], isp = bsum[(3 + c) * 32 + (lane & 31)];
      v[E_DIFF + c] = w * (dd + idf + 0.0f);                                   // nerf.py:875
      v[E_SPEC + c] = w * (ds + isp + 0.0f);
      v[E_ALB + c] = w * ts(RC_TS_ALBEDO + c);
      v[E_DD + c] = w * dd; v[E_DS + c] = w * ds;
      v[E_ID + c] = w * (idf + 0.0f); v[E_IS + c] = w * (isp + 0.0f);
      v[E_DVIZ + c] = act_s ? dd + ds : 0.0f;                                  // direct_rgb_viz: plain sum over samples
    }
    v[E_OCC] = w * ts(RC_TS_OCC);
    v[E_IRR] = w * ts(RC_TS_IRRAD);
    v[E_NDL] = w * ts(RC_TS_NDOTL);
    v[E_W] = w;
    wave_sum_n<E_COUNT>(v);
    if (lane == 0 && ray_ok) {
      auto st3 = [&](float* o, float x, float y, float z) { if (o) { o[3 * ray] = x; o[3 * ray + 1] = y; o[3 * ray + 2] = z; } };
      st3(a.out_diffuse_rgb, v[E_DIFF], v[E_DIFF + 1], v[E_DIFF + 2]);
      st3(a.out_specular_rgb, v[E_SPEC], v[E_SPEC + 1], v[E_SPEC + 2]);
      st3(a.out_albedo_rgb, v[E_ALB], v[E_ALB + 1], v[E_ALB + 2]);
      st3(a.out_occ, v[E_OCC], v[E_OCC], v[E_OCC]);
      st3(a.out_irradiance_rgb, v[E_IRR], v[E_IRR], v[E_IRR]);
      st3(a.out_n_dot_l_rgb, v[E_NDL], v[E_NDL], v[E_NDL]);
      st3(a.out_direct_diffuse_rgb, v[E_DD], v[E_DD + 1], v[E_DD + 2]);
      st3(a.out_direct_specular_rgb, v[E_DS], v[E_DS + 1], v[E_DS + 2]);
      st3(a.out_indirect_diffuse_rgb, v[E_ID], v[E_ID + 1], v[E_ID + 2]);
      st3(a.out_indirect_specular_rgb, v[E_IS], v[E_IS + 1], v[E_IS + 2]);
      st3(a.out_direct_rgb_viz, v[E_DVIZ], v[E_DVIZ + 1], v[E_DVIZ + 2]);
      st3(a.out_indirect_occ, v[E_W], v[E_W], v[E_W]);          // incoming_acc == 1 (surface_light_field.py:887,1067)
      st3(a.out_light_radiance_rgb, v[E_W], v[E_W], v[E_W]);    // light_radiance_mult == 1 (nerf.py:1118)
    }
  }
#ifdef RC_STAMPS
  if (lane == 0 && ray_ok && a.stamps) {
    unsigned long long* d = a.stamps + ray * 8;
    d[0] = st_begin; d[1] = st_mfma; d[2] = st_epi; d[3] = st_tiles_end; d[4] = __builtin_amdgcn_s_memtime();
    d[5] = st_kernel_begin; d[6] = st_t1; d[7] = st_t2;
  }
#endif
}

}  // namespace

void rc_launch_shadow_rays(const RcShadowRayArgs& a, hipStream_t stream) {
  if (a.n <= 0) return;
  hipLaunchKernelGGL(k_shadow_rays, dim3((unsigned)((a.n + 255) / 256)), dim3(256), 0, stream, a);
}

int rc_transient_shader_frags() { return TFrags::COUNT; }
int rc_transient_bins_frags() { return kBinFrags; }

void rc_launch_transient_shader(const RcTransShaderArgs& a, hipStream_t stream) {
  if (a.n <= 0) return;
  static std::atomic<uint64_t> prepared{0};
  const int lds = (kRingFloats + kWaves * kTAct * 64) * (int)sizeof(float);
  if (rc_first_use_on_device(prepared)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_transient_shader), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  }
  const int64_t ntiles = (a.n + 31) / 32;
  hipLaunchKernelGGL(k_transient_shader, dim3((unsigned)((ntiles + kWaves - 1) / kWaves)), dim3(kWaves * 64), lds, stream, a);
}

#ifdef RC_STAMPS
static unsigned long long* g_bins_stamps = nullptr;
extern "C" void* rc_debug_bins_stamps() { return g_bins_stamps; }
#endif

void rc_launch_transient_bins(const RcTransBinsArgs& a0, hipStream_t stream) {
  RcTransBinsArgs a = a0;
  if (a.n_rays <= 0) return;
#ifdef RC_STAMPS
  if (!g_bins_stamps) (void)hipMalloc((void**)&g_bins_stamps, (size_t)65536 * 8 * sizeof(unsigned long long));
  a.stamps = a.n_rays <= 65536 ? g_bins_stamps : nullptr;
#endif
  static std::atomic<uint64_t> prepared{0};
  const int lds = (kRingFloats + kWaves * kWaveLds) * (int)sizeof(float);
  if (rc_first_use_on_device(prepared)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_transient_bins), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  }
  hipLaunchKernelGGL(k_transient_bins, dim3((unsigned)((a.n_rays + kWaves - 1) / kWaves)), dim3(kWaves * 64), lds, stream, a);
}
