// Internal declarations shared by the gfx950 kernels and the C-ABI host (rc_api.hip).
#pragma once
#include <atomic>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rc_abi.h"

#define RC_WAVE 64
#define RC_MAX_GRID_LEVELS 8

// x / d.  Every BASELINE config divides by powers of two here (contraction radius 2, bounding box [-1, 1] -> extent 2):
// the product with the exact reciprocal 2^-e is then the same correctly rounded value as the quotient, at one
// instruction instead of the ~11 of an IEEE fp32 division.  `d` is wave-uniform (a kernel argument): one branch.
__device__ __forceinline__ float rc_div(float x, float d) {
  const uint32_t b = __float_as_uint(d);
  const bool pow2 = (b & 0x807FFFFFu) == 0u && b >= 0x01000000u && b <= 0x7E000000u;      // +2^e, e in [-125, 125]
  return pow2 ? x * __uint_as_float(0x7F000000u - b) : x / d;
}

// float32 constants the reference hard-codes (internal/math.py:24-26).
#define RC_TINY 1.17549435e-38f
#define RC_FMAX 3.40282347e+38f
#define RC_EPS 1.1920929e-07f

// math.safe_exp (internal/math.py:186-192): exp(jnp.clip(x, finfo.min, 70)).  jnp.clip = minimum(maximum(x, lo), hi) and
// both PROPAGATE a NaN (v_max_f32 / v_min_f32 return the other operand): selects on ordered compares keep the NaN, so a
// NaN raw density stays NaN as in the reference (oracle/JAX_CALLS.md, row `jnp.clip`).
__device__ __forceinline__ float rc_safe_exp(float x) {
  return expf(x > 70.0f ? 70.0f : (x < -RC_FMAX ? -RC_FMAX : x));
}

// ---------------------------------------------------------------------------------------------
// Hash grid
// ---------------------------------------------------------------------------------------------
struct RcGridLevel {
  const float* table;   // dense: [N,N,N,F] indexed [x,y,z]; hash: [T,F]
  int32_t size;         // N
  int32_t dense;        // 1: dense grid, 0: hash table
  uint32_t entries;     // N^3 or T
  uint32_t mask;        // T-1 if T is a power of two, else 0
  const float* cell;    // dense F = 1 levels of the proposal grids: cell table ((N+3)^3 cells x 8 corners, zero padding
                        // baked in; built with the fused kernel's tables) or nullptr
  const float* rec;     // hashed levels [kRcFusedDenseLevels, + kRcRecLevels) of the F = 1 proposal grids, [.., + kRcRec4Levels)
                        // of the F = 4 density grid: cell records ((N+1)^3 cell origins x the 8 hashed corner entries,
                        // rc_launch_build_hrec) or nullptr
};

struct RcGridDev {
  RcGridLevel lvl[RC_MAX_GRID_LEVELS];
  int32_t num_levels;
  int32_t num_features;
  float bbox;
  float precondition;
};

// points: world-space [n,3] (AoS) or SoA [3][n] (soa_in != 0).  Output feature-major [L*F][ldo]
// (feature_major != 0) or row-major [n][L*F].  contract_radius <= 0 disables the contraction.
// interleaved [grid A | grid B] tables of two F = 4 grids with the same level geometry, one per level (rc_api.hip
// build_fused_tables)
struct RcPairTables { const float* t[RC_MAX_GRID_LEVELS]; };
void rc_launch_hashgrid_pair(const RcGridDev& g, const RcPairTables& pt, const float* points_soa, const int32_t* src,
                             int64_t n_src, int64_t n, float* out_a, float* out_b, int64_t ldo, float contract_radius,
                             hipStream_t stream);
// two F = 4 grids looked up at the same row-major [n,3] points in one launch (row-major [n, L*4] outputs)
void rc_launch_hashgrid_two(const RcGridDev& ga, const RcGridDev& gb, const float* points, int64_t n, float* out_a, float* out_b,
                            float contract_radius, hipStream_t stream);
void rc_launch_hashgrid(const RcGridDev& g, const float* points, int soa_in, int64_t n, float* out,
                        int feature_major, int64_t ldo, float contract_radius, float* jac_out,
                        hipStream_t stream);

// ---------------------------------------------------------------------------------------------
// Sampling / compositing
// ---------------------------------------------------------------------------------------------
struct RcSampleArgs {
  // ray batch
  const float* origins; const float* directions; const float* viewdirs;
  const float* near; const float* far; const float* normals;
  int64_t n_rays;
  // previous level (P bins); level 0: prev_sdist == nullptr (sdist=[0,1], w=[1])
  const float* prev_sdist;    // [n, P+1]
  const float* prev_tdist;    // [n, P+1]
  const float* prev_density;  // [n*P]
  int32_t P;
  // outputs for the previous level
  float* prev_weights;        // [n*P] or nullptr
  // this level
  int32_t S;
  const float* jitter;        // [n] or nullptr
  float* sdist;               // [n, S+1]
  float* tdist;               // [n, S+1]
  float* means;               // SoA [3][n*S]
  // constants
  float anneal, padding;
  int32_t secondary;          // near replacement from the surface normal + far clamp (secondary rays)
  int32_t use_raydist;        // sample in power-ladder distance (Model.get_bg_and_raydist, models.py:183-191)
  float raydist_p, raydist_premult, eps_dot_min, far_clamp;
  // use_raydist with ONE (near, far) for the whole batch (the material stage's secondary rays): power_ladder(near),
  // power_ladder(far) computed once on the device (rc_launch_ladder_bounds) instead of by every wave; nullptr otherwise
  const float* s_bounds;
};
void rc_launch_sample(const RcSampleArgs& a, hipStream_t stream);

// Stand-alone stepfun.sample_intervals (parity tests): t [n,P+1], logits [n,P] -> out [n,S+1].
void rc_launch_sample_intervals(const float* t, const float* logits, int64_t n, int P, int S,
                                const float* jitter, float* out, hipStream_t stream);

struct RcCompositeArgs {
  const float* directions; const float* origins; const float* lights;
  int64_t n_rays;
  int32_t S;                  // samples of the last level
  const float* tdist;         // [n, S+1]
  const float* density;       // [n*S]
  const float* means;         // SoA [3][n*S]
  const float* normals_pred;  // SoA [3][n*S]
  const float* normals_grad;  // SoA [3][n*S] or nullptr
  const float* shade;         // SoA [RC_SHADE_CH][n*Sf] per shaded sample
  int32_t Sf;                 // shaded samples per ray: S (no resampling) or 1
  const int32_t* inds;        // [n] selected sample when Sf == 1
  const float* filt_weight;   // [n] importance weight w/(n*p+1e-8) when Sf == 1
  float* weights;             // [n*S] out
  float bg;
  float pct[3];
  rc_outputs out;
};
void rc_launch_composite(const RcCompositeArgs& a, hipStream_t stream);

// Categorical resampling of the last level (models.py:193-292), num_resample == 1.
struct RcResampleArgs {
  int64_t n_rays; int32_t S;
  const float* tdist; const float* density; const float* directions;
  const float* gumbel;        // [n,S] or nullptr (then inds_in must be given)
  const int32_t* inds_in;     // [n] or nullptr
  int32_t* inds_out;          // [n]
  float* filt_weight;         // [n]
  float* weights;             // [n*S] out (weights_no_filter)
  float* acc_out;             // [n] or nullptr: sum of the weights, added as k_composite adds it (rc_launch_composite_pick)
  int32_t* src_out;           // [n] or nullptr: flat index ray * S + pick of the picked sample (per-pick lookups read through it)
  // material stage: position and predicted normal of the picked sample, gathered here (all four or none)
  const float* means;         // SoA [3][n*S]
  const float* normals;       // SoA [3][n*S]
  float* pts_out;             // [n,3]
  float* nrm_out;             // [n,3]
};
void rc_launch_resample(const RcResampleArgs& a, hipStream_t stream);
// k_composite for the case "one resampled sample per ray, only rgb and acc wanted" (the batched secondary trace): the
// weights and their sum are k_resample's, the only non-zero filtered weight sits on the pick -- one thread per ray.
// shade_rgb: the three colour channels of the shader's output, [3][n].  Same values, bit for bit.
void rc_launch_composite_pick(int64_t n, const float* shade_rgb, const float* filt_weight, const float* acc, float bg,
                              float* out_rgb, float* out_acc, hipStream_t stream);
// out[0] = power_ladder(near'), out[1] = power_ladder(far') with near' / far' as sample_level_ray derives them from
// (near, far) for a secondary ray without a surface normal
void rc_launch_ladder_bounds(float near, float far, float far_clamp, float p, float premult, float* out, hipStream_t stream);

// ---------------------------------------------------------------------------------------------
// MFMA MLP kernels
// ---------------------------------------------------------------------------------------------
// Per-sample shader outputs (SoA channel-major).
enum { RC_SH_RGB = 0, RC_SH_AD = 3, RC_SH_ID = 6, RC_SH_IS = 9, RC_SH_TINT = 12, RC_SHADE_CH = 15 };

struct RcDensityMlpArgs {
  const float* feat;          // feature-major [K][ld]
  int64_t n; int64_t ld;
  int32_t K;                  // 6, 7 or 32
  const float* wstream;       // packed MFMA fragment stream [d0 | d1 | out]
  const float* means;         // SoA [3][n] (validity mask); with `src`: SoA [3][n_src], point p reads column src[p]
  const int32_t* src; int64_t n_src;
  float density_bias, contract_radius, bbox;
  int32_t last;               // 1: also write hidden feature + predicted normals
  float* density;             // [n]
  float* hbuf;                // [n/32][32 steps][64] hidden feature in accumulator layout
  float* normals_pred;        // SoA [3][n]
  const float* jac;           // [3][K][ld] d feature / d contracted coordinate (k_hashgrid_fwd<F, true>) or nullptr
  float* normals_grad;        // SoA [3][n] analytic normals (needs jac; the stream must carry the backward fragments)
};
void rc_launch_density_mlp(const RcDensityMlpArgs& a, hipStream_t stream);

// One proposal level as one launch (rc_level.hip): grid lookup + density MLP, density only.
struct RcLevelArgs {
  const RcGridDev* grid;
  const float* means;         // SoA [3][n]
  int64_t n;
  const float* wstream;       // the level's "dens_<l>" fragment stream
  float density_bias, contract_radius;
  float* density;             // [n]
  int cu_reserve = 0;         // CUs the launch leaves to a kernel released beside it (k_level_ray only)
};
bool rc_level_supported(const RcGridDev& g);
void rc_launch_level(const RcLevelArgs& a, hipStream_t stream);
// the level's sampling (rc_launch_sample) and the level itself as ONE launch, one ray per wave
bool rc_level_ray_supported(const RcGridDev& g, int S);
void rc_launch_level_ray(const RcLevelArgs& a, const RcSampleArgs& sa, hipStream_t stream);

struct RcShaderArgs {
  int64_t n;                  // shaded points
  int64_t n_src;              // points of the last level (stride of the SoA inputs)
  const int32_t* src;         // [n] source point index or nullptr (identity)
  int32_t samples_per_ray;    // shaded samples per ray (ray = point / samples_per_ray)
  const float* hbuf;          // hidden density feature, accumulator layout, indexed by source point
  const float* app;           // appearance features, feature-major [32][n] (already gathered per shaded point)
  const float* normals_pred;  // SoA [3][n_src]
  const float* viewdirs;      // [n_rays,3]
  const float* wstream;       // packed MFMA fragment stream [heads | s0 | i0 | i1 | io | s1 | s2 | sb | so]
  const float* ide_coef;      // IDE polynomial table (device)
  float roughness_bias, irradiance_bias, ambient_bias, rgb_max, slf_ambient_bias;
  float* shade;               // SoA [RC_SHADE_CH][n]
  void* debug;                // diagnostic builds (-DRC_STAMPS): per-tile cycle stamps
};
void rc_launch_shader(const RcShaderArgs& a, hipStream_t stream);

// Model-level EnvMap MLP on ray directions (secondary-ray background).
struct RcEnvMapArgs {
  int64_t n; const float* viewdirs;   // [n,3]
  const float* wstream;               // packed fragments [e0 | e1 | e2 | eb(x) | eb(inputs) | out]
  float rgb_bias; float* env_rgb;     // [n,3]
};
void rc_launch_envmap(const RcEnvMapArgs& a, hipStream_t stream);

// IDE table layout (built on the host, see rc_api.hip): for deg_view 5 there are 36 (l,m)
// terms; term i has polynomial coefficients in z of degree <= 16 and an xy power m.
#include "rc_pack_host.h"      // RC_IDE_TERMS, RC_IDE_ZPOW, RcIdeTable, rc_cell_corner (host-only header)

// ---------------------------------------------------------------------------------------------
// Material stage (rc_material.hip)
// ---------------------------------------------------------------------------------------------
enum { RC_MAT_CH = 5 };   // per point: albedo rgb, roughness, metalness
enum { RC_VMF_CH = 5 };   // per lobe: normalised mean xyz, kappa, softmax weight
enum { RC_SMP_CH = 5 };   // per secondary sample: local light dir xyz, pdf, MIS weight

struct RcMatHeadArgs {
  int64_t n; const float* feat;        // row-major [n,32] material-grid features
  const float* w0; const float* b0;    // Flax kernel [32,128], bias
  const float* w1; const float* b1;    // [128,10]
  float min_roughness; float* mat;     // [n, RC_MAT_CH]
};
void rc_launch_material_head(const RcMatHeadArgs& a, hipStream_t st);
void rc_launch_material_composite_all(int64_t n, int S, const float* weights, const float* mat, float* out_albedo,
                                      float* out_rough, float* out_metal, float* out_f0, float f0, hipStream_t st);

struct RcLightHeadArgs {
  int64_t n; const float* feat;        // [n,32] light-grid features
  const float* w0; const float* b0; const float* w1; const float* b1; const float* w2; const float* b2;
  const float* pts; const float* noise;   // [n,3], [n,128,3]
  float vmf_scale; float* vmf;         // [n,128,RC_VMF_CH]
  float* vmf_logit;                    // [n,128]: the lobe logits as the softmax sees them (the categorical lobe draw takes these)
};
void rc_launch_light_head(const RcLightHeadArgs& a, hipStream_t st);
void rc_launch_shading_heads(const RcMatHeadArgs& m, const RcLightHeadArgs& l, hipStream_t st);     // both in one launch

struct RcBrdfSampleArgs {
  int64_t n; int32_t Ks, Kd, Kc;
  const float* pts; const float* nrm; const float* viewdirs; const float* lights; const float* mat; const float* vmf;
  const float* spec_u1; const float* spec_u2; const float* cos_u1; const float* cos_u2;
  const int32_t* vmf_lobe; const float* vmf_v; const float* vmf_tmp;
  const float* vmf_lobe_gumbel;   // [n,128] or nullptr: lobe = argmax(logit + gumbel) when vmf_lobe is nullptr
  const float* vmf_logit;         // [n,128] lobe logits of the light head (read only for that draw)
  float normal_eps, near, far;
  float* sec_origins; float* sec_dirs; float* sec_near; float* sec_far; float* sec_lights;   // [n*(Ks+Kd), .]
  float* samples;       // [n, Ks+Kd, RC_SMP_CH]
  float* local_view;    // [n,3]
};
void rc_launch_brdf_sample(const RcBrdfSampleArgs& a, hipStream_t st);

struct RcMatIntegrateArgs {
  int64_t n; int32_t Ks, Kd, S;
  const float* mat; const float* samples; const float* local_view;
  const float* sec_rgb; const float* sec_acc; const float* sec_env;
  const float* weights;       // [n,S] unfiltered weights of the primary rays
  const float* filt_weight;   // [n]
  const float* pts; const float* nrm; const float* origins; const float* lights;
  float f0, rgb_max, bg;
  rc_mat_outputs out;
};
void rc_launch_material_integrate(const RcMatIntegrateArgs& a, hipStream_t st);

// ---------------------------------------------------------------------------------------------
// Fused cache forward (rc_fused.hip): one launch per batch of primary rays
// ---------------------------------------------------------------------------------------------
struct RcFusedLaunch {
  rc_rays rays; int64_t n;
  const float* jitter[3]; int32_t num_samples[3];
  const RcGridDev* grid[4];            // proposal 0, 1, 2 + appearance
  const float* pair_table[RC_MAX_GRID_LEVELS];   // level-2 density and appearance tables interleaved entry by entry
                                                 // (dense levels: cell tables of interleaved pairs)
  const float* cell_table[2][RC_MAX_GRID_LEVELS]; // dense levels of proposal grids 0 / 1 as cell tables (else NULL)
  const float* wstream;                // [density MLP 0 | 1 | 2 (+ backward) | shader], see rc_fused_stream_offsets
  const float* ide_coef;
  float anneal, padding, density_bias, contract_radius, bg; float pct[3];
  float roughness_bias, irradiance_bias, ambient_bias, rgb_max, slf_ambient_bias;
  rc_outputs out;
  // front end only (time-resolved cache): stop behind the last proposal level, results into the workspace buffers of
  // the launch-per-stage plan (wstream then ends at the shader's offset)
  int32_t front, want_grad;
  int32_t export_samples;              // full kernel + f_tdist / f_density / f_means / f_normals_pred of the last level
  int32_t direct;                      // weight fragments straight from global memory (no LDS ring, no workgroup barriers)
  int32_t team;                        // two wavefronts per ray, two workgroups per CU (rc_fused2.hip); plain pass only
  int32_t stagger_cycles;              // (team, experiment) late start of the second half of the grid
  int32_t prio_mode;                   // (team) priority scheme of the younger workgroup of a CU
  int32_t use_raydist; float raydist_p, raydist_premult;
  float* f_tdist; float* f_density; float* f_means; float* f_normals_pred; float* f_normals_grad; float* f_hbuf; float* f_app;
};
// the fused kernels are compiled for this level layout of every grid: levels [0, kRcFusedDenseLevels) dense (16, 32, 64
// cells a side against 2^19 entries), the others hashed
constexpr int kRcFusedDenseLevels = 3;
// ... and the first kRcRecLevels hashed levels of the F = 1 grids are read through cell RECORDS: one 32-byte record per
// cell origin with the 8 hashed corner values side by side, so a lookup is ONE sector instead of four (rc_dev_grid.h
// kLevelHRec).  69 MB per grid for the 128^3 level, 543 MB for 256^3, 4.3 GB for 512^3.  Measured (same box, builds with
// 0 | 1 | 2 record levels, profiles/r04_ab_cell_records.txt): fused kernel 124.8 | 124.3 | 123.5 us per 1024 rays, 1720 |
// 1707 | 1716 us per 16 384, material stage 1.437 | 1.414 | 1.422 ms -- one level is worth 0.5-1.7 %, the second nothing:
// four x-pair sectors out of a 2 MiB table (L2) cost what one sector out of a table behind the L2 costs
// (profiles/r04_gather_cell_records.txt: 65 G lookups/s against 54-59).  One level it is.
#ifndef RC_REC_LEVELS
#define RC_REC_LEVELS 1
#endif
constexpr int kRcRecLevels = RC_REC_LEVELS;
// The same for the F = 4 density grid of the last level, which the level kernels read on the lean pass (density of all
// 32 samples of a secondary ray before one is picked): its hashed tables are 8 MiB each, 40 MiB together -- every sector
// is a fabric read there (k_level_ray<4, 8, 32>: 22 M L2 misses per 32 768-ray trace, the memory system's random-sector
// rate), and a 128-byte record (8 corners x 16 bytes = one cache line) replaces ~4.4 sectors by 2.
// 275 MB for the 128^3 level, 2.2 GB for 256^3.
#ifndef RC_REC4_LEVELS
#define RC_REC4_LEVELS 2
#endif
constexpr int kRcRec4Levels = RC_REC4_LEVELS;
// dst[(N+1)^3][8]: record (qx, qy, qz) in [0, N]^3 = cell origin (qx - 1, qy - 1, qz - 1), corner c = 4 b0 + 2 b1 + b2 at
// origin + (b0, b1, b2) through the level's hash (power-of-two table)
void rc_launch_build_hrec(const float* table, int N, uint32_t mask, int F, float* dst, hipStream_t stream);
int rc_fused_stream_offsets(int* l0, int* l1, int* l2, int* sh);   // returns the total fragment count
void rc_launch_fused(const RcFusedLaunch& L, hipStream_t stream);
// dst[cell][corner][dst_stride floats, written F at dst_off]: the 8 corners of every cell of the zero-padded dense
// level `src` [N^3][F] (cells (N + 3)^3, corner order b0 b1 b2 as in grid_combine)
void rc_launch_build_cells(const float* src, int N, int F, float* dst, int dst_stride, int dst_off, hipStream_t stream);


// ---------------------------------------------------------------------------------------------
// Time-resolved cache (rc_transient.hip)
// ---------------------------------------------------------------------------------------------
// per-sample channels written by k_transient_shader ([RC_TS_COUNT][n], channel-major)
enum { RC_TS_DD = 0, RC_TS_DS = 3, RC_TS_ALBEDO = 6, RC_TS_TIB = 9, RC_TS_ROUGH = 12, RC_TS_NDOTL = 13, RC_TS_IRRAD = 14,
       RC_TS_OCC = 15, RC_TS_LDIST = 16, RC_TS_RDIST = 17, RC_TS_CAMDIST = 18, RC_TS_COUNT = 19 };

struct RcTransShaderArgs {
  int64_t n; int32_t samples_per_ray;
  const float* hbuf; const float* app; const float* means; const float* normals;   // means / normals: SoA [3][n]
  const float* origins; const float* viewdirs; const float* lights; const float* cam_origins;   // per ray [.,3]
  const float* occ;                        // optional: acc of the shadow ray of each sample [n] (nerf.py:1300-1340)
  float occ_threshold;
  const float* wstream; const float* ide_coef;
  float roughness_bias, albedo_bias, brdf_bias, rgb_max, contract_radius;
  float light_power, light_near; int32_t use_falloff, light_zero;
  float* irr_feat;                         // [tiles][32 steps][64 lanes]
  float* slf_feat;                         // [tiles][64 steps][64 lanes]
  float* tshade;                           // [RC_TS_COUNT][n]
};

struct RcTransBinsArgs {
  int64_t n_rays;
  unsigned long long* stamps;              // diagnostic builds (-DRC_STAMPS) only
  const float* wstream;                    // per column tile: 65 SLF output fragments | 33 transient_indirect fragments
  const float* slf_feat; const float* irr_feat; const float* tshade; const float* weights;   // weights [n_rays][32]
  float exposure, shift, max_dists, irradiance_bias, slf_rgb_bias, indirect_scale, rgb_max, light_near;
  int32_t bin_zero_threshold_light, light_zero, n_taps;
  const float* taps;                       // temporal filter (device), n_taps entries
  float* out_rgb; float* out_direct; float* out_indirect;               // [n_rays][700][3]
  float* out_ti_diffuse; float* out_ti_specular;                         // unshifted composites [n_rays][700][3]
  float* out_direct_rgb; float* out_indirect_rgb; float* out_integrated_rgb;
  float* out_diffuse_rgb; float* out_specular_rgb; float* out_albedo_rgb; float* out_occ; float* out_indirect_occ;
  float* out_irradiance_rgb; float* out_light_radiance_rgb; float* out_n_dot_l_rgb; float* out_direct_diffuse_rgb;
  float* out_direct_specular_rgb; float* out_indirect_diffuse_rgb; float* out_indirect_specular_rgb; float* out_direct_rgb_viz;
};
struct RcShadowRayArgs {
  int64_t n; int32_t samples_per_ray;      // n = shaded samples
  const float* means; const float* normals;            // SoA [3][n]
  const float* lights;                                 // per primary ray [.,3]
  float normal_eps, shadow_near, shadow_far, light_near;
  float* origins; float* dirs; float* near; float* far; float* out_normals; float* out_lights;   // AoS [n,3] / [n]
};
void rc_launch_shadow_rays(const RcShadowRayArgs& a, hipStream_t stream);
int rc_transient_shader_frags();
int rc_transient_bins_frags();
void rc_launch_transient_shader(const RcTransShaderArgs& a, hipStream_t stream);
void rc_launch_transient_bins(const RcTransBinsArgs& a, hipStream_t stream);


// ---------------------------------------------------------------------------------------------
// On-device ray generation (rc_camera.hip)
// ---------------------------------------------------------------------------------------------
struct RcCastArgs {
  int64_t n;
  const int32_t* pix_x; const int32_t* pix_y;      // explicit pixel batch, or NULL: the rectangle below, row-major
  int32_t x0, y0, width;
  float pixtocam[9]; float rot[9]; float trans[3]; float light[3];
  float near_v, far_v;
  int32_t camtype;                                  // 0 perspective, 1 panoramic, 2 fisheye, 3 fisheye (equisolid)
  int32_t has_distortion; float dist[6];            // k1 k2 k3 k4 p1 p2
  int32_t has_ndc; float ndc_xmult, ndc_ymult;      // 1 / pixtocam_ndc[0][2], 1 / pixtocam_ndc[1][2]
  int32_t has_z_range; float z_lo, z_hi;            // cast_ray_batch's z_range
  const float* pix_dx; const float* pix_dy;         // sub-pixel jitter offsets [n] or NULL
  float* origins; float* directions; float* viewdirs; float* radii; float* imageplane; float* look; float* up;
  float* lights; float* near; float* far;
};
void rc_launch_cast_rays(const RcCastArgs& a, hipStream_t stream);

// Training backward of one level's density field (rc_train.hip)
struct RcDensityBwdArgs {
  const float* feat;          // grid features, feature-major [K][ld] (k_hashgrid_fwd)
  int64_t n; int64_t ld;
  int32_t K;
  const float* wstream;       // [d0 | d1 | out | W1^T | W0^T] fragments
  const float* points;        // world-space [n,3]
  float density_bias, contract_radius, bbox;
  const float* d_density;     // [n] upstream
  const float* d_feature;     // [n,64] upstream or nullptr
  float* density;             // [n]
  float* graw;                // [n] d L / d raw density
  float* a1; float* a2; float* d2; float* d1;   // point-major [n,64]
  float* fe;                  // point-major [n,32] staged grid features
  float* dfeat;               // feature-major [K][ld] d L / d grid feature
};
void rc_launch_density_bwd(const RcDensityBwdArgs& a, hipStream_t stream);
struct RcWgradArgs {
  const float* a1; const float* d2; const float* fe; const float* d1; const float* a2; const float* graw;
  int64_t n; int32_t K; int64_t steps_per_wave;
  float* partial;             // [rc_wgrad_waves(n) / 4][partial stride]: one per workgroup
};
int rc_wgrad_waves(int64_t n);
int rc_wgrad_partial_floats(int nwaves);
// grads: [W0 K x 64 | b0 64 | W1 64 x 64 | b1 64 | Wout 64 | bout 1], accumulated into
void rc_launch_wgrad(RcWgradArgs a, int K, float* grads, hipStream_t stream);
struct RcGridScatterArgs {
  RcGridDev grid;             // geometry of the forward tables
  float* gtable[RC_MAX_GRID_LEVELS];   // gradient tables, same layout as the forward tables
  const float* points;        // world-space [n,3]
  int64_t n; int64_t ld;
  const float* dfeat;         // feature-major [L*F][ld], or point-major [n][L*F] (point_major != 0)
  float contract_radius;
  int32_t point_major;
  int32_t level0;             // first level of this launch (blockIdx.y = level - level0)
  uint32_t lds_levels;        // bit l: level l is summed through LDS by k_grid_scatter_small
};
// small_stream (optional): where the LDS-accumulated levels (k_grid_scatter_small) go; they are independent of the
// other levels' scatter and of each other (disjoint tables) -- the caller orders both streams
void rc_launch_grid_scatter(const RcGridScatterArgs& a, hipStream_t stream, hipStream_t small_stream = nullptr, bool use_small_stream = false);
bool rc_train_prepare();     // LDS opt-in of the scatter kernels (call once outside any stream capture)

// Random fill (rc_prng.hip)
enum { RC_PRNG_BITS = 0, RC_PRNG_UNIFORM = 1, RC_PRNG_NORMAL = 2, RC_PRNG_GUMBEL = 3 };
struct RcPrngArgs {
  uint32_t key0, key1;
  int mode;
  float lo, hi;
  int64_t n;
  uint32_t* out;
};
void rc_launch_prng_fill(const RcPrngArgs& a, hipStream_t stream);


// Kernel attributes (dynamic-LDS limit) are per device: true the first time the calling thread's current device
// shows up for this `mask` (one mask per kernel), so a process driving several GPUs sets them on each.
// CU count of the calling thread's current device, asked once per device (it sits on the launch path of the persistent
// level kernels).
inline int rc_device_cus() {
  static std::atomic<int> cached[64];
  int dev = 0;
  (void)hipGetDevice(&dev);
  int v = cached[dev & 63].load(std::memory_order_relaxed);
  if (v <= 0) {
    v = 256;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
    cached[dev & 63].store(v, std::memory_order_relaxed);
  }
  return v;
}

inline bool rc_first_use_on_device(std::atomic<uint64_t>& mask) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  const uint64_t bit = 1ull << (dev & 63);
  return (mask.fetch_or(bit) & bit) == 0;
}
