/*
 * rc_abi.h -- C ABI of the MI355X radiance-cache ray-batch renderer.
 *
 * The reference (benattal/neural-radiance-caching) has no FFI: the hot path sits
 * behind three Python call signatures.  These entry points are what
 * a host binding for that path would bind; each one cites the reference interface it
 * replaces (file:line relative to the reference tree).  Plain C types only: no torch,
 * no HIP types (a stream is passed as `void*` holding a hipStream_t).
 *
 * Conventions
 *   - every function returns 0 on success or a negative rc_status; the message is
 *     available through rc_last_error(); nothing aborts or throws across the ABI;
 *   - all ray / random / output buffers are DEVICE pointers owned by the caller
 *     (float32, row-major, leading dimension = rays); the handle owns its weight
 *     copies and its workspace; after the first call at a given n_rays no allocation
 *     happens inside rc_render_rays;
 *   - work is enqueued asynchronously on the caller's stream (mirrors pmap's async
 *     dispatch, internal/train_utils.py:3821-3830); synchronisation is the caller's;
 *   - a handle is bound to one device and is not thread-safe.
 */
#ifndef RC_ABI_H_
#define RC_ABI_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RC_ABI_VERSION 4
#define RC_MAX_LEVELS 3

typedef struct rc_handle rc_handle;

typedef enum {
  RC_OK = 0,
  RC_ERR_INVALID_ARG = -1,
  RC_ERR_HIP = -2,
  RC_ERR_MISSING_WEIGHT = -3,
  RC_ERR_SHAPE = -4,
  RC_ERR_UNSUPPORTED = -5,
  RC_ERR_NO_DEVICE = -6,
  RC_ERR_HOST = -7             /* a C++ exception (e.g. std::bad_alloc) caught at the boundary */
} rc_status;

/* One multiresolution dense+hash encoding.
 * Replaces HashEncoding's constructor fields (internal/grid_utils.py:739-805). */
typedef struct {
  int32_t hash_map_size;       /* T */
  int32_t max_grid_size;       /* N_max */
  int32_t min_grid_size;       /* N_min */
  int32_t num_features;        /* F: 1 or 4 */
  float bbox;                  /* bbox_scaling: cube [-bbox, bbox]^3 */
  float precondition_scaling;  /* x10 */
} rc_grid_config;

/* Resolved render-time configuration (the values the reference takes from its gin
 * chain; field provenance is listed in neural-radiance-caching_amd/config.py). */
typedef struct {
  uint32_t abi_version;        /* must be RC_ABI_VERSION */
  int32_t num_levels;          /* proposal rounds (3) */
  int32_t num_samples[RC_MAX_LEVELS];      /* (64, 64, 32): sampling_strategy, internal/sampling.py:53 */
  rc_grid_config proposal_grids[RC_MAX_LEVELS];
  rc_grid_config appearance_grid;
  rc_grid_config material_grid;
  rc_grid_config light_grid;
  float anneal;                /* sampling.py:326-339 */
  float resample_padding;
  float raydist_p;             /* power_ladder p (secondary rays) */
  float raydist_premult;
  float shadow_normal_eps_dot_min;
  float density_bias;          /* geometry.py:320 */
  float contract_radius;       /* coord.contract_radius_2 */
  float roughness_bias;
  float irradiance_bias;
  float ambient_irradiance_bias;
  float rgb_max;
  float slf_ambient_bias;
  float env_rgb_bias;
  float env_map_distance;
  float bg_intensity;          /* VolumeIntegrator.bg_intensity_range (equal ends) */
  float percentiles[3];        /* (5, 50, 95) */
  int32_t num_resample;        /* 1 */
  /* material stage */
  float diffuse_sample_fraction; /* MaterialMLP.diffuse_sample_fraction (0.5) */
  float secondary_normal_eps;    /* Config.secondary_normal_eps (1e-2) */
  float secondary_near;          /* MaterialMLP.near_min (5e-2) */
  float secondary_far;           /* Config.secondary_far (2) */
  float min_roughness;           /* 0.01 */
  float default_F_0;             /* 0.04 */
  float vmf_scale;               /* LightMLP.vmf_scale (20) */
  int32_t num_vmf;               /* LightMLP.num_components (128) */
} rc_config;

/* One named parameter tensor.  `name` is the Flax tree path
 * ("params/Cache/Sampler/MLP_0/density_grid/grid_016", ".../density_layers_0/kernel", ...),
 * i.e. what flax.training.checkpoints stores (internal/train_utils.py:4035-4088).
 * Dense kernels are [in, out]. */
typedef struct {
  const char* name;
  const void* data;            /* float32, contiguous */
  int32_t ndim;
  int64_t shape[4];
  int32_t on_device;           /* 0: host pointer, 1: device pointer */
} rc_tensor_desc;

/* Ray batch: the fields of utils.Rays (internal/utils.py:142-169) that the path reads. */
typedef struct {
  const float* origins;        /* [n,3] */
  const float* directions;     /* [n,3] */
  const float* viewdirs;       /* [n,3] */
  const float* near;           /* [n]   */
  const float* far;            /* [n]   */
  const float* lights;         /* [n,3] (light_dists extra) */
  const float* normals;        /* [n,3] or NULL; secondary rays only (sampling.py:182-205) */
} rc_rays;

/* Explicit random inputs standing in for jax.random (threefry is not reproduced).
 * A NULL rc_randoms*, or a NULL member, selects the reference's rng=None branch. */
typedef struct {
  const float* jitter[RC_MAX_LEVELS]; /* [n] U[0,1) per level: stepfun.sample single_jitter (stepfun.py:197-202) */
  const float* gumbel;                /* [n, S_last] standard Gumbel: jax.random.categorical (models.py:242-247) */
  const int32_t* resample_inds;       /* [n] optional: overrides the categorical draw (filtered_sampler_inds) */
} rc_randoms;

/* pass_mask bits */
#define RC_PASS_CACHE      0x1u   /* cache-only forward (BaseNeRFModel.__call__, internal/models.py:657-774) */
#define RC_PASS_SECONDARY  0x2u   /* is_secondary=True: far clamp, power-ladder distances, bg 0, resample, EnvMap */
#define RC_PASS_RESAMPLE   0x4u   /* force categorical resampling to num_resample samples (models.py:193-292) */
#define RC_PASS_NO_ENVMAP  0x8u   /* use_env_map=False for secondary rays (material.py:2191-2217) */

/* Output slots: keys of the reference's `render` dict (integrator results,
 * internal/render.py:172-247, internal/integration.py:199-231, internal/models.py:2087-2158).
 * The `cache_<k>` keys of _finalize_outputs are aliases of these and are produced by the
 * host layer, not by extra device buffers. */
typedef enum {
  RC_OUT_RGB = 0,               /* [n,3] */
  RC_OUT_ACC,                   /* [n]   */
  RC_OUT_DISTANCE_MEAN,         /* [n]   */
  RC_OUT_DISTANCE_PERCENTILE_5, /* [n]   */
  RC_OUT_DISTANCE_MEDIAN,       /* [n]   */
  RC_OUT_DISTANCE_PERCENTILE_95,/* [n]   */
  RC_OUT_DIFFUSE_RGB,           /* [n,3] */
  RC_OUT_SPECULAR_RGB,
  RC_OUT_DIRECT_RGB,            /* == ambient_rgb == direct_diffuse_rgb == ambient_diffuse_rgb (+ exact 0) */
  RC_OUT_INDIRECT_RGB,
  RC_OUT_ALBEDO_RGB,
  RC_OUT_INDIRECT_DIFFUSE_RGB,
  RC_OUT_INDIRECT_SPECULAR_RGB,
  RC_OUT_INDIRECT_OCC,          /* [n,3] */
  RC_OUT_MEANS,                 /* [n,3] */
  RC_OUT_NORMALS,               /* [n,3] analytic (density gradient) */
  RC_OUT_NORMALS_PRED,          /* [n,3] == normals_to_use */
  RC_OUT_RAY_DISTS,             /* [n]   */
  RC_OUT_LIGHT_DISTS,           /* [n]   */
  RC_OUT_ENV_MAP_RGB,           /* [n,3] secondary rays only */
  RC_OUT_RGB_NO_ENV,            /* [n,3] secondary: rgb before the EnvMap composite (rgb_no_stopgrad - env) */
  RC_OUT_COUNT
} rc_output_id;

typedef struct {
  float* ptr[RC_OUT_COUNT];    /* device pointers; NULL = not requested */
} rc_outputs;

/* ---- material stage (config 3): BaseMaterialModel.__call__ with use_material / use_light_sampler
 * (internal/models.py:1144-1254, 1398-1694).  Explicit random inputs for everything the reference
 * draws with jax.random on that path; Ks = round(K (1 - diffuse_fraction)), Kd = K - Ks, Kc = Kd / 2. */
typedef struct {
  const float* gumbel;            /* [n, S_last]  categorical pick of the shading sample (models.py:1418-1438) */
  const float* vmf_noise;         /* [n, 128, 3]  N(0,1): LightMLP.get_vmfs means_random (light_sampler.py:142-144) */
  const float* spec_u1;           /* [n, Ks]      RandomGenerator2D.sample for the GGX sampler (render_utils.py:322-352) */
  const float* spec_u2;           /* [n, Ks] */
  const float* cos_u1;            /* [n, Kc]      ... for the cosine sampler */
  const float* cos_u2;            /* [n, Kc] */
  const int32_t* vmf_lobe;        /* [n]          categorical lobe pick of sample_vmf_vars (render_utils.py:1360-1372) */
  const float* vmf_v;             /* [n, Kd-Kc, 2] N(0,1) (render_utils.py:1409-1410) */
  const float* vmf_tmp;           /* [n, Kd-Kc]   U[0,1) (render_utils.py:1413) */
  const float* sec_jitter[RC_MAX_LEVELS];  /* [n*(Ks+Kd)] per level: jitter of the secondary rays, block [n*Ks | n*Kd] */
  const float* sec_gumbel;        /* [n*(Ks+Kd), S_last] */
  /* optional (NULL: draw from the Gumbel noise above): the categorical picks themselves, as rc_randoms.resample_inds
   * does for rc_render_rays -- filtered_sampler_inds of the primary rays (models.py:1418-1438) and of the batched
   * secondary trace (material.py:2191-2217 -> models.py:193-292).  With both given the noise members may be NULL. */
  const int32_t* resample_inds;     /* [n]          */
  const int32_t* sec_resample_inds; /* [n*(Ks+Kd)]  block [n*Ks | n*Kd] like sec_jitter */
  /* optional, instead of vmf_lobe (then NULL): standard Gumbel noise [n, 128]; the lobe is drawn on the device as
   * argmax_j(logit_j + g_j), which is jax.random.categorical(key, logits) of sample_vmf_vars (render_utils.py:1357-1372)
   * when g = jax.random.gumbel(key, [n, 128]) */
  const float* vmf_lobe_gumbel;
} rc_material_randoms;

typedef enum {
  RC_MOUT_RGB = 0,                 /* [n,3] material_rgb */
  RC_MOUT_ACC,                     /* [n]   */
  RC_MOUT_DIRECT_RGB, RC_MOUT_INDIRECT_RGB, RC_MOUT_DIFFUSE_RGB, RC_MOUT_SPECULAR_RGB,
  RC_MOUT_DIRECT_DIFFUSE_RGB, RC_MOUT_DIRECT_SPECULAR_RGB, RC_MOUT_INDIRECT_DIFFUSE_RGB,
  RC_MOUT_INDIRECT_SPECULAR_RGB,   /* [n,3] each */
  RC_MOUT_INDIRECT_OCC,            /* [n]   */
  RC_MOUT_LIGHTING_IRRADIANCE,     /* [n,3] */
  RC_MOUT_MATERIAL_ALBEDO,         /* [n,3] composited over all samples (models.py:1845-1912) */
  RC_MOUT_MATERIAL_ROUGHNESS,      /* [n]   */
  RC_MOUT_MATERIAL_METALNESS,      /* [n]   */
  RC_MOUT_MATERIAL_F_0,            /* [n]   */
  RC_MOUT_MEANS,                   /* [n,3] of the filtered sample */
  RC_MOUT_NORMALS_TO_USE,          /* [n,3] */
  RC_MOUT_RAY_DISTS,               /* [n]   */
  RC_MOUT_LIGHT_DISTS,             /* [n]   */
  RC_MOUT_COUNT
} rc_mat_output_id;

typedef struct {
  float* ptr[RC_MOUT_COUNT];
} rc_mat_outputs;

/* -- lifecycle: replaces models.construct_model / model.init (internal/models.py:2323-2358) */
int rc_create(const rc_config* cfg, int device, rc_handle** out);
void rc_destroy(rc_handle* h);
const char* rc_last_error(const rc_handle* h);   /* h may be NULL: last error of rc_create */
int rc_abi_version(void);
/* Arithmetic of the shader / EnvMap MLP layers this library was built with (csrc/rc_pack_host.h RC_SPLIT_MFMA):
 * 0 = fp32 MFMA (v_mfma_f32_32x32x2_f32, the exact fp32 chain); 1 = every fp32 operand split exactly into three bf16
 * pieces, six products per 16 k on v_mfma_f32_32x32x16_bf16, fp32 accumulation (same error against fp64 as the fp32 chain:
 * tests/test_gpu_parity.py, the noise-floor test).  The density MLPs of the proposal levels are fp32 MFMA in both. */
int rc_mlp_arithmetic(void);

/* -- weights: replaces flax `variables` passed to model.apply (internal/train_utils.py:3796-3814)
 * May be called several times; tensors with unknown names are rejected.
 * Ordering contract: the copies are blocking hipMemcpy calls on the null stream.  Device-resident sources
 * (on_device = 1) must be complete when the call is made -- work still running on a non-blocking stream is not
 * waited for; synchronise that stream (or the device) first, as the Python binding does.  The call returns after
 * the copies have finished, and later render calls on any stream see the new weights. */
int rc_load_weights(rc_handle* h, const rc_tensor_desc* descs, int32_t n);

/* -- the hot path: replaces model.apply(variables, rng, rays, ...)["render"] for the cache stage
 * (BaseMaterialModel.__call__ -> BaseNeRFModel.__call__, internal/models.py:1144-1254, 657-774). */
int rc_render_rays(rc_handle* h, const rc_rays* rays, int64_t n_rays, const rc_randoms* rnd,
                   uint32_t pass_mask, const rc_outputs* out, void* stream);

/* -- the chunk loop of models.render_image (internal/models.py:2412-2514) for passes that need no random inputs:
 * n_chunks consecutive batches of `chunk` rays out of the arrays `rays` points to (the caller has edge-padded the last
 * one: utils.shard / np.pad(mode="edge"), internal/utils.py:333-343, models.py:2437-2441), chunk i enqueued on
 * streams[i % n_streams] with its outputs at out0->ptr[k] + i * out_stride floats (k over the requested outputs).
 * Exactly n_chunks calls of rc_render_rays(rnd = NULL) -- same kernels, same results, same stream semantics per chunk --
 * without a trip through the caller's language per chunk.  (ABI v4.) */
int rc_render_chunks(rc_handle* h, const rc_rays* rays, int64_t chunk, int64_t n_chunks, uint32_t pass_mask,
                     const rc_outputs* out0, int64_t out_stride, void* const* streams, int32_t n_streams);

/* -- model.apply(..., passes=("cache","light","material")) for the material stage: the cache pass on the
 * primary rays (-> cache_out, the `cache_<k>` keys), one resampled shading point per ray, light sampler,
 * BRDF importance sampling of K secondary rays per point, ONE batched secondary trace of n*K rays through
 * the same cache kernels (is_secondary, resample, no env map) + the model-level EnvMap along the same
 * rays, Monte-Carlo BRDF integration and the MaterialIntegrator composite (-> mat_out).
 * rnd->jitter[] drives the primary rays (NULL: deterministic branch).
 * Stream semantics: the call is ordered on `stream` like every other entry point.  Inside it, work the secondary
 * trace does not wait for (light sampler, material-only composite, EnvMap) runs on a stream the handle owns, forked
 * from and joined back to `stream` with events before the call's last kernel: nothing for the caller to synchronise. */
int rc_render_material(rc_handle* h, const rc_rays* rays, int64_t n_rays, const rc_randoms* rnd,
                       const rc_material_randoms* mrnd, int32_t num_secondary_samples,
                       const rc_outputs* cache_out, const rc_mat_outputs* mat_out, void* stream);

/* -- the all-gather of the rendered pixels across the GPUs of one node: replaces
 * jax.lax.all_gather(render_dict, axis_name="batch") inside render_eval_fn (internal/train_utils.py:3795-3815) for a
 * host that drives RCCL itself (the Python host layer uses torch.distributed instead).
 *   nccl_comm   an ncclComm_t of the caller (one rank per GPU, this handle's device), passed as void*
 *   local       this rank's outputs, n_local rays each ([n_local,3] / [n_local] per slot)
 *   full        [world * n_local, .] per slot; only slots non-NULL in BOTH structs are gathered
 * Every rank must pass the same n_local and the same slot set (pad the last shard, as models.render_image pads its
 * last chunk).  All slots go out as ONE grouped collective (ncclGroupStart/End) on `stream`, asynchronously.
 * RCCL is resolved at run time from the library the process has already loaded (librccl.so, else RC_RCCL_LIBRARY):
 * the comm and the calls then belong to the same RCCL instance.  RC_ERR_UNSUPPORTED when no RCCL can be found. */
int rc_allgather_outputs(rc_handle* h, void* nccl_comm, const rc_outputs* local, int64_t n_local, const rc_outputs* full,
                         void* stream);

/* -- single operators on the path (used by the parity tests and by the roofline bench)
 * HashEncoding.__call__ incl. the contraction (internal/grid_utils.py:808-905, coord.py:37-69):
 * grid_id: 0..2 proposal density grids, 3 appearance, 4 material, 5 light.
 * points [n,3] world coordinates; features_out [n, L*F] row-major. */
int rc_hashgrid_lookup(rc_handle* h, int32_t grid_id, const float* points, int64_t n,
                       float* features_out, int32_t apply_contraction, void* stream);
/* stepfun.sample_intervals (internal/stepfun.py:207-250): t [n,P+1], logits [n,P] -> out [n,S+1];
 * jitter [n] U[0,1) or NULL. */
int rc_sample_intervals(rc_handle* h, const float* t, const float* logits, int64_t n, int32_t num_bins,
                        int32_t num_samples, const float* jitter, float* out, void* stream);

/* -- introspection for tests / profiling: named internal buffers of the last rc_render_rays
 * ("sdist0", "tdist2", "density1", "weights2", "shade_rgb", ...).  Returns RC_ERR_INVALID_ARG
 * for unknown names.  count = number of float32 elements. */
int rc_workspace_ptr(rc_handle* h, const char* name, void** ptr, int64_t* count);
/* Per-stage device time (ms, hipEvents recorded on the launch stream), averaged over the calls
 * issued since profiling was last (re)enabled (ring of 16).  mode 0 off, 1 every stage,
 * 2 only the dominant kernel (cache shader / the fused kernel; two events per call), 3 like 2 on every 8th call
 * (the last 16 sampled calls span 128 calls; an event record costs ~1.3 us of stream time).  Profiled calls launch eagerly. */
int rc_set_profiling(rc_handle* h, int32_t mode);
/* Launch mode of rc_render_rays: 0 = eager kernel launches, 1 = capture a hipGraph the second time
 * an identical call (same sizes and pointers) is seen and replay it afterwards (default),
 * 2 = capture on first sight.  Applies to the launch-per-stage plan; the one-launch fused plan (rc_set_fused) is always
 * launched plainly (a one-node graph replays with a larger gap between launches than a plain launch). */
int rc_set_graph_mode(rc_handle* h, int32_t mode);
/* Kernel plan.  1 (default): the plain cache pass (pass_mask == RC_PASS_CACHE) is one fused launch per batch, all
 * intermediates on chip (no workspace: rc_workspace_ptr then has nothing to show) -- TWO wavefronts per ray in 4-wave
 * workgroups of two rays, two workgroups per CU (csrc/rc_fused2.hip); every other pass runs one launch per stage, where
 * a proposal level whose samples only hand their density on is ONE launch (grid lookup + density MLP, weights resident
 * in LDS; from 24 576 rays on also the level's sampling, one ray per wave), and rc_render_material renders its primary
 * rays with the fused launch (one wavefront per ray, per-sample results exported).  3: like 1 with the plain cache pass
 * on the one-wavefront-per-ray form of the fused kernel (csrc/rc_fused.hip, the round-1/2 kernel).  2: like 1 with the
 * plain cache pass on the launch-per-stage plan too and the sampling always in its own kernel.  0: the plain
 * launch-per-stage plan everywhere, grid lookup and density MLP as separate kernels (materialises
 * sdist/tdist/means/features/density/weights per level in the workspace).  All plans evaluate the same arithmetic in
 * the same order: results are bitwise equal
 * (internal/models.py:1237-1386 / sampling.py:155-353 / nerf.py:426-693). */
int rc_set_fused(rc_handle* h, int32_t mode);
int rc_stage_count(void);
const char* rc_stage_name(int32_t stage);
int rc_stage_times_ms(rc_handle* h, float* out_ms, int32_t n);

/* ------------------------------------------------------------------------------------------------
 * Time-resolved cache (BASELINE configs[4]): TransientNeRFModel.__call__ for primary rays
 * (internal/models.py:912-985 over BaseNeRFModel.__call__ :657-774) = ProposalVolumeSampler ->
 * TransientNeRFMLP (internal/nerf.py:561-938, 1656-1797) -> TransientVolumeIntegrator
 * (internal/integration.py:343-551, internal/render.py:250-507).
 * ------------------------------------------------------------------------------------------------ */
typedef struct rc_transient_config {
  int32_t n_bins;                   /* Config.n_bins (700 is the only compiled size)                   */
  float exposure_time;              /* Config.exposure_time                                             */
  float tfilter_sigma;              /* Config.tfilter_sigma (0: no temporal filter)                     */
  float transient_shift;            /* Config.transient_shift                                           */
  int32_t bin_zero_threshold_light; /* Config.bin_zero_threshold_light                                  */
  float light_near;                 /* Config.light_near                                                */
  int32_t light_zero;               /* Config.light_zero                                                */
  int32_t use_falloff;              /* Config.use_falloff                                               */
  float indirect_scale;             /* TransientNeRFMLP.indirect_scale                                  */
  float rgb_max;                    /* TransientNeRFMLP.rgb_max                                         */
  float albedo_bias;                /* TransientNeRFMLP.albedo_bias (activation softplus)               */
  float brdf_bias;                  /* BaseNeRFMLP.brdf_bias                                            */
  float irradiance_bias;            /* TransientNeRFMLP.irradiance_bias                                 */
  float slf_rgb_bias;               /* TransientSurfaceLightFieldMLP.rgb_bias                           */
  int32_t use_occlusions;           /* Config.use_occlusions for every ray (the Trainer's vis_only override,
                                       engine/trainer.py:198-202): one weights-only shadow ray per shaded
                                       sample through the cache (internal/nerf.py:1193-1342)              */
  float occ_threshold;              /* Config.occ_threshold_min (== _max)                               */
  float shadow_near;                /* Config.shadow_near_min (== _max)                                 */
  float shadow_far;                 /* Config.secondary_far                                             */
  int32_t reserved[6];
} rc_transient_config;

/* Switches a freshly created handle to the transient model: rc_load_weights then expects the
 * TransientNeRFMLP inventory (direct_tint_layer, albedo_layer, brdf_layers_*, irradiance_layers_*,
 * transient_indirect_layer, light_power, SurfaceLightField with lights and 3 n_bins + 1 outputs) and
 * rc_render_transient becomes available (rc_render_rays / rc_render_material are then refused). */
int rc_set_transient(rc_handle* h, const rc_transient_config* t);

typedef enum rc_transient_output_id {
  RC_TOUT_RGB = 0,                 /* [n, n_bins, 3]  transient_direct (filtered) + transient_indirect   */
  RC_TOUT_TRANSIENT_DIRECT_VIZ,    /* [n, n_bins, 3]                                                     */
  RC_TOUT_TRANSIENT_INDIRECT_VIZ,  /* [n, n_bins, 3]  (= the reference's final "transient_indirect")     */
  RC_TOUT_TRANSIENT_INDIRECT_DIFFUSE,   /* [n, n_bins, 3] unshifted composite (integration.py extras)    */
  RC_TOUT_TRANSIENT_INDIRECT_SPECULAR,  /* [n, n_bins, 3]                                                */
  RC_TOUT_INTEGRATED_RGB,          /* [n, 3] sum of RGB over bins                                        */
  RC_TOUT_DIRECT_RGB, RC_TOUT_INDIRECT_RGB,
  RC_TOUT_DIFFUSE_RGB, RC_TOUT_SPECULAR_RGB, RC_TOUT_ALBEDO_RGB, RC_TOUT_OCC, RC_TOUT_INDIRECT_OCC,
  RC_TOUT_IRRADIANCE_RGB, RC_TOUT_LIGHT_RADIANCE_RGB, RC_TOUT_N_DOT_L_RGB, RC_TOUT_DIRECT_DIFFUSE_RGB,
  RC_TOUT_DIRECT_SPECULAR_RGB, RC_TOUT_INDIRECT_DIFFUSE_RGB, RC_TOUT_INDIRECT_SPECULAR_RGB,
  RC_TOUT_DIRECT_RGB_VIZ,          /* [n, 3] each                                                        */
  RC_TOUT_ACC, RC_TOUT_DISTANCE_MEAN, RC_TOUT_DISTANCE_MEDIAN, RC_TOUT_DISTANCE_PERCENTILE_5,
  RC_TOUT_DISTANCE_PERCENTILE_95,  /* [n]                                                                */
  RC_TOUT_MEANS, RC_TOUT_NORMALS, RC_TOUT_NORMALS_PRED,   /* [n, 3]                                       */
  RC_TOUT_RAY_DISTS, RC_TOUT_LIGHT_DISTS,                 /* [n]                                          */
  RC_TOUT_COUNT
} rc_transient_output_id;
typedef struct rc_transient_outputs { float* ptr[RC_TOUT_COUNT]; } rc_transient_outputs;   /* NULL = not wanted */

/* rays->lights is required; cam_origins [n,3] is the camera centre of each ray (Rays.cam_origins,
 * internal/inverse_render/render_utils.py:1733-1740).  Direct-light bins beyond n_bins spill into the
 * next ray of THIS call's batch exactly as the reference's flattened scatter does (internal/render.py:447-475).
 * shadow_rnd (use_occlusions only): per-level jitter of the n * 32 shadow rays, NULL = deterministic branch. */
int rc_render_transient(rc_handle* h, const rc_rays* rays, const float* cam_origins, int64_t n, const rc_randoms* rnd,
                        const rc_randoms* shadow_rnd, const rc_transient_outputs* out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * On-device ray generation: camera_utils.pixels_to_rays + cast_ray_batch
 * (internal/camera_utils.py:896-1072, 1225-1329) for one camera: ProjectionType.PERSPECTIVE / FISHEYE /
 * FISHEYE_EQUISOLID / PANORAMIC, optional radial + tangential distortion (:795-890, Newton undistortion, 10 steps),
 * optional NDC (convert_to_ndc, :50-111; radii then from the offsets between NDC origins, :1058-1066), optional
 * sub-pixel jitter offsets (:943-957, handed over as explicit random tensors like rc_randoms) and optional z_range
 * cropping (cast_ray_batch, :1291-1299; rays_planes_intersection, :1143-1164).  The outputs are the device arrays
 * rc_cast_outputs points to.
 * ------------------------------------------------------------------------------------------------ */
typedef struct rc_camera {
  float pixtocam[9];     /* inverse intrinsics, row-major [3,3] (camera_utils.get_pixtocam)              */
  float camtoworld[12];  /* extrinsics, row-major [3,4]                                                  */
  float light[3];        /* lights[cam_idx] (camera_utils.py:1288)                                       */
  float near, far;       /* Pixels.near / Pixels.far                                                     */
  int32_t camtype;       /* 0 ProjectionType.PERSPECTIVE, 1 PANORAMIC (cast_spherical_rays, camera_utils.py:1415-1443,
                          * 1013-1024: pixtocam = diag(2 pi / W, pi / H, 1), (theta, phi) -> direction),
                          * 2 FISHEYE (equidistant, theta = min(pi, r)), 3 FISHEYE_EQUISOLID (theta = 2 asin(r / 2)) (:991-1011) */
  /* -- ABI v3 -- */
  int32_t has_distortion;   /* distortion_params is not None (:981-989)                                  */
  float distortion[6];      /* k1, k2, k3, k4, p1, p2                                                    */
  int32_t has_ndc;          /* pixtocam_ndc is not None (:1052-1066)                                     */
  float pixtocam_ndc[9];    /* inverse intrinsics of the NDC projection, row-major [3,3]                 */
  /* -- ABI v4 -- */
  int32_t has_z_range;      /* z_range is not None: origins += directions * t_min, directions *= t_max - t_min for
                             * the slab z in [z_range[0], z_range[1]] (camera_utils.py:1291-1299)                 */
  float z_range[2];
  const float* pix_dx;      /* [n] device arrays or NULL (jitter = 0): the offsets pixels_to_rays adds to the pixel     */
  const float* pix_dy;      /* coordinates when jitter > 0 -- U(-0.5, 0.5) or N(0, 0.25), plus a second uniform when
                             * jitter_scale > 1 (:943-957); the caller draws them (rc_prng_fill + prng.py)         */
} rc_camera;
typedef struct rc_cast_outputs {
  float* origins; float* directions; float* viewdirs;   /* [n,3] */
  float* radii;                                         /* [n]   */
  float* imageplane;                                    /* [n,2] */
  float* look; float* up; float* lights;                /* [n,3] */
  float* near; float* far;                              /* [n]   */
} rc_cast_outputs;                                      /* NULL = not wanted */
/* pix_x / pix_y: int32 device arrays [n] (Pixels.pix_x_int / pix_y_int), or both NULL for the rectangle
 * [x0, x0 + width) x [y0, y0 + height) in row-major order (n = width * height). */
int rc_cast_rays(rc_handle* h, const rc_camera* cam, const int32_t* pix_x, const int32_t* pix_y, int64_t n,
                 int32_t x0, int32_t y0, int32_t width, int32_t height, const rc_cast_outputs* out, void* stream);

/* ---- jax.random-compatible random tensors, generated in HBM ----------------------
 * Replaces the reference's jax.random.uniform / normal / categorical(gumbel) draws on the path
 * (internal/stepfun.py:200-202 per-ray jitter, internal/models.py:240-247 resampling noise,
 * internal/light_sampler.py:140-142 constant vMF mean noise) for the PRNG the reference pins (jax==0.4.16,
 * requirements.txt:2: threefry2x32, uint32[2] keys, non-partitionable counters).  Key derivation
 * (jax.random.split through internal/utils.py:118-123 random_split) is 2-4 blocks per call and stays on the host
 * (neural-radiance-caching_amd/prng.py).
 * out: device array of n 32-bit words -- uint32 for RC_PRNG_BITS, float otherwise; element i is what
 * jax.random.{bits,uniform,normal,gumbel}(key, (n,)) holds at i (any shape with n elements, row-major).
 * minval/maxval: uniform only (normal and gumbel use jax's own ranges). */
typedef enum rc_prng_mode { RC_PRNG_MODE_BITS = 0, RC_PRNG_MODE_UNIFORM = 1, RC_PRNG_MODE_NORMAL = 2,
                            RC_PRNG_MODE_GUMBEL = 3 } rc_prng_mode;
int rc_prng_fill(rc_handle* h, const uint32_t key[2], int32_t mode, float minval, float maxval, int64_t n, void* out,
                 void* stream);

/* ---- training backward of one proposal level's density field ----------------------
 * Replaces, for the parameters of Cache/Sampler/MLP_<level>, the gradients jax.value_and_grad(loss_fn) yields in
 * train_step (internal/train_utils.py:3128-3131) for the sub-graph HashEncoding.__call__
 * (internal/grid_utils.py:808-905) -> DensityMLP.run_network (internal/geometry.py:155-168) ->
 * convert_raw_density (internal/geometry.py:318-341), given the upstream gradients of its two outputs.
 *   points     [n,3] world-space sample means (device)
 *   d_density  [n]    d L / d density            d_feature  [n,64] d L / d feature (the hidden vector the shader
 *                                                 reads) or NULL
 *   grads      device buffer of rc_density_grad_size floats, layout rc_density_grad_layout (tensor names, offsets
 *              and shapes of the reference's parameter tree); gradients are ACCUMULATED into it (zero it per step)
 *   density_out [n] or NULL: the forward value, for the caller's loss
 * Table gradients use hardware float atomics (order-dependent in the last bits); the MLP gradients are reduced in a
 * fixed order.  Across ranks the caller averages `grads` (jax.lax.pmean, train_utils.py:3133-3135) with one
 * all-reduce over RCCL (nrc_amd.train.allreduce_grads). */
typedef struct rc_grad_segment {
  char name[160];          /* e.g. params/Cache/Sampler/MLP_2/density_grid/hash_2048 */
  int64_t offset, size;    /* in floats */
  int32_t ndim;
  int64_t shape[4];
} rc_grad_segment;
int64_t rc_density_grad_size(rc_handle* h, int32_t level);
int rc_density_grad_layout(rc_handle* h, int32_t level, rc_grad_segment* segs, int32_t capacity, int32_t* count);
int rc_density_backward(rc_handle* h, int32_t level, const float* points, int64_t n, const float* d_density,
                        const float* d_feature, float* grads, float* density_out, void* stream);

/* Transpose of rc_hashgrid_lookup for any grid of the handle (0-2 proposal density grids, 3 appearance, 4 material,
 * 5 light): d_features [n, L*F] (the layout rc_hashgrid_lookup writes) is scattered into `grads`, a device buffer of
 * `total` floats holding the grid's tables in level order, each laid out like the loaded tensor
 * (rc_hashgrid_grad_layout; names = the reference's parameter paths).  Accumulates; hardware float atomics.
 * What jax's autodiff emits for HashEncoding.__call__ (internal/grid_utils.py:808-905) inside train_step. */
int rc_hashgrid_grad_layout(rc_handle* h, int32_t grid_id, rc_grad_segment* segs, int32_t capacity, int32_t* count, int64_t* total);
int rc_hashgrid_backward(rc_handle* h, int32_t grid_id, const float* points, int64_t n, const float* d_features, float* grads,
                         int32_t apply_contraction, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RC_ABI_H_ */
