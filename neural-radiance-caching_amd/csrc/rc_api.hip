// C-ABI host of the radiance-cache renderer (see include/rc_abi.h).
//
// Owns: device copies of the hash/dense tables, the MFMA-fragment-packed MLP weights, the
// per-batch workspace and the launch sequence that replaces BaseNeRFModel.__call__
// (internal/models.py:657-774): 3 x [resample -> grid lookup -> density MLP] -> (categorical
// resample) -> appearance grid -> cache shader -> volume compositing.
#include <dlfcn.h>
#include <math.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>

#include <array>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "rc_internal.h"
#include "rc_pack_host.h"

void rc_launch_hashgrid_src(const RcGridDev& g, const float* points, int soa_in, const int32_t* src, int64_t n_src,
                            int64_t n, float* out, int feature_major, int64_t ldo, float contract_radius,
                            float* jac_out, hipStream_t stream);
int rc_shader_lds_bytes();
int rc_weight_chunk_floats();
void rc_shader_prepare();

using namespace rcpack;

namespace {

thread_local std::string g_create_error;

enum Stage {
  ST_SAMPLE0 = 0, ST_GRID0, ST_MLP0, ST_SAMPLE1, ST_GRID1, ST_MLP1, ST_SAMPLE2, ST_GRID2, ST_MLP2,
  ST_RESAMPLE, ST_GRID_APP, ST_SHADER, ST_COMPOSITE, ST_COUNT
};
const char* kStageNames[ST_COUNT] = {"sample0", "grid0", "mlp0", "sample1", "grid1", "mlp1", "sample2",
                                     "grid2", "mlp2", "resample", "grid_app", "shader", "composite"};

struct DevBuf {
  float* p = nullptr;
  size_t bytes = 0;
};

constexpr int kEvSlots = 16;

// Everything that is baked into the kernel arguments of one rc_render_rays call.
struct RenderKey {
  int64_t n; uint32_t mask; int slot; int ws_slot;
  const void* rays[7]; const void* rnd[RC_MAX_LEVELS + 2]; const void* out[RC_OUT_COUNT];
  bool operator==(const RenderKey& o) const { return memcmp(this, &o, sizeof(RenderKey)) == 0; }
};
struct GraphEntry {
  RenderKey key;
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
};

struct GridState {
  rc_grid_config cfg{};
  std::string prefix;
  std::vector<int> sizes;
  std::vector<DevBuf> tables;
  std::vector<bool> loaded;
  RcGridDev dev{};
};

}  // namespace

struct rc_handle {
  rc_config cfg{};
  int device = 0;
  std::string err;
  GridState grids[6];
  std::map<std::string, HostLayer> layers;   // key: path without /kernel|/bias
  bool packed_dirty = true;
  uint64_t layers_gen = 1;                   // bumped whenever a dense layer is (re)loaded
  uint64_t train_gen[RC_MAX_LEVELS] = {};    // layers_gen the training stream of a level was packed at
  bool have_envmap = false;
  bool have_material = false;
  // packed MFMA fragments (device)
  std::map<std::string, DevBuf> packs;
  DevBuf ide_table;
  // workspace (ws_prefix selects the slot: "" / "p1:".. one per caller stream so that independent
  // batches enqueued on different streams overlap; "s:" batched secondary trace of the material stage)
  std::string ws_prefix;
  // Who used a workspace set last.  Sets 0-3 serve rc_render_rays (one per caller stream, least recently used one taken
  // over when a fifth stream shows up); set 0 also serves rc_render_material / rc_render_transient (with its "s:"
  // companion for their batched secondary trace); set 4 ("t:") serves rc_density_backward.  A call whose stream differs
  // from the set's previous user first waits for that user's last call (event), so two streams never run on one set at
  // the same time.
  struct WsGroup { hipStream_t stream = nullptr; bool used = false; hipEvent_t done = nullptr; uint64_t last_use = 0; };
  WsGroup groups[5];
  uint64_t use_clock = 0;
  int64_t ws_rays = 0;
  std::map<std::string, DevBuf> ws;
  std::map<std::string, int64_t> ws_count;
  // profiling: ring of event sets, one set per render call (slot = call % kEvSlots)
  // mode 0 off, 1 every stage, 2 only the dominant kernel (cache shader), 3 like 2 on every 8th call
  int profiling = 0;
  hipEvent_t ev[kEvSlots][ST_COUNT + 1]{};
  bool ev_used[kEvSlots]{};
  bool ev_created = false;
  int64_t prof_calls = 0;
  // time-resolved cache (rc_set_transient): TransientNeRFMLP inventory + rc_render_transient
  bool transient = false;
  rc_transient_config tcfg{};
  float light_power = 0.0f;
  bool have_light_power = false;
  DevBuf taps;
  int n_taps = 0;
  // fused per-ray kernel for the plain cache pass (fused_mode: 0 never, 1 whenever eligible)
  int fused_mode = 1;
  RcFusedLaunch fused_tmpl{};      // launch descriptor of the fused plan, resolved at repack (build_fused_template)
  int fused_stagger = getenv("RC_FUSED_STAGGER") ? atoi(getenv("RC_FUSED_STAGGER")) : 0;   // experiment (rc_fused2.hip)
  int fused_prio = getenv("RC_FUSED_PRIO") ? atoi(getenv("RC_FUSED_PRIO")) : 1;             // rc_fused2.hip: 1 = the younger workgroup of a CU leads through the lookups
#ifdef RC_FUSED_DIRECT_EXPERIMENT
  int fused_direct = getenv("RC_FUSED_DIRECT") ? atoi(getenv("RC_FUSED_DIRECT")) : 0;   // experiment switch (see rc_fused.hip, DIRECT)
#else
  int fused_direct = 0;
#endif
  bool fused_ok = false;
  bool fused_front_ok = false;               // transient handles: the FRONT variant of the fused kernel is usable
  // hipGraph replay of the launch sequence (graph_mode: 0 off, 1 capture when a call repeats, 2 always)
  int graph_mode = 1;
  hipStream_t cap_stream = nullptr;
  // rc_render_material: work that the secondary trace does not wait for (material-only composite over all samples,
  // EnvMap along the secondary rays) runs on this stream, forked from / joined to the caller's with events
  float* sec_sbounds = nullptr;            // power-ladder image of the secondary rays' (near, far): constants of the config
  hipStream_t side_stream = nullptr;
  hipEvent_t ev_side[4] = {nullptr, nullptr, nullptr, nullptr};
  // rc_density_backward: weight gradients ([0]) and the LDS-accumulated table levels ([1]) beside the scatter of the other
  // levels on the caller's stream (highest priority; events: fork, join [0], join [1])
  hipStream_t train_stream[2] = {nullptr, nullptr};
  hipEvent_t ev_train[3] = {nullptr, nullptr, nullptr};
  std::vector<GraphEntry> graphs;
  RenderKey last_key{};
  bool have_last_key = false;
};

namespace {

#define RC_HIP(h, call)                                                                      \
  do {                                                                                       \
    hipError_t e_ = (call);                                                                  \
    if (e_ != hipSuccess) {                                                                  \
      (h)->err = std::string(#call) + ": " + hipGetErrorString(e_);                          \
      return RC_ERR_HIP;                                                                     \
    }                                                                                        \
  } while (0)

int fail(rc_handle* h, int code, const std::string& msg) {
  h->err = msg;
  return code;
}

// Nothing may propagate out of an extern "C" entry point (include/rc_abi.h): std::bad_alloc / std::length_error of the
// host-side containers end up here and come back as a status code.
int rc_caught(rc_handle* h, const char* what) noexcept {
  try {
    const std::string msg = std::string("exception in the host layer: ") + (what ? what : "unknown");
    if (h) h->err = msg; else g_create_error = msg;
  } catch (...) {
  }
  return RC_ERR_HOST;
}
#define RC_TRY try {
#define RC_CATCH(h)                                                       \
  } catch (const std::exception& e_) { return rc_caught((h), e_.what()); } \
  catch (...) { return rc_caught((h), nullptr); }

std::vector<int> grid_sizes(const rc_grid_config& g) {
  // grid_utils.py:773-794 with scale_supersample = 1
  const int n = 1 + (int)lround(log2((double)g.max_grid_size / g.min_grid_size));
  std::vector<int> s;
  for (int i = 0; i < n; ++i) s.push_back((int)lround(g.min_grid_size * pow(2.0, i)));
  return s;
}
bool is_dense(const rc_grid_config& g, int n) { return (int64_t)n * n * n <= g.hash_map_size; }
std::string level_name(const rc_grid_config& g, const std::vector<int>& sizes, int n) {
  const int width = (int)std::to_string(sizes.back()).size();
  std::string d = std::to_string(n);
  while ((int)d.size() < width) d = "0" + d;
  return std::string(is_dense(g, n) ? "grid_" : "hash_") + d;
}

void init_grid(GridState& gs, const rc_grid_config& cfg, const std::string& prefix) {
  gs.cfg = cfg;
  gs.prefix = prefix;
  gs.sizes = grid_sizes(cfg);
  gs.tables.assign(gs.sizes.size(), DevBuf{});
  gs.loaded.assign(gs.sizes.size(), false);
  memset(&gs.dev, 0, sizeof(gs.dev));
  gs.dev.num_levels = (int)gs.sizes.size();
  gs.dev.num_features = cfg.num_features;
  gs.dev.bbox = cfg.bbox;
  gs.dev.precondition = cfg.precondition_scaling;
  for (size_t l = 0; l < gs.sizes.size(); ++l) {
    const int n = gs.sizes[l];
    RcGridLevel& L = gs.dev.lvl[l];
    L.size = n;
    L.dense = is_dense(cfg, n) ? 1 : 0;
    L.entries = L.dense ? (uint32_t)n * n * n : (uint32_t)cfg.hash_map_size;
    const uint32_t t = (uint32_t)cfg.hash_map_size;
    L.mask = (!L.dense && (t & (t - 1)) == 0) ? t - 1 : 0;
  }
}

// The dense layers the cache path reads: path -> (in, out).
std::map<std::string, std::pair<int, int>> dense_inventory(const rc_config& c, const rc_transient_config* tc = nullptr) {
  std::map<std::string, std::pair<int, int>> m;
  const int W = 64, B = 128;
  for (int l = 0; l < c.num_levels; ++l) {
    const std::string base = "params/Cache/Sampler/MLP_" + std::to_string(l);
    const int K = (int)grid_sizes(c.proposal_grids[l]).size() * c.proposal_grids[l].num_features;
    m[base + "/density_layers_0"] = {K, W};
    m[base + "/density_layers_1"] = {W, W};
    m[base + "/output_density_layer"] = {W, 1};
    if (l == c.num_levels - 1) m[base + "/pred_normals_layer"] = {W, 3};
  }
  const std::string sh = "params/Cache/Shader";
  const int feat = W + (int)grid_sizes(c.appearance_grid).size() * c.appearance_grid.num_features;
  if (tc) {
    // TransientNeRFMLP + TransientSurfaceLightFieldMLP (nerf.py:232-414, surface_light_field.py:343-403):
    // pos_enc degree 2 for lights and BRDF dots (15 values), IDE degree 5 (72 values)
    m[sh + "/bottleneck_layer"] = {feat, B};
    m[sh + "/roughness_layer"] = {feat, 1};
    m[sh + "/tint_layer"] = {feat, 3};
    m[sh + "/direct_tint_layer"] = {feat, 3};
    m[sh + "/albedo_layer"] = {feat, 3};
    m[sh + "/integrated_brdf_layers_0"] = {B + 1, 64};
    m[sh + "/integrated_brdf_layers_1"] = {64, 64};
    m[sh + "/output_integrated_brdf_layer"] = {64, 1};
    m[sh + "/brdf_layers_0"] = {B + 15, 64};
    m[sh + "/brdf_layers_1"] = {64, 64};
    m[sh + "/output_brdf_layer"] = {64, 1};
    m[sh + "/irradiance_layers_0"] = {feat + 15, 64};
    m[sh + "/irradiance_layers_1"] = {64, 64};
    m[sh + "/transient_indirect_layer"] = {64, 3 * tc->n_bins};
    const std::string sl = sh + "/SurfaceLightField";
    const int in = B + 72 + 15;
    m[sl + "/layer_0"] = {in, 128};
    m[sl + "/layer_1"] = {128, 128};
    m[sl + "/layer_2"] = {128, 128};
    m[sl + "/layer_bottleneck"] = {128 + in, 128};
    m[sl + "/output_rgba_layer"] = {128, 3 * tc->n_bins + 1};
    return m;
  }
  m[sh + "/bottleneck_layer"] = {feat, B};
  m[sh + "/roughness_layer"] = {feat, 1};
  m[sh + "/tint_layer"] = {feat, 3};
  m[sh + "/ambient_irradiance_layer"] = {feat, 3};
  m[sh + "/irradiance_layer"] = {feat, 3};
  m[sh + "/integrated_brdf_layers_0"] = {B + 1, 64};
  m[sh + "/integrated_brdf_layers_1"] = {64, 64};
  m[sh + "/output_integrated_brdf_layer"] = {64, 1};
  auto slf = [&](const std::string& p, int in, int w, int bott) {
    m[p + "/layer_0"] = {in, w};
    m[p + "/layer_1"] = {w, w};
    m[p + "/layer_2"] = {w, w};
    m[p + "/layer_bottleneck"] = {w + in, bott};
    m[p + "/output_rgba_layer"] = {bott, 4};
    m[p + "/output_ambient_rgb_layer"] = {bott, 3};
  };
  slf(sh + "/SurfaceLightField", B + 72, 128, 128);
  slf(sh + "/EnvMap", 38, 128, 128);          // dead work in the reference; accepted, never read
  slf("params/Cache/EnvMap", 27, 256, 128);
  // material / light stages (accepted so that one checkpoint loads; used by later passes)
  m["params/MaterialShader/bottleneck_layer"] = {32, B};
  m["params/MaterialShader/pred_brdf_layer"] = {B, 10};
  m["params/LightSampler/layers_0"] = {32, 64};
  m["params/LightSampler/layers_1"] = {64, 64};
  m["params/LightSampler/output_layer"] = {64, 640};
  return m;
}

void append(std::vector<float>& dst, const std::vector<float>& v) { dst.insert(dst.end(), v.begin(), v.end()); }
// The kernels pull the stream in whole chunks: pad with zero fragments.
std::vector<float> pad_stream(std::vector<float> v) {
  const size_t c = (size_t)rc_weight_chunk_floats();
  v.resize((v.size() + c - 1) / c * c, 0.0f);
  return v;
}

int upload(rc_handle* h, const std::string& key, const std::vector<float>& v) {
  DevBuf& b = h->packs[key];
  if (b.bytes != v.size() * sizeof(float)) {
    if (b.p) RC_HIP(h, hipFree(b.p));
    RC_HIP(h, hipMalloc((void**)&b.p, v.size() * sizeof(float)));
    b.bytes = v.size() * sizeof(float);
  }
  RC_HIP(h, hipMemcpy(b.p, v.data(), b.bytes, hipMemcpyHostToDevice));
  return RC_OK;
}

const HostLayer* need(rc_handle* h, const std::string& path, std::string& missing) {
  auto it = h->layers.find(path);
  if (it == h->layers.end() || !it->second.have_kernel || !it->second.have_bias) {
    if (missing.empty()) missing = path;
    return nullptr;
  }
  return &it->second;
}

// Geometry the fused per-ray kernel is written for: 3 proposal levels of 64 / 64 / 32 samples on grids of 6 / 7 / 8
// levels, the level-2 density grid and the appearance grid with identical level geometry, power-of-two hash tables.
bool fused_geometry_ok(rc_handle* h) {
  const rc_config& c = h->cfg;
  bool ok = c.num_levels == 3 && c.num_samples[0] == 64 && c.num_samples[1] == 64 && c.num_samples[2] == 32;
  const int want_lv[4] = {6, 7, 8, 8}, want_f[4] = {1, 1, 4, 4};
  for (int g = 0; g < 4 && ok; ++g)
    ok = h->grids[g].dev.num_levels == want_lv[g] && h->grids[g].dev.num_features == want_f[g];
  // the two half-waves of the last level look up the density / appearance grid with shared level geometry
  ok = ok && h->grids[2].dev.bbox == h->grids[3].dev.bbox && h->grids[2].dev.precondition == h->grids[3].dev.precondition;
  for (int l = 0; l < 8 && ok; ++l) {
    const RcGridLevel &A = h->grids[2].dev.lvl[l], &B = h->grids[3].dev.lvl[l];
    ok = A.dense == B.dense && A.size == B.size && A.mask == B.mask && A.entries == B.entries;
  }
  for (int g = 0; g < 4 && ok; ++g)        // hashed levels: power-of-two tables only (index = hash & mask)
    for (int l = 0; l < h->grids[g].dev.num_levels && ok; ++l)
      ok = h->grids[g].dev.lvl[l].dense || h->grids[g].dev.lvl[l].mask != 0;
  // the kernels are compiled for the reference's level layout (16 ... 2048 cells a side against 2^19 entries): the
  // first kRcFusedDenseLevels levels dense, the others hashed (the kind of a level is a compile-time constant there)
  for (int g = 0; g < 4 && ok; ++g)
    for (int l = 0; l < h->grids[g].dev.num_levels && ok; ++l)
      ok = (h->grids[g].dev.lvl[l].dense != 0) == (l < kRcFusedDenseLevels);
  return ok;
}

// Derived table copies of the fused kernel: interleaved level-2 pairs, cell tables of the dense levels.
int build_fused_tables(rc_handle* h) {
  int rc;
  {
    // level-2 density + appearance tables interleaved entry by entry (same index in both grids): [dens 4 | app 4]
    for (int l = 0; l < h->grids[2].dev.num_levels; ++l) {
      if (h->grids[2].dev.lvl[l].dense) continue;          // dense levels: cell tables below
      const size_t entries = h->grids[2].dev.lvl[l].entries;
      DevBuf& b = h->packs["pair_" + std::to_string(l)];
      const size_t bytes = entries * 8 * sizeof(float);
      if (b.bytes != bytes) {
        if (b.p) RC_HIP(h, hipFree(b.p));
        RC_HIP(h, hipMalloc((void**)&b.p, bytes));
        b.bytes = bytes;
      }
      RC_HIP(h, hipMemcpy2D(b.p, 32, h->grids[2].dev.lvl[l].table, 16, 16, entries, hipMemcpyDeviceToDevice));
      RC_HIP(h, hipMemcpy2D(b.p + 4, 32, h->grids[3].dev.lvl[l].table, 16, 16, entries, hipMemcpyDeviceToDevice));
    }
    // dense levels as cell tables (the 8 corners of every cell of the zero-padded volume side by side):
    // proposal grids 0 / 1: 8 floats per cell; level-2 pair: 8 x [density 4 | appearance 4] per cell
    auto cells = [&](const std::string& key, size_t floats, float** out) -> int {
      DevBuf& b = h->packs[key];
      if (b.bytes != floats * sizeof(float)) {
        if (b.p) RC_HIP(h, hipFree(b.p));
        RC_HIP(h, hipMalloc((void**)&b.p, floats * sizeof(float)));
        b.bytes = floats * sizeof(float);
      }
      *out = b.p;
      return RC_OK;
    };
    for (int g = 0; g < 3; ++g)
      for (int l = 0; l < h->grids[g].dev.num_levels; ++l) {
        const RcGridLevel& L = h->grids[g].dev.lvl[l];
        if (!L.dense) {
          // the first hashed levels of the F = 1 grids as cell records (kLevelHRec): (N + 1)^3 records of 8 floats
          // g == 2: the F = 4 density grid (level kernels' lean pass).  Its records are 2.5 GB per handle: RC_REC4_TABLES=0 in
          // the environment leaves them out (the level kernels test L.rec at run time; the material step is 2.4 % slower)
          static const bool rec4_on = !(getenv("RC_REC4_TABLES") && getenv("RC_REC4_TABLES")[0] == '0');
          const int nrl = g < 2 ? kRcRecLevels : (rec4_on ? kRcRec4Levels : 0);
          if (l < kRcFusedDenseLevels + nrl && L.mask != 0) {
            const int F = h->grids[g].dev.num_features;
            const size_t nrec = (size_t)(L.size + 1) * (L.size + 1) * (L.size + 1);
            float* dst = nullptr;
            if ((rc = cells("rec" + std::to_string(g) + "_" + std::to_string(l), nrec * 8 * F, &dst))) return rc;
            rc_launch_build_hrec(L.table, L.size, L.mask, F, dst, nullptr);
            h->grids[g].dev.lvl[l].rec = dst;
          }
          continue;
        }
        const size_t ncell = (size_t)(L.size + 3) * (L.size + 3) * (L.size + 3);
        float* dst = nullptr;
        if (g < 2) {
          if ((rc = cells("cell" + std::to_string(g) + "_" + std::to_string(l), ncell * 8, &dst))) return rc;
          rc_launch_build_cells(L.table, L.size, 1, dst, 1, 0, nullptr);
          h->grids[g].dev.lvl[l].cell = dst;       // the launch-per-stage gather reads it as well
        } else {
          if ((rc = cells("pair_" + std::to_string(l), ncell * 64, &dst))) return rc;      // replaces the flat pair table
          rc_launch_build_cells(L.table, L.size, 4, dst, 8, 0, nullptr);
          rc_launch_build_cells(h->grids[3].dev.lvl[l].table, L.size, 4, dst, 8, 4, nullptr);
        }
      }
  }
  RC_HIP(h, hipDeviceSynchronize());
  return RC_OK;
}

// The per-batch launch descriptor of the fused plan with every pointer and constant that only changes with the weights.
void build_fused_template(rc_handle* h) {
  const rc_config& c = h->cfg;
  RcFusedLaunch F{};
  for (int l = 0; l < 3; ++l) F.num_samples[l] = c.num_samples[l];
  for (int g = 0; g < 4; ++g) F.grid[g] = &h->grids[g].dev;
  for (int l = 0; l < h->grids[2].dev.num_levels; ++l) F.pair_table[l] = h->packs["pair_" + std::to_string(l)].p;
  for (int g = 0; g < 2; ++g)
    for (int l = 0; l < h->grids[g].dev.num_levels; ++l)
      F.cell_table[g][l] = h->grids[g].dev.lvl[l].dense ? h->packs["cell" + std::to_string(g) + "_" + std::to_string(l)].p
                                                        : h->grids[g].dev.lvl[l].rec;      // cell records of a kLevelHRec level, or NULL
  F.wstream = h->packs["fused"].p; F.ide_coef = h->ide_table.p;
  F.anneal = c.anneal; F.padding = c.resample_padding; F.density_bias = c.density_bias;
  F.contract_radius = c.contract_radius; F.bg = c.bg_intensity;
  for (int i = 0; i < 3; ++i) F.pct[i] = c.percentiles[i];
  F.roughness_bias = c.roughness_bias; F.irradiance_bias = c.irradiance_bias; F.ambient_bias = c.ambient_irradiance_bias;
  F.rgb_max = c.rgb_max; F.slf_ambient_bias = c.slf_ambient_bias;
  h->fused_tmpl = F;
}

// The handle's side stream (work that only feeds a call's outputs, forked from / joined to the caller's stream with events)
// at the lowest priority: the kernels on the caller's stream -- the critical path -- get the CUs first.
// Helper streams are per PROCESS and device, created together by the first rc_create and never destroyed: [0] lowest
// priority (the material stage's side work), [1], [2] highest priority (rc_density_backward).  Measured on this runtime
// (ROCm 7.2, profiles/r04_helper_streams.txt): with streams that come and go with their handles, whichever multi-stream
// call ran SECOND in a process was slow -- every kernel of the call +50 us or 2-3x, ~100 us of host time between calls:
// bench.py's train line after its material line 0.20 -> 0.37 ms per level-2 call, the other order the material stage
// 1.39 -> 1.95 ms.  Sharing the streams across handles did not cure it, creating all of them before any work did.
// Handles are driven by one host thread per GPU; every call forks from and joins to the caller's stream with the
// handle's own events, so sharing the streams only orders the helper work of two handles.
hipStream_t rc_helper_stream(int which) {
  static hipStream_t pool[64][3] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  hipStream_t& s = pool[dev & 63][which];
  if (!s) {
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    if (hipStreamCreateWithPriority(&s, hipStreamNonBlocking, which == 0 ? prio_lo : prio_hi) != hipSuccess) s = nullptr;
  }
  return s;
}

int ensure_side_stream(rc_handle* h) {
  if (h->side_stream) return RC_OK;
  h->side_stream = rc_helper_stream(0);
  if (!h->side_stream) return fail(h, RC_ERR_HIP, "side stream");
  for (hipEvent_t& e : h->ev_side) RC_HIP(h, hipEventCreateWithFlags(&e, hipEventDisableTiming));
  return RC_OK;
}

int repack_transient(rc_handle* h);

int repack(rc_handle* h) {
  std::string missing;
  const rc_config& c = h->cfg;
  std::vector<float> fused_parts[4];
  // all grids of the cache path must be loaded
  for (int g = 0; g < 4; ++g)
    for (size_t l = 0; l < h->grids[g].sizes.size(); ++l)
      if (!h->grids[g].loaded[l])
        return fail(h, RC_ERR_MISSING_WEIGHT,
                    "missing weight: " + h->grids[g].prefix + "/" + level_name(h->grids[g].cfg, h->grids[g].sizes, h->grids[g].sizes[l]));
  for (int l = 0; l < c.num_levels; ++l) {
    const std::string base = "params/Cache/Sampler/MLP_" + std::to_string(l);
    const HostLayer* d0 = need(h, base + "/density_layers_0", missing);
    const HostLayer* d1 = need(h, base + "/density_layers_1", missing);
    const HostLayer* dout = need(h, base + "/output_density_layer", missing);
    const HostLayer* dn = (l == c.num_levels - 1) ? need(h, base + "/pred_normals_layer", missing) : nullptr;
    if (!missing.empty()) return fail(h, RC_ERR_MISSING_WEIGHT, "missing weight: " + missing);
    std::vector<Step> s;
    steps_natural(s, d0->in, 0); step_bias(s);
    // density MLPs: fp32 fragments in every build (rc_pack_host.h rc_lfr32, rc_dev_mlp.h mlp_layer_d)
    std::vector<float> stream = pack_f32(s, {tile_full(d0, 0), tile_full(d0, 1)});
    s.clear(); steps_acc(s, 2, 0); step_bias(s);
    append(stream, pack_f32(s, {tile_full(d1, 0), tile_full(d1, 1)}));
    std::vector<Col> regs = {Col{dout, 0}};
    if (dn) { regs.push_back(Col{dn, 0}); regs.push_back(Col{dn, 1}); regs.push_back(Col{dn, 2}); }
    append(stream, pack_dot(regs, 2, false));
    if (dn) {
      // backward fragments for the analytic normals (last level): W1^T, W0^T (w_out is kept from the forward dot)
      HostLayer w1t, w0t;
      w1t.in = d1->out; w1t.out = d1->in; w1t.kernel.resize(d1->kernel.size()); w1t.bias.assign(w1t.out, 0.0f);
      for (int r = 0; r < d1->in; ++r) for (int c2 = 0; c2 < d1->out; ++c2) w1t.kernel[(size_t)c2 * w1t.out + r] = d1->kernel[(size_t)r * d1->out + c2];
      w0t.in = d0->out; w0t.out = d0->in; w0t.kernel.resize(d0->kernel.size()); w0t.bias.assign(w0t.out, 0.0f);
      for (int r = 0; r < d0->in; ++r) for (int c2 = 0; c2 < d0->out; ++c2) w0t.kernel[(size_t)c2 * w0t.out + r] = d0->kernel[(size_t)r * d0->out + c2];
      std::vector<Step> sb; steps_acc(sb, 2, 0);
      append(stream, pack_f32(sb, {tile_full(&w1t, 0, 0, false), tile_full(&w1t, 1, 0, false)}));
      append(stream, pack_f32(sb, {tile_full(&w0t, 0, 0, false)}));
    }
    if (l < 3) fused_parts[l] = stream;
    int rc = upload(h, "dens_" + std::to_string(l), pad_stream(stream));
    if (rc) return rc;
  }
  if (h->transient) {
    // front end of the time-resolved cache through the fused kernel's FRONT variant: [dens 0 | dens 1 | dens 2 (+ bwd)]
    int off[4];
    rc_fused_stream_offsets(&off[0], &off[1], &off[2], &off[3]);
    bool ok = fused_geometry_ok(h);
    std::vector<float> stream;
    for (int p = 0; p < 3 && ok; ++p) {
      ok = (int)(stream.size() / 64) == off[p];
      append(stream, fused_parts[p]);
    }
    if (ok && (int)(stream.size() / 64) < off[3] && off[3] - (int)(stream.size() / 64) < 4) stream.resize((size_t)off[3] * 64, 0.0f);
    ok = ok && (int)(stream.size() / 64) == off[3];
    h->fused_front_ok = ok;
    if (ok) {
      int rc = upload(h, "fused_front", pad_stream(stream));
      if (rc) return rc;
      if ((rc = build_fused_tables(h))) return rc;
    }
    return repack_transient(h);
  }
  const std::string sh = "params/Cache/Shader";
  const HostLayer* bott = need(h, sh + "/bottleneck_layer", missing);
  const HostLayer* rough = need(h, sh + "/roughness_layer", missing);
  const HostLayer* tint = need(h, sh + "/tint_layer", missing);
  const HostLayer* amb = need(h, sh + "/ambient_irradiance_layer", missing);
  const HostLayer* irr = need(h, sh + "/irradiance_layer", missing);
  const HostLayer* i0 = need(h, sh + "/integrated_brdf_layers_0", missing);
  const HostLayer* i1 = need(h, sh + "/integrated_brdf_layers_1", missing);
  const HostLayer* io = need(h, sh + "/output_integrated_brdf_layer", missing);
  const std::string sl = sh + "/SurfaceLightField";
  const HostLayer* l0 = need(h, sl + "/layer_0", missing);
  const HostLayer* l1 = need(h, sl + "/layer_1", missing);
  const HostLayer* l2 = need(h, sl + "/layer_2", missing);
  const HostLayer* lb = need(h, sl + "/layer_bottleneck", missing);
  const HostLayer* la = need(h, sl + "/output_ambient_rgb_layer", missing);
  if (!missing.empty()) return fail(h, RC_ERR_MISSING_WEIGHT, "missing weight: " + missing);
  {
    std::vector<float> stream;
    std::vector<Step> s;
    // The shader bottleneck is linear and only feeds linear layers: fold it (fp64 products) into SLF layer_0, the
    // input part of SLF layer_bottleneck and integrated_brdf_layers_0 (see rc_dev_mlp.h, kShActSteps).
    const int FE = bott->in;                                 // 96 (the bottleneck is 96 -> 128)
    auto fold = [&](const HostLayer* L, int row0, int extra_rows) { return rcpack::fold_linear(*bott, *L, row0, extra_rows); };
    HostLayer l0f = fold(l0, 0, 72), lbf = fold(lb, 128, 72), i0f = fold(i0, 0, 1);
    for (int o = 0; o < l0->out; ++o) l0f.bias[o] = (float)((double)l0f.bias[o] + (double)l0->bias[o]);
    for (int o = 0; o < lb->out; ++o) lbf.bias[o] = (float)((double)lbf.bias[o] + (double)lb->bias[o]);
    for (int o = 0; o < i0->out; ++o) i0f.bias[o] = (float)((double)i0f.bias[o] + (double)i0->bias[o]);
    // heads: feature = [density feature (acc order) | appearance (natural)] + bias
    steps_acc(s, 2, 0); steps_natural(s, 32, 64); step_bias(s);
    std::vector<Col> regs = {Col{rough, 0}, Col{tint, 0}, Col{tint, 1}, Col{tint, 2}, Col{amb, 0},
                             Col{amb, 1},   Col{amb, 2},  Col{irr, 0},  Col{irr, 1},  Col{irr, 2}};
    append(stream, pack(s, {tile_by_reg(regs)}));
    // s0: folded SLF layer_0 + folded input part of layer_bottleneck over [feature | IDE (real | imag) | bias]
    s.clear(); steps_acc(s, 2, 0); steps_natural(s, 32, 64);
    for (int i = 0; i < 36; ++i) s.push_back({{FE + i, FE + 36 + i}});
    step_bias(s);
    append(stream, pack(s, {tile_full(&l0f, 0), tile_full(&l0f, 1), tile_full(&l0f, 2), tile_full(&l0f, 3),
                            tile_full(&lbf, 0), tile_full(&lbf, 1), tile_full(&lbf, 2), tile_full(&lbf, 3)}));
    // integrated BRDF: folded first layer on [feature | (n.v | bias)]
    s.clear(); steps_acc(s, 2, 0); steps_natural(s, 32, 64); s.push_back({{FE, -2}});
    append(stream, pack(s, {tile_full(&i0f, 0), tile_full(&i0f, 1)}));
    s.clear(); steps_acc(s, 2, 0); step_bias(s);
    append(stream, pack(s, {tile_full(i1, 0), tile_full(i1, 1)}));
    append(stream, pack_dot({Col{io, 0}}, 2));
    // SLF trunk
    s.clear(); steps_acc(s, 4, 0); step_bias(s);
    append(stream, pack(s, {tile_full(l1, 0), tile_full(l1, 1), tile_full(l1, 2), tile_full(l1, 3)}));
    append(stream, pack(s, {tile_full(l2, 0), tile_full(l2, 1), tile_full(l2, 2), tile_full(l2, 3)}));
    std::vector<Step> sb; steps_acc(sb, 4, 0);
    append(stream, pack(sb, {tile_full(lb, 0, 0, false), tile_full(lb, 1, 0, false), tile_full(lb, 2, 0, false),
                             tile_full(lb, 3, 0, false)}));
    append(stream, pack_dot({Col{la, 0}, Col{la, 1}, Col{la, 2}}, 4));
    fused_parts[3] = stream;
    int rc = upload(h, "shader", pad_stream(stream));
    if (rc) return rc;
  }
  {
    // fused per-ray kernel (rc_fused.hip): one stream [density MLP 0 | 1 | 2 | shader]; compiled for the
    // hotdog layout only (3 levels of 64/64/32 samples, 6/7/32 density features, 32 appearance features)
    int off[4];
    const int nf = rc_fused_stream_offsets(&off[0], &off[1], &off[2], &off[3]);
    bool ok = fused_geometry_ok(h);
    std::vector<float> stream;
    for (int p = 0; p < 4 && ok; ++p) {
      if (p == 3 && (int)(stream.size() / 64) < off[p] && off[p] - (int)(stream.size() / 64) < 4) stream.resize((size_t)off[p] * 64, 0.0f);   // split layers start on a whole piece (F_SH)
      ok = (int)(stream.size() / 64) == off[p];
      append(stream, fused_parts[p]);
    }
    ok = ok && (int)(stream.size() / 64) == nf;
    h->fused_ok = ok;
    if (ok) {
      int rc = upload(h, "fused", pad_stream(stream));
      if (rc) return rc;
      if ((rc = build_fused_tables(h))) return rc;
    }
  }
  {
    // model-level EnvMap (secondary-ray background); optional until a secondary pass is requested
    std::string miss;
    const std::string ep = "params/Cache/EnvMap";
    const HostLayer* e0 = need(h, ep + "/layer_0", miss);
    const HostLayer* e1 = need(h, ep + "/layer_1", miss);
    const HostLayer* e2 = need(h, ep + "/layer_2", miss);
    const HostLayer* eb = need(h, ep + "/layer_bottleneck", miss);
    const HostLayer* eo = need(h, ep + "/output_rgba_layer", miss);
    h->have_envmap = miss.empty();
    if (h->have_envmap) {
      std::vector<float> stream;
      std::vector<Step> s;
      auto tiles8 = [&](const HostLayer* L) {
        std::vector<Tile> t;
        for (int i = 0; i < 8; ++i) t.push_back(tile_full(L, i));
        return t;
      };
      steps_natural(s, 27, 0); step_bias(s);
      append(stream, pack(s, tiles8(e0)));
      s.clear(); steps_acc(s, 8, 0); step_bias(s);
      append(stream, pack(s, tiles8(e1)));
      append(stream, pack(s, tiles8(e2)));
      s.clear(); steps_acc(s, 8, 0);
      append(stream, pack(s, {tile_full(eb, 0, 0, false), tile_full(eb, 1, 0, false), tile_full(eb, 2, 0, false),
                              tile_full(eb, 3, 0, false)}));
      s.clear(); steps_natural(s, 27, 0); step_bias(s);
      append(stream, pack(s, {tile_full(eb, 0, 256), tile_full(eb, 1, 256), tile_full(eb, 2, 256), tile_full(eb, 3, 256)}));
      s.clear(); steps_acc(s, 4, 0); step_bias(s);
      append(stream, pack(s, {tile_by_reg({Col{eo, 0}, Col{eo, 1}, Col{eo, 2}})}));
      int rc = upload(h, "envmap", pad_stream(stream));
      if (rc) return rc;
    }
  }
  {
    // material / light heads run as plain per-point kernels on the Flax layout
    h->have_material = true;
    for (const char* pth : {"params/MaterialShader/bottleneck_layer", "params/MaterialShader/pred_brdf_layer",
                            "params/LightSampler/layers_0", "params/LightSampler/layers_1", "params/LightSampler/output_layer"}) {
      std::string miss;
      const HostLayer* L = need(h, pth, miss);
      if (!L) { h->have_material = false; continue; }
      int rc = upload(h, std::string("raw:") + pth + "/kernel", L->kernel);
      if (rc) return rc;
      if ((rc = upload(h, std::string("raw:") + pth + "/bias", L->bias))) return rc;
    }
    for (int g = 4; g < 6; ++g)
      for (size_t l = 0; l < h->grids[g].sizes.size(); ++l)
        if (!h->grids[g].loaded[l]) h->have_material = false;
  }
  if (!h->ide_table.p) {
    RcIdeTable tb;
    build_ide_table(tb);
    RC_HIP(h, hipMalloc((void**)&h->ide_table.p, sizeof(tb)));
    h->ide_table.bytes = sizeof(tb);
    RC_HIP(h, hipMemcpy(h->ide_table.p, &tb, sizeof(tb), hipMemcpyHostToDevice));
  }
  if (h->fused_ok) build_fused_template(h);
  h->packed_dirty = false;
  return RC_OK;
}

// ---------------------------------------------------------------------------------------------
// Workspace
// ---------------------------------------------------------------------------------------------
void drop_graphs(rc_handle* h);

int ws_alloc(rc_handle* h, const std::string& name0, int64_t count) {
  const std::string name = h->ws_prefix + name0;
  DevBuf& b = h->ws[name];
  const size_t bytes = (size_t)count * sizeof(float);
  if (b.bytes < bytes) {
    // captured graphs have workspace pointers baked into their kernel arguments: none survives a reallocation
    // (whichever entry point grew the buffer), so this is the one place that invalidates them
    drop_graphs(h);
    if (b.p) RC_HIP(h, hipFree(b.p));
    RC_HIP(h, hipMalloc((void**)&b.p, bytes));
    b.bytes = bytes;
  }
  h->ws_count[name] = count;
  return RC_OK;
}
float* W(rc_handle* h, const std::string& name) { return h->ws[h->ws_prefix + name].p; }

int ensure_workspace(rc_handle* h, int64_t n) {
  // ws_alloc only (re)allocates when a buffer is too small and always records the current element
  // count, so after the largest batch has been seen this is allocation-free.
  const rc_config& c = h->cfg;
  int rc;
  for (int l = 0; l < c.num_levels; ++l) {
    const int64_t S = c.num_samples[l];
    const std::string L = std::to_string(l);
    const int LF = h->grids[l].dev.num_levels * h->grids[l].dev.num_features;
    if ((rc = ws_alloc(h, "sdist" + L, n * (S + 1)))) return rc;
    if ((rc = ws_alloc(h, "tdist" + L, n * (S + 1)))) return rc;
    if ((rc = ws_alloc(h, "means" + L, 3 * n * S))) return rc;
    if ((rc = ws_alloc(h, "feat" + L, (int64_t)LF * n * S))) return rc;
    if ((rc = ws_alloc(h, "density" + L, n * S))) return rc;
    if ((rc = ws_alloc(h, "weights" + L, n * S))) return rc;
  }
  const int64_t S2 = c.num_samples[c.num_levels - 1];
  const int64_t np = n * S2;
  if ((rc = ws_alloc(h, "hbuf", ((np + 31) / 32) * 32 * 64))) return rc;
  if ((rc = ws_alloc(h, "normals_pred", 3 * np))) return rc;
  if ((rc = ws_alloc(h, "normals_grad", 3 * np))) return rc;
  if ((rc = ws_alloc(h, "jac", 3 * 32 * np))) return rc;
  if ((rc = ws_alloc(h, "app", 32 * np))) return rc;
  if ((rc = ws_alloc(h, "shade", RC_SHADE_CH * np))) return rc;
  if ((rc = ws_alloc(h, "debug", 32 * ((np + 31) / 32) + 64))) return rc;
  if ((rc = ws_alloc(h, "env_rgb", 3 * n))) return rc;
  if ((rc = ws_alloc(h, "rgb_noenv", 3 * n))) return rc;
  if ((rc = ws_alloc(h, "acc_ws", n))) return rc;
  if ((rc = ws_alloc(h, "inds", n))) return rc;
  if ((rc = ws_alloc(h, "src_idx", n))) return rc;
  if ((rc = ws_alloc(h, "filt_weight", n))) return rc;
  if ((rc = ws_alloc(h, "acc_sel", n))) return rc;
  // lean resampling pass: last-level features / hidden vector / predicted normals of the picked samples only
  if ((rc = ws_alloc(h, "feat_sel", 32 * n))) return rc;
  if ((rc = ws_alloc(h, "hbuf_sel", ((n + 31) / 32) * 32 * 64))) return rc;
  if ((rc = ws_alloc(h, "normals_sel", 3 * n))) return rc;
  if ((rc = ws_alloc(h, "density_sel", n))) return rc;
  if (h->transient && h->ws_prefix.empty()) {
    const int64_t tiles = (np + 31) / 32;
    if ((rc = ws_alloc(h, "t_irr", tiles * 32 * 64))) return rc;
    if ((rc = ws_alloc(h, "t_slf", tiles * 64 * 64))) return rc;
    if ((rc = ws_alloc(h, "tshade", (int64_t)RC_TS_COUNT * np))) return rc;
  }
  if (n > h->ws_rays) h->ws_rays = n;
  return RC_OK;
}

// roctx ranges around the stages (tracing), for `rocprofv3 --marker-trace`.  The marker library is resolved at run
// time and only when RC_ROCTX=1 is set (no link dependency, nothing on the launch path otherwise); a range covers the
// host-side enqueue of a stage -- the kernels run asynchronously and carry their own names in the kernel trace.
struct Roctx {
  int (*push)(const char*) = nullptr;
  int (*pop)() = nullptr;
  bool tried = false, open = false;
};
Roctx g_roctx;
bool roctx_on() {
  if (!g_roctx.tried) {
    g_roctx.tried = true;
    const char* env = getenv("RC_ROCTX");
    if (env && atoi(env) != 0) {
      void* lib = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_GLOBAL);
      if (!lib) lib = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
      if (lib) {
        g_roctx.push = reinterpret_cast<int (*)(const char*)>(dlsym(lib, "roctxRangePushA"));
        g_roctx.pop = reinterpret_cast<int (*)()>(dlsym(lib, "roctxRangePop"));
        if (!g_roctx.push || !g_roctx.pop) { g_roctx.push = nullptr; g_roctx.pop = nullptr; }
      }
    }
  }
  return g_roctx.push != nullptr;
}
// close the open stage range (if any) and open `name` (nullptr: only close)
void roctx_stage(const char* name) {
  if (!roctx_on()) return;
  if (g_roctx.open) { (void)g_roctx.pop(); g_roctx.open = false; }
  if (name) { (void)g_roctx.push(name); g_roctx.open = true; }
}
struct RoctxScope {       // a whole ABI call
  bool on;
  explicit RoctxScope(const char* name) : on(roctx_on()) { if (on) (void)g_roctx.push(name); }
  ~RoctxScope() { if (on) { roctx_stage(nullptr); (void)g_roctx.pop(); } }
};

void stage_mark(rc_handle* h, int slot, int idx, hipStream_t s) {
  roctx_stage(idx < ST_COUNT ? kStageNames[idx] : nullptr);
  if (slot < 0) return;
  if (h->profiling >= 2 && idx != ST_SHADER && idx != ST_SHADER + 1) return;
  (void)hipEventRecord(h->ev[slot][idx], s);
}

int ws_enter(rc_handle* h, int g, hipStream_t st) {
  rc_handle::WsGroup& G = h->groups[g];
  if (!G.done) RC_HIP(h, hipEventCreateWithFlags(&G.done, hipEventDisableTiming));
  if (G.used && G.stream != st) RC_HIP(h, hipStreamWaitEvent(st, G.done, 0));
  G.stream = st; G.used = true; G.last_use = ++h->use_clock;
  return RC_OK;
}
int ws_leave(rc_handle* h, int g, hipStream_t st) {
  RC_HIP(h, hipEventRecord(h->groups[g].done, st));
  return RC_OK;
}
// Workspace set of rc_render_rays for caller stream `st`: its own, else a free one, else the least recently used.
int ws_pick(rc_handle* h, hipStream_t st) {
  int free_i = -1, lru = 0;
  for (int i = 0; i < 4; ++i) {
    if (h->groups[i].used && h->groups[i].stream == st) return i;
    if (!h->groups[i].used && free_i < 0) free_i = i;
    if (h->groups[i].last_use < h->groups[lru].last_use) lru = i;
  }
  return free_i >= 0 ? free_i : lru;
}

void drop_graphs(rc_handle* h) {
  for (auto& g : h->graphs) {
    if (g.exec) (void)hipGraphExecDestroy(g.exec);
    if (g.graph) (void)hipGraphDestroy(g.graph);
  }
  h->graphs.clear();
  h->have_last_key = false;
}

// Secondary rays: rgb += env * (1 - acc) (Model._composite_env_map, internal/models.py:423-460).
__global__ void k_env_combine(int64_t n, const float* rgb_noenv, const float* acc, const float* env, float* rgb_out,
                              float* acc_out, float* env_out, float* noenv_out) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const float a = acc[r];
  if (acc_out) acc_out[r] = a;
  for (int c = 0; c < 3; ++c) {
    const float base = rgb_noenv[3 * r + c];
    const float e = env ? env[3 * r + c] : 0.0f;
    if (rgb_out) rgb_out[3 * r + c] = env ? base + e * (1.0f - a) : base;
    if (env_out) env_out[3 * r + c] = e;
    if (noenv_out) noenv_out[3 * r + c] = base;
  }
}

}  // namespace

extern "C" {

int rc_abi_version(void) { return RC_ABI_VERSION; }
int rc_mlp_arithmetic(void) { return kRcSplit ? 1 : 0; }
int rc_stage_count(void) { return ST_COUNT; }
const char* rc_stage_name(int32_t s) { return (s >= 0 && s < ST_COUNT) ? kStageNames[s] : ""; }

const char* rc_last_error(const rc_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int rc_create(const rc_config* cfg, int device, rc_handle** out) {
  RC_TRY
  if (!cfg || !out) { g_create_error = "rc_create: null argument"; return RC_ERR_INVALID_ARG; }
  if (cfg->abi_version != RC_ABI_VERSION) { g_create_error = "rc_create: abi_version mismatch"; return RC_ERR_INVALID_ARG; }
  if (cfg->num_levels < 1 || cfg->num_levels > RC_MAX_LEVELS) { g_create_error = "rc_create: num_levels out of range"; return RC_ERR_INVALID_ARG; }
  for (int l = 0; l < cfg->num_levels; ++l)
    if (cfg->num_samples[l] < 2 || cfg->num_samples[l] > 64) { g_create_error = "rc_create: num_samples must be in [2, 64]"; return RC_ERR_UNSUPPORTED; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { g_create_error = "rc_create: no HIP device"; return RC_ERR_NO_DEVICE; }
  if (device < 0 || device >= ndev) { g_create_error = "rc_create: bad device index"; return RC_ERR_INVALID_ARG; }
  if (hipSetDevice(device) != hipSuccess) { g_create_error = "rc_create: hipSetDevice failed"; return RC_ERR_HIP; }
  std::unique_ptr<rc_handle> hp(new rc_handle());
  rc_handle* h = hp.get();
  h->cfg = *cfg;
  h->device = device;
  // all helper streams of the process now, in a fixed order, before any work (rc_helper_stream: a low-priority stream
  // first created AFTER the high-priority ones had run took the material stage from 1.39 to 1.95 ms)
  for (int i = 0; i < 3; ++i) (void)rc_helper_stream(i);
  const rc_grid_config* gcfgs[6] = {&cfg->proposal_grids[0], &cfg->proposal_grids[1], &cfg->proposal_grids[2],
                                    &cfg->appearance_grid, &cfg->material_grid, &cfg->light_grid};
  const char* prefixes[6] = {"params/Cache/Sampler/MLP_0/density_grid", "params/Cache/Sampler/MLP_1/density_grid",
                             "params/Cache/Sampler/MLP_2/density_grid", "params/Cache/Shader/appearance_grid",
                             "params/MaterialShader/material_grid", "params/LightSampler/light_grid"};
  for (int g = 0; g < 6; ++g) {
    if (gcfgs[g]->num_features != 1 && gcfgs[g]->num_features != 4) {
      g_create_error = "rc_create: num_features must be 1 or 4";
      return RC_ERR_UNSUPPORTED;
    }
    if ((int)grid_sizes(*gcfgs[g]).size() > RC_MAX_GRID_LEVELS) {
      g_create_error = "rc_create: too many grid levels";
      return RC_ERR_UNSUPPORTED;
    }
    init_grid(h->grids[g], *gcfgs[g], prefixes[g]);
  }
  // shapes the MFMA kernels are written for
  const int K0 = h->grids[0].dev.num_levels * h->grids[0].dev.num_features;
  const int K1 = h->grids[1].dev.num_levels * h->grids[1].dev.num_features;
  const int K2 = h->grids[2].dev.num_levels * h->grids[2].dev.num_features;
  const int KA = h->grids[3].dev.num_levels * h->grids[3].dev.num_features;
  auto ks_ok = [](int K) { const int ks = (K + 1) / 2 + 1; return ks == 4 || ks == 5 || ks == 17; };
  if (cfg->num_levels != 3 || !ks_ok(K0) || !ks_ok(K1) || K2 != 32 || KA != 32) {
    g_create_error = "rc_create: unsupported grid feature widths (kernels are built for the hotdog/ngp_yobo shapes)";
    return RC_ERR_UNSUPPORTED;
  }
  *out = hp.release();
  return RC_OK;
  RC_CATCH(nullptr)
}

void rc_destroy(rc_handle* h) {
  if (!h) return;
  try {
  (void)hipSetDevice(h->device);
  for (auto& g : h->grids)
    for (auto& t : g.tables)
      if (t.p) (void)hipFree(t.p);
  for (auto& kv : h->packs) if (kv.second.p) (void)hipFree(kv.second.p);
  for (auto& kv : h->ws) if (kv.second.p) (void)hipFree(kv.second.p);
  if (h->ide_table.p) (void)hipFree(h->ide_table.p);
  drop_graphs(h);
  for (auto& G : h->groups) if (G.done) (void)hipEventDestroy(G.done);
  if (h->cap_stream) (void)hipStreamDestroy(h->cap_stream);
  // side_stream / train_stream belong to the process (rc_helper_stream)
  if (h->sec_sbounds) (void)hipFree(h->sec_sbounds);
  for (hipEvent_t e : h->ev_train) if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : h->ev_side) if (e) (void)hipEventDestroy(e);
  if (h->ev_created)
    for (int s = 0; s < kEvSlots; ++s)
      for (int i = 0; i <= ST_COUNT; ++i) (void)hipEventDestroy(h->ev[s][i]);
  } catch (...) {
  }
  delete h;
}

int rc_load_weights(rc_handle* h, const rc_tensor_desc* descs, int32_t n) {
  RC_TRY
  if (!h) return RC_ERR_INVALID_ARG;
  if (!descs && n > 0) return fail(h, RC_ERR_INVALID_ARG, "rc_load_weights: null descs");
  RC_HIP(h, hipSetDevice(h->device));
  const auto inv = dense_inventory(h->cfg, h->transient ? &h->tcfg : nullptr);
  for (int i = 0; i < n; ++i) {
    const rc_tensor_desc& d = descs[i];
    if (!d.name || !d.data) return fail(h, RC_ERR_INVALID_ARG, "rc_load_weights: null name/data");
    const std::string name = d.name;
    int64_t count = 1;
    for (int k = 0; k < d.ndim; ++k) count *= d.shape[k];
    bool handled = false;
    // grid tables
    for (int g = 0; g < 6 && !handled; ++g) {
      GridState& gs = h->grids[g];
      if (name.compare(0, gs.prefix.size() + 1, gs.prefix + "/") != 0) continue;
      const std::string leaf = name.substr(gs.prefix.size() + 1);
      for (size_t l = 0; l < gs.sizes.size(); ++l) {
        if (leaf != level_name(gs.cfg, gs.sizes, gs.sizes[l])) continue;
        const RcGridLevel& L = gs.dev.lvl[l];
        const int64_t expect = (int64_t)L.entries * gs.cfg.num_features;
        bool shape_ok = count == expect && d.shape[d.ndim - 1] == gs.cfg.num_features;
        if (L.dense) shape_ok = shape_ok && d.ndim == 4 && d.shape[0] == L.size && d.shape[1] == L.size && d.shape[2] == L.size;
        else shape_ok = shape_ok && d.ndim == 2 && d.shape[0] == gs.cfg.hash_map_size;
        if (!shape_ok) return fail(h, RC_ERR_SHAPE, "rc_load_weights: bad shape for " + name);
        DevBuf& b = gs.tables[l];
        const size_t bytes = (size_t)expect * sizeof(float);
        if (!b.p) { RC_HIP(h, hipMalloc((void**)&b.p, bytes)); b.bytes = bytes; }
        RC_HIP(h, hipMemcpy(b.p, d.data, bytes, d.on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
        gs.dev.lvl[l].table = b.p;
        gs.dev.lvl[l].cell = nullptr;    // the derived cell table is stale until the next repack
        gs.dev.lvl[l].rec = nullptr;
        gs.loaded[l] = true;
        h->packed_dirty = true;          // derived device copies (interleaved level-2 tables) follow the tables
        handled = true;
        break;
      }
      if (!handled) return fail(h, RC_ERR_INVALID_ARG, "rc_load_weights: unknown grid level " + name);
    }
    if (handled) continue;
    if (h->transient && name == "params/Cache/Shader/light_power") {     // nerf.py:400-409
      if (count != 1) return fail(h, RC_ERR_SHAPE, "rc_load_weights: bad shape for " + name);
      if (d.on_device) RC_HIP(h, hipMemcpy(&h->light_power, d.data, sizeof(float), hipMemcpyDeviceToHost));
      else memcpy(&h->light_power, d.data, sizeof(float));
      h->have_light_power = true;
      continue;
    }
    // dense layers
    const size_t slash = name.rfind('/');
    if (slash == std::string::npos) return fail(h, RC_ERR_INVALID_ARG, "rc_load_weights: unknown tensor " + name);
    const std::string path = name.substr(0, slash), leaf = name.substr(slash + 1);
    auto it = inv.find(path);
    if (it == inv.end() || (leaf != "kernel" && leaf != "bias"))
      return fail(h, RC_ERR_INVALID_ARG, "rc_load_weights: unknown tensor " + name);
    HostLayer& L = h->layers[path];
    L.in = it->second.first;
    L.out = it->second.second;
    std::vector<float>& dst = leaf == "kernel" ? L.kernel : L.bias;
    if (leaf == "kernel") {
      if (d.ndim != 2 || d.shape[0] != L.in || d.shape[1] != L.out) return fail(h, RC_ERR_SHAPE, "rc_load_weights: bad shape for " + name);
    } else {
      if (d.ndim != 1 || d.shape[0] != L.out) return fail(h, RC_ERR_SHAPE, "rc_load_weights: bad shape for " + name);
    }
    dst.resize((size_t)count);
    if (d.on_device) RC_HIP(h, hipMemcpy(dst.data(), d.data, count * sizeof(float), hipMemcpyDeviceToHost));
    else memcpy(dst.data(), d.data, count * sizeof(float));
    (leaf == "kernel" ? L.have_kernel : L.have_bias) = true;
    h->packed_dirty = true;
    ++h->layers_gen;
  }
  drop_graphs(h);   // table pointers / packed fragments are baked into captured kernel arguments
  return RC_OK;
  RC_CATCH(h)
}

int rc_set_profiling(rc_handle* h, int32_t enabled) {
  RC_TRY
  if (!h) return RC_ERR_INVALID_ARG;
  RC_HIP(h, hipSetDevice(h->device));
  if (enabled && !h->ev_created) {
    for (int s = 0; s < kEvSlots; ++s)
      for (int i = 0; i <= ST_COUNT; ++i) RC_HIP(h, hipEventCreate(&h->ev[s][i]));
    h->ev_created = true;
  }
  if (enabled < 0 || enabled > 3) return fail(h, RC_ERR_INVALID_ARG, "rc_set_profiling: mode must be 0, 1, 2 or 3");
  h->profiling = enabled;
  h->prof_calls = 0;
  for (int s = 0; s < kEvSlots; ++s) h->ev_used[s] = false;
  return RC_OK;
  RC_CATCH(h)
}

int rc_set_fused(rc_handle* h, int32_t mode) {
  RC_TRY
  if (!h) return RC_ERR_INVALID_ARG;
  if (mode < 0 || mode > 3) return fail(h, RC_ERR_INVALID_ARG, "rc_set_fused: mode must be 0, 1, 2 or 3");
  if (mode != h->fused_mode) drop_graphs(h);
  h->fused_mode = mode;
  return RC_OK;
  RC_CATCH(h)
}

int rc_set_graph_mode(rc_handle* h, int32_t mode) {
  RC_TRY
  if (!h) return RC_ERR_INVALID_ARG;
  if (mode < 0 || mode > 2) return fail(h, RC_ERR_INVALID_ARG, "rc_set_graph_mode: mode must be 0, 1 or 2");
  h->graph_mode = mode;
  if (mode == 0) drop_graphs(h);
  return RC_OK;
  RC_CATCH(h)
}

int rc_stage_times_ms(rc_handle* h, float* out_ms, int32_t n) {
  RC_TRY
  if (!h || !out_ms) return RC_ERR_INVALID_ARG;
  if (!h->ev_created) return fail(h, RC_ERR_INVALID_ARG, "rc_stage_times_ms: profiling was not enabled");
  RC_HIP(h, hipSetDevice(h->device));
  // mean over the event sets recorded since profiling was (re)enabled (at most the last kEvSlots calls)
  double sum[ST_COUNT] = {0};
  int used = 0;
  for (int s = 0; s < kEvSlots; ++s) {
    if (!h->ev_used[s]) continue;
    RC_HIP(h, hipEventSynchronize(h->ev[s][h->profiling == 1 ? ST_COUNT : ST_SHADER + 1]));
    for (int i = 0; i < ST_COUNT; ++i) {
      float ms = 0.0f;
      if (h->profiling == 1 || i == ST_SHADER) RC_HIP(h, hipEventElapsedTime(&ms, h->ev[s][i], h->ev[s][i + 1]));
      sum[i] += ms;
    }
    ++used;
  }
  if (!used) return fail(h, RC_ERR_INVALID_ARG, "rc_stage_times_ms: no profiled render call yet");
  for (int i = 0; i < ST_COUNT && i < n; ++i) out_ms[i] = (float)(sum[i] / used);
  return RC_OK;
  RC_CATCH(h)
}

int rc_workspace_ptr(rc_handle* h, const char* name, void** ptr, int64_t* count) {
  RC_TRY
  if (!h || !name || !ptr || !count) return RC_ERR_INVALID_ARG;
  auto it = h->ws.find(name);
  if (it == h->ws.end()) return fail(h, RC_ERR_INVALID_ARG, std::string("rc_workspace_ptr: unknown buffer ") + name);
  *ptr = it->second.p;
  *count = h->ws_count[name];
  return RC_OK;
  RC_CATCH(h)
}

int rc_hashgrid_lookup(rc_handle* h, int32_t grid_id, const float* points, int64_t n, float* features_out,
                       int32_t apply_contraction, void* stream) {
  RC_TRY
  if (!h) return RC_ERR_INVALID_ARG;
  if (grid_id < 0 || grid_id >= 6) return fail(h, RC_ERR_INVALID_ARG, "rc_hashgrid_lookup: bad grid_id");
  if (n < 0 || (n > 0 && (!points || !features_out))) return fail(h, RC_ERR_INVALID_ARG, "rc_hashgrid_lookup: null buffer");
  GridState& gs = h->grids[grid_id];
  for (size_t l = 0; l < gs.sizes.size(); ++l)
    if (!gs.loaded[l]) return fail(h, RC_ERR_MISSING_WEIGHT, "rc_hashgrid_lookup: missing " + gs.prefix + "/" + level_name(gs.cfg, gs.sizes, gs.sizes[l]));
  RC_HIP(h, hipSetDevice(h->device));
  const int LF = gs.dev.num_levels * gs.dev.num_features;
  rc_launch_hashgrid(gs.dev, points, 0, n, features_out, 0, LF, apply_contraction ? h->cfg.contract_radius : 0.0f,
                     nullptr, (hipStream_t)stream);
  RC_HIP(h, hipGetLastError());
  return RC_OK;
  RC_CATCH(h)
}

int rc_sample_intervals(rc_handle* h, const float* t, const float* logits, int64_t n, int32_t num_bins,
                        int32_t num_samples, const float* jitter, float* out, void* stream) {
  RC_TRY
  if (!h) return RC_ERR_INVALID_ARG;
  if (num_bins < 1 || num_bins > 64 || num_samples < 2 || num_samples > 64)
    return fail(h, RC_ERR_UNSUPPORTED, "rc_sample_intervals: bins/samples must be <= 64 (samples >= 2)");
  if (n < 0 || (n > 0 && (!t || !logits || !out))) return fail(h, RC_ERR_INVALID_ARG, "rc_sample_intervals: null buffer");
  RC_HIP(h, hipSetDevice(h->device));
  rc_launch_sample_intervals(t, logits, n, num_bins, num_samples, jitter, out, (hipStream_t)stream);
  RC_HIP(h, hipGetLastError());
  return RC_OK;
  RC_CATCH(h)
}

namespace {

struct RenderArgs {
  rc_rays rays; rc_randoms rnd; bool have_rnd; int64_t n; uint32_t mask; rc_outputs out; int slot; bool fused;
  const rc_transient_outputs* tout = nullptr; const float* cam_origins = nullptr;
  const rc_randoms* shadow_rnd = nullptr; bool weights_only = false; bool force_grad = false;
  bool export_samples = false;     // fused plan: leave tdist / density / means / normals_pred of the last level in the workspace
  const float* s_bounds = nullptr; // secondary rays with ONE (near, far): power-ladder bounds computed once (RcSampleArgs)
  // material stage: the EnvMap of the trace's directions is released on `env_side` when the LAST proposal level is
  // launched, which leaves `env_reserve` CUs to it (both kernels keep a CU's LDS to themselves; see rc_render_material)
  const RcEnvMapArgs* env = nullptr; hipStream_t env_side = nullptr; hipEvent_t env_ready = nullptr, env_done = nullptr;
  int env_reserve = 0;
  bool* env_released = nullptr;    // set when the launch plan taken had the spot (the caller releases the EnvMap itself otherwise)
};

void enqueue_transient_tail(rc_handle* h, const RenderArgs& A, hipStream_t st);

// Enqueue the whole launch sequence on `st` (also used under stream capture).
void enqueue_all(rc_handle* h, const RenderArgs& A, hipStream_t st) {
  const rc_config& c = h->cfg;
  const int NL = c.num_levels;
  const int64_t n = A.n;
  const rc_rays* rays = &A.rays;
  const rc_randoms* rnd = A.have_rnd ? &A.rnd : nullptr;
  const bool secondary = (A.mask & RC_PASS_SECONDARY) != 0;
  const bool resample = secondary || (A.mask & RC_PASS_RESAMPLE);
  // lean: no per-sample consumer of the last level's hidden feature / predicted normals besides the shader
  const bool lean = resample && !A.tout && !A.weights_only && !A.force_grad && !A.out.ptr[RC_OUT_NORMALS_PRED] &&
                    !A.out.ptr[RC_OUT_NORMALS];
  const int slot = A.slot;
  if (A.fused) {
    // everything of the launch that only changes when weights are (re)packed was resolved then (fused_template): no
    // string keys, no map lookups on the per-batch path
    RcFusedLaunch F = h->fused_tmpl;
    F.rays = A.rays; F.n = n;
    for (int l = 0; l < 3; ++l) F.jitter[l] = rnd ? rnd->jitter[l] : nullptr;
    F.out = A.out;
    F.direct = h->fused_direct;
    // mode 3: the one-wavefront-per-ray form.  Builds with the split-MFMA shader (RC_SPLIT_MFMA) run it for mode 1 as well:
    // the two-wave kernel was unstable with the split form in every layer (rc_dev_mlp.h INSTABILITY; cause not found)
    // (RC_TEAM_SPLIT=1 in the environment puts the two-wave kernel back under the split form: the configuration
    // for diagnosis: with the density MLPs fp32, as they are now, tools/stress_repeat.py has not shown a differing launch there)
    static const bool team_split = getenv("RC_TEAM_SPLIT") && getenv("RC_TEAM_SPLIT")[0] == '1';
    F.team = (h->fused_mode == 1 && (!kRcSplit || team_split)) ? 1 : 0;
    F.stagger_cycles = h->fused_stagger;
    F.prio_mode = h->fused_prio;
    if (A.export_samples) {
      const std::string L2 = std::to_string(NL - 1);
      F.export_samples = 1;
      F.f_tdist = W(h, "tdist" + L2); F.f_density = W(h, "density" + L2); F.f_means = W(h, "means" + L2);
      F.f_normals_pred = W(h, "normals_pred");
    }
    // profiling: the single launch is reported as the "shader" stage, every other stage as 0
    for (int i = 0; i <= ST_SHADER; ++i) stage_mark(h, slot, i, st);
    roctx_stage(F.team ? "k_cache_fused_team" : "k_cache_fused");
    rc_launch_fused(F, st);
    for (int i = ST_SHADER + 1; i <= ST_COUNT; ++i) stage_mark(h, slot, i, st);
    return;
  }
  if (A.tout && h->fused_front_ok && h->fused_mode != 0 && !secondary && !resample && !A.weights_only) {
    // time-resolved cache on primary rays: the whole proposal sampler + appearance lookup as ONE launch (the FRONT
    // variant of the fused kernel), results in the workspace buffers the stages behind it read
    const std::string L2 = std::to_string(NL - 1);
    RcFusedLaunch F{};
    F.rays = A.rays; F.n = n;
    for (int l = 0; l < 3; ++l) { F.jitter[l] = rnd ? rnd->jitter[l] : nullptr; F.num_samples[l] = c.num_samples[l]; }
    for (int g = 0; g < 4; ++g) F.grid[g] = &h->grids[g].dev;
    for (int l = 0; l < h->grids[2].dev.num_levels; ++l) F.pair_table[l] = h->packs["pair_" + std::to_string(l)].p;
    for (int g = 0; g < 2; ++g)
      for (int l = 0; l < h->grids[g].dev.num_levels; ++l)
        F.cell_table[g][l] = h->grids[g].dev.lvl[l].dense ? h->packs["cell" + std::to_string(g) + "_" + std::to_string(l)].p
                                                        : h->grids[g].dev.lvl[l].rec;      // cell records of a kLevelHRec level, or NULL
    F.wstream = h->packs["fused_front"].p; F.ide_coef = nullptr;
    F.anneal = c.anneal; F.padding = c.resample_padding; F.density_bias = c.density_bias;
    F.contract_radius = c.contract_radius; F.bg = c.bg_intensity;
    memset(&F.out, 0, sizeof(F.out));
    F.front = 1;
    F.want_grad = (A.out.ptr[RC_OUT_NORMALS] != nullptr || A.force_grad) ? 1 : 0;
    F.use_raydist = 1; F.raydist_p = c.raydist_p; F.raydist_premult = c.raydist_premult;   // TransientNeRFModel (see below)
    F.f_tdist = W(h, "tdist" + L2); F.f_density = W(h, "density" + L2); F.f_means = W(h, "means" + L2);
    F.f_normals_pred = W(h, "normals_pred"); F.f_normals_grad = W(h, "normals_grad"); F.f_hbuf = W(h, "hbuf"); F.f_app = W(h, "app");
    rc_launch_fused(F, st);
    enqueue_transient_tail(h, A, st);
    return;
  }
  for (int l = 0; l < NL; ++l) {
    const std::string L = std::to_string(l), Lp = std::to_string(l - 1);
    const int S = c.num_samples[l];
    const int64_t np = n * S;
    RcSampleArgs sa{};
    sa.origins = rays->origins; sa.directions = rays->directions; sa.viewdirs = rays->viewdirs;
    sa.near = rays->near; sa.far = rays->far; sa.normals = rays->normals; sa.n_rays = n;
    if (l > 0) {
      sa.prev_sdist = W(h, "sdist" + Lp); sa.prev_tdist = W(h, "tdist" + Lp); sa.prev_density = W(h, "density" + Lp);
      sa.P = c.num_samples[l - 1];
      sa.prev_weights = W(h, "weights" + Lp);
    } else {
      sa.P = 1;
    }
    sa.S = S;
    sa.jitter = rnd ? rnd->jitter[l] : nullptr;
    sa.sdist = W(h, "sdist" + L); sa.tdist = W(h, "tdist" + L); sa.means = W(h, "means" + L);
    sa.anneal = c.anneal; sa.padding = c.resample_padding;
    sa.secondary = secondary ? 1 : 0;
    sa.use_raydist = (secondary || h->transient) ? 1 : 0;     // TransientNeRFModel: use_raydist_for_secondary_only = False
    sa.raydist_p = c.raydist_p; sa.raydist_premult = c.raydist_premult;
    sa.eps_dot_min = c.shadow_normal_eps_dot_min; sa.far_clamp = c.env_map_distance;
    sa.s_bounds = (secondary && !rays->normals) ? A.s_bounds : nullptr;
    const bool want_grad = (l == NL - 1) && (A.out.ptr[RC_OUT_NORMALS] != nullptr || A.force_grad);
    // only the density of this level's samples is consumed behind it (proposal levels; the last level on the lean
    // resampling pass): grid lookup + density MLP as ONE launch, weights resident in LDS (rc_level.hip) -- and on the
    // default plan (rc_set_fused 1) the level's sampling in front of it in the same launch, one ray per wave
    const bool level_kernel = h->fused_mode != 0 && !h->profiling && rc_level_supported(h->grids[l].dev) && !want_grad &&
                              (l < NL - 1 || (lean && !A.tout));
    RcLevelArgs la{};
    la.grid = &h->grids[l].dev; la.means = W(h, "means" + L); la.n = np; la.wstream = h->packs["dens_" + L].p;
    la.density_bias = c.density_bias; la.contract_radius = c.contract_radius; la.density = W(h, "density" + L);
    // (one ray per wave pays from ~100 rays per CU on: measured break-even of the whole pass near 32 k rays, +27 us at 1-4 k)
    if (level_kernel && (h->fused_mode == 1 || h->fused_mode == 3) && n >= 24576 && rc_level_ray_supported(h->grids[l].dev, S)) {
      roctx_stage(l == 0 ? "level0 (sample+grid+mlp)" : (l == 1 ? "level1 (sample+grid+mlp)" : "level2 (sample+grid+mlp)"));
      if (A.env && l == NL - 1) {
        (void)hipEventRecord(A.env_ready, st);
        (void)hipStreamWaitEvent(A.env_side, A.env_ready, 0);
        rc_launch_envmap(*A.env, A.env_side);
        (void)hipEventRecord(A.env_done, A.env_side);
        la.cu_reserve = A.env_reserve;
        if (A.env_released) *A.env_released = true;
      }
      rc_launch_level_ray(la, sa, st);
      continue;
    }
    stage_mark(h, slot, ST_SAMPLE0 + 3 * l, st);
    rc_launch_sample(sa, st);

    stage_mark(h, slot, ST_GRID0 + 3 * l, st);
    if (level_kernel) {
      stage_mark(h, slot, ST_MLP0 + 3 * l, st);
      rc_launch_level(la, st);
      continue;
    }
    rc_launch_hashgrid(h->grids[l].dev, W(h, "means" + L), 1, np, W(h, "feat" + L), 1, np, c.contract_radius,
                       want_grad ? W(h, "jac") : nullptr, st);

    RcDensityMlpArgs da{};
    da.feat = W(h, "feat" + L); da.n = np; da.ld = np;
    da.K = h->grids[l].dev.num_levels * h->grids[l].dev.num_features;
    da.wstream = h->packs["dens_" + L].p;
    da.means = W(h, "means" + L);
    da.density_bias = c.density_bias; da.contract_radius = c.contract_radius; da.bbox = h->grids[l].cfg.bbox;
    da.last = (l == NL - 1) ? 1 : 0;
    da.density = W(h, "density" + L);
    // lean resampling pass: the hidden feature / predicted normals are only needed at the ONE sample per ray picked
    // below, so they are not written for all of them here (256 + 12 bytes per sample) but recomputed for the picks
    da.hbuf = (da.last && !lean) ? W(h, "hbuf") : nullptr;
    da.normals_pred = (da.last && !lean) ? W(h, "normals_pred") : nullptr;
    da.jac = want_grad ? W(h, "jac") : nullptr;
    da.normals_grad = want_grad ? W(h, "normals_grad") : nullptr;
    stage_mark(h, slot, ST_MLP0 + 3 * l, st);
    rc_launch_density_mlp(da, st);
  }
  const std::string LL = std::to_string(NL - 1);
  const int S2 = c.num_samples[NL - 1];
  const int64_t np2 = n * S2;
  if (A.weights_only) {
    // weights_only=True (models.py:944-982): no shader; only acc = sum of the last level's weights is consumed
    RcCompositeArgs ca{};
    ca.directions = rays->directions; ca.origins = rays->origins; ca.lights = rays->lights; ca.n_rays = n; ca.S = S2;
    ca.tdist = W(h, "tdist" + LL); ca.density = W(h, "density" + LL); ca.means = W(h, "means" + LL);
    ca.normals_pred = W(h, "normals_pred"); ca.normals_grad = nullptr;
    ca.shade = W(h, "shade"); ca.Sf = S2; ca.weights = W(h, "weights" + LL);
    ca.bg = 0.0f;
    ca.pct[0] = c.percentiles[0]; ca.pct[1] = c.percentiles[1]; ca.pct[2] = c.percentiles[2];
    ca.out = A.out;
    rc_launch_composite(ca, st);
    return;
  }
  stage_mark(h, slot, ST_RESAMPLE, st);
  const int32_t* src = nullptr;
  if (resample) {
    RcResampleArgs ra{};
    ra.n_rays = n; ra.S = S2; ra.tdist = W(h, "tdist" + LL); ra.density = W(h, "density" + LL);
    ra.directions = rays->directions; ra.gumbel = rnd->gumbel; ra.inds_in = rnd->resample_inds;
    ra.inds_out = (int32_t*)W(h, "inds"); ra.filt_weight = W(h, "filt_weight"); ra.weights = W(h, "weights" + LL);
    ra.acc_out = W(h, "acc_sel");
    ra.src_out = (int32_t*)W(h, "src_idx");
    rc_launch_resample(ra, st);
    src = (const int32_t*)W(h, "src_idx");
  }
  const int64_t nsh = resample ? n : np2;
  bool pair_lookup = false;
  if (lean) {
    // density MLP of the last level once more, on the picked samples only (same kernel, same per-point arithmetic:
    // bitwise the values the full pass would have stored), compact outputs
    const int l2 = NL - 1;
    const int K2 = h->grids[l2].dev.num_levels * h->grids[l2].dev.num_features;
    // with the fused plan's interleaved tables at hand: this lookup and the appearance lookup below in one pass
    pair_lookup = h->fused_ok && l2 == 2 && nsh == n && h->fused_mode != 0;
    if (pair_lookup) {
      RcPairTables pt{};
      for (int l = 0; l < h->grids[2].dev.num_levels; ++l) pt.t[l] = h->fused_tmpl.pair_table[l];
      rc_launch_hashgrid_pair(h->grids[2].dev, pt, W(h, "means" + LL), src, np2, n, W(h, "feat_sel"), W(h, "app"), n, c.contract_radius, st);
    } else {
      rc_launch_hashgrid_src(h->grids[l2].dev, W(h, "means" + LL), 1, src, np2, n, W(h, "feat_sel"), 1, n, c.contract_radius,
                             nullptr, st);
    }
    RcDensityMlpArgs ds{};
    ds.feat = W(h, "feat_sel"); ds.n = n; ds.ld = n; ds.K = K2; ds.wstream = h->packs["dens_" + LL].p;
    ds.means = W(h, "means" + LL); ds.src = src; ds.n_src = np2;
    ds.density_bias = c.density_bias; ds.contract_radius = c.contract_radius; ds.bbox = h->grids[l2].cfg.bbox;
    ds.last = 1; ds.density = W(h, "density_sel"); ds.hbuf = W(h, "hbuf_sel"); ds.normals_pred = W(h, "normals_sel");
    rc_launch_density_mlp(ds, st);
  }
  stage_mark(h, slot, ST_GRID_APP, st);
  if (!pair_lookup)
    rc_launch_hashgrid_src(h->grids[3].dev, W(h, "means" + LL), 1, src, np2, nsh, W(h, "app"), 1, nsh,
                           c.contract_radius, nullptr, st);
  if (A.tout) {
    enqueue_transient_tail(h, A, st);
    return;
  }
  stage_mark(h, slot, ST_SHADER, st);
  {
    RcShaderArgs s{};
    s.n = nsh; s.n_src = lean ? n : np2; s.src = lean ? nullptr : src; s.samples_per_ray = resample ? 1 : S2;
    s.hbuf = W(h, lean ? "hbuf_sel" : "hbuf"); s.app = W(h, "app");
    s.normals_pred = W(h, lean ? "normals_sel" : "normals_pred"); s.viewdirs = rays->viewdirs;
    s.wstream = h->packs["shader"].p; s.ide_coef = h->ide_table.p;
    s.roughness_bias = c.roughness_bias; s.irradiance_bias = c.irradiance_bias; s.ambient_bias = c.ambient_irradiance_bias;
    s.rgb_max = c.rgb_max; s.slf_ambient_bias = c.slf_ambient_bias;
    s.shade = W(h, "shade");
    s.debug = W(h, "debug");
    rc_launch_shader(s, st);
  }
  stage_mark(h, slot, ST_COMPOSITE, st);
  {
    RcCompositeArgs ca{};
    ca.directions = rays->directions; ca.origins = rays->origins; ca.lights = rays->lights; ca.n_rays = n; ca.S = S2;
    ca.tdist = W(h, "tdist" + LL); ca.density = W(h, "density" + LL); ca.means = W(h, "means" + LL);
    ca.normals_pred = lean ? nullptr : W(h, "normals_pred");
    ca.normals_grad = A.out.ptr[RC_OUT_NORMALS] ? W(h, "normals_grad") : nullptr;
    ca.shade = W(h, "shade"); ca.Sf = resample ? 1 : S2;
    ca.inds = resample ? (const int32_t*)W(h, "inds") : nullptr;
    ca.filt_weight = resample ? W(h, "filt_weight") : nullptr;
    ca.weights = W(h, "weights" + LL);
    ca.bg = secondary ? 0.0f : c.bg_intensity;
    ca.pct[0] = c.percentiles[0]; ca.pct[1] = c.percentiles[1]; ca.pct[2] = c.percentiles[2];
    ca.out = A.out;
    if (secondary) {
      ca.out.ptr[RC_OUT_RGB] = W(h, "rgb_noenv");
      ca.out.ptr[RC_OUT_ACC] = W(h, "acc_ws");
    }
    // one resampled sample per ray and nothing but rgb / acc wanted (the batched secondary trace): the weights and their
    // sum are k_resample's, one thread per ray finishes (17 -> 3 us for 32 768 rays; same bits)
    bool pick_only = resample;
    for (int id = 0; id < RC_OUT_COUNT && pick_only; ++id)
      if (ca.out.ptr[id] && id != RC_OUT_RGB && id != RC_OUT_ACC) pick_only = false;
    if (pick_only) rc_launch_composite_pick(n, W(h, "shade"), W(h, "filt_weight"), W(h, "acc_sel"), ca.bg, ca.out.ptr[RC_OUT_RGB], ca.out.ptr[RC_OUT_ACC], st);
    else rc_launch_composite(ca, st);
  }
  if (secondary) {
    const bool use_env = !(A.mask & RC_PASS_NO_ENVMAP);
    if (use_env) {
      RcEnvMapArgs ea{};
      ea.n = n; ea.viewdirs = rays->viewdirs; ea.wstream = h->packs["envmap"].p;
      ea.rgb_bias = c.env_rgb_bias; ea.env_rgb = W(h, "env_rgb");
      rc_launch_envmap(ea, st);
    }
    hipLaunchKernelGGL(k_env_combine, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n,
                       (const float*)W(h, "rgb_noenv"), (const float*)W(h, "acc_ws"),
                       use_env ? (const float*)W(h, "env_rgb") : (const float*)nullptr, A.out.ptr[RC_OUT_RGB],
                       A.out.ptr[RC_OUT_ACC], A.out.ptr[RC_OUT_ENV_MAP_RGB], A.out.ptr[RC_OUT_RGB_NO_ENV]);
  }
  stage_mark(h, slot, ST_COUNT, st);
}

}  // namespace

int rc_render_rays(rc_handle* h, const rc_rays* rays, int64_t n, const rc_randoms* rnd, uint32_t pass_mask,
                   const rc_outputs* out, void* stream_v) {
  RC_TRY
  if (!h) return RC_ERR_INVALID_ARG;
  if (!rays || !out) return fail(h, RC_ERR_INVALID_ARG, "rc_render_rays: null rays/outputs");
  if (n < 0) return fail(h, RC_ERR_INVALID_ARG, "rc_render_rays: negative n_rays");
  if (n == 0) return RC_OK;
  if (!rays->origins || !rays->directions || !rays->viewdirs || !rays->near || !rays->far)
    return fail(h, RC_ERR_INVALID_ARG, "rc_render_rays: origins/directions/viewdirs/near/far are required");
  RoctxScope roctx_call("rc_render_rays");
  if (h->transient) return fail(h, RC_ERR_UNSUPPORTED, "rc_render_rays: this handle renders the time-resolved cache (rc_render_transient)");
  if (!(pass_mask & RC_PASS_CACHE)) return fail(h, RC_ERR_UNSUPPORTED, "rc_render_rays: pass_mask must include RC_PASS_CACHE");
  const bool secondary = (pass_mask & RC_PASS_SECONDARY) != 0;
  const bool resample = secondary || (pass_mask & RC_PASS_RESAMPLE);
  if (resample && h->cfg.num_resample != 1) return fail(h, RC_ERR_UNSUPPORTED, "rc_render_rays: num_resample must be 1");
  if (resample && !(rnd && (rnd->gumbel || rnd->resample_inds)))
    return fail(h, RC_ERR_INVALID_ARG, "rc_render_rays: resampling needs rc_randoms.gumbel or .resample_inds");
  RC_HIP(h, hipSetDevice(h->device));
  hipStream_t st = (hipStream_t)stream_v;
  int rc;
  if (h->packed_dirty) {
    drop_graphs(h);
    if ((rc = repack(h))) return rc;
  }
  if (secondary && !(pass_mask & RC_PASS_NO_ENVMAP) && !h->have_envmap)
    return fail(h, RC_ERR_MISSING_WEIGHT, "missing weight: params/Cache/EnvMap/* (secondary rays composite the model-level EnvMap)");
  const bool fused = (h->fused_mode == 1 || h->fused_mode == 3) && h->fused_ok && pass_mask == RC_PASS_CACHE;
  // one workspace set per caller stream (up to 4): calls on different streams do not share buffers.  The fused kernel
  // keeps every intermediate on chip: no workspace, no set.
  const int ws_slot = fused ? 0 : ws_pick(h, st);
  struct PrefixGuard {
    rc_handle* h;
    ~PrefixGuard() { h->ws_prefix = ""; }
  } guard{h};
  h->ws_prefix = ws_slot == 0 ? "" : "p" + std::to_string(ws_slot) + ":";
  if (!fused) {
    if ((rc = ws_enter(h, ws_slot, st))) return rc;
    if ((rc = ensure_workspace(h, n))) return rc;      // a reallocation drops the captured graphs (ws_alloc)
  }
  struct LeaveGuard {                                  // every exit below records the set's "done" event
    rc_handle* h; int slot; hipStream_t st; bool on;
    ~LeaveGuard() { if (on) (void)ws_leave(h, slot, st); }
  } leave{h, ws_slot, st, !fused};
  rc_shader_prepare();

  RenderArgs A{};
  A.rays = *rays;
  A.have_rnd = rnd != nullptr;
  if (rnd) A.rnd = *rnd;
  A.n = n; A.mask = pass_mask; A.out = *out;
  A.fused = fused;
  A.slot = -1;
  if (h->profiling) {
    // mode 3: events around the dominant kernel of every 8th call only (an event record costs ~1.3 us of stream time)
    const int64_t call = h->prof_calls++;
    if (h->profiling != 3 || call % 8 == 0) {
      A.slot = (int)((h->profiling == 3 ? call / 8 : call) % kEvSlots);
      h->ev_used[A.slot] = true;
    }
  }

  // event records are not replayable graph nodes: profile eagerly.  The fused plan is ONE launch: a plain launch
  // queues back to back with the previous one (gap < 1 us), the replay of a one-node graph leaves ~5 us between them.
  if (h->graph_mode == 0 || h->profiling || fused) {
    enqueue_all(h, A, st);
    RC_HIP(h, hipGetLastError());
    return RC_OK;
  }
  RenderKey key;
  memset(&key, 0, sizeof(key));
  key.n = n; key.mask = pass_mask; key.slot = A.slot; key.ws_slot = ws_slot;
  const void* rp[7] = {rays->origins, rays->directions, rays->viewdirs, rays->near, rays->far, rays->lights, rays->normals};
  memcpy(key.rays, rp, sizeof(rp));
  if (rnd) {
    for (int l = 0; l < RC_MAX_LEVELS; ++l) key.rnd[l] = rnd->jitter[l];
    key.rnd[RC_MAX_LEVELS] = rnd->gumbel; key.rnd[RC_MAX_LEVELS + 1] = rnd->resample_inds;
  }
  for (int i = 0; i < RC_OUT_COUNT; ++i) key.out[i] = out->ptr[i];
  for (auto& g : h->graphs)
    if (g.key == key) {
      RC_HIP(h, hipGraphLaunch(g.exec, st));
      return RC_OK;
    }
  const bool capture = h->graph_mode == 2 || (h->have_last_key && h->last_key == key);
  h->last_key = key;
  h->have_last_key = true;
  if (!capture) {
    enqueue_all(h, A, st);
    RC_HIP(h, hipGetLastError());
    return RC_OK;
  }
  // capture on a private stream (the caller's may be the legacy default stream), replay on the caller's
  if (!h->cap_stream) RC_HIP(h, hipStreamCreateWithFlags(&h->cap_stream, hipStreamNonBlocking));
  if (h->graphs.size() >= 64) drop_graphs(h);
  GraphEntry e;
  e.key = key;
  RC_HIP(h, hipStreamBeginCapture(h->cap_stream, hipStreamCaptureModeThreadLocal));
  enqueue_all(h, A, h->cap_stream);
  hipError_t ce = hipStreamEndCapture(h->cap_stream, &e.graph);
  if (ce != hipSuccess || !e.graph) {
    // capture unsupported for this sequence: fall back to eager launches for good
    (void)hipGetLastError();
    h->graph_mode = 0;
    enqueue_all(h, A, st);
    RC_HIP(h, hipGetLastError());
    return RC_OK;
  }
  RC_HIP(h, hipGraphInstantiate(&e.exec, e.graph, nullptr, nullptr, 0));
  h->graphs.push_back(e);
  RC_HIP(h, hipGraphLaunch(e.exec, st));
  return RC_OK;
  RC_CATCH(h)
}

namespace {
// RCCL entry points, resolved from the instance the process already has (torch ships its own librccl.so: linking a
// second copy into this library would split the communicator state between two instances).
struct RcclApi {
  int (*all_gather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
  int (*group_start)() = nullptr;
  int (*group_end)() = nullptr;
  const char* (*error_string)(int) = nullptr;
  bool tried = false;
};
RcclApi g_rccl;
bool load_rccl() {
  if (g_rccl.tried) return g_rccl.all_gather != nullptr;
  g_rccl.tried = true;
  void* lib = nullptr;
  const char* env = getenv("RC_RCCL_LIBRARY");
  if (env && *env) lib = dlopen(env, RTLD_NOW | RTLD_GLOBAL);
  for (const char* nm : {"librccl.so", "librccl.so.1"}) if (!lib) lib = dlopen(nm, RTLD_NOW | RTLD_NOLOAD);
  for (const char* nm : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) if (!lib) lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
  if (!lib) return false;
  g_rccl.all_gather = (decltype(g_rccl.all_gather))dlsym(lib, "ncclAllGather");
  g_rccl.group_start = (decltype(g_rccl.group_start))dlsym(lib, "ncclGroupStart");
  g_rccl.group_end = (decltype(g_rccl.group_end))dlsym(lib, "ncclGroupEnd");
  g_rccl.error_string = (decltype(g_rccl.error_string))dlsym(lib, "ncclGetErrorString");
  if (!g_rccl.group_start || !g_rccl.group_end) g_rccl.all_gather = nullptr;
  return g_rccl.all_gather != nullptr;
}
const int kOutWidth[RC_OUT_COUNT] = {3, 1, 1, 1, 1, 1, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 1, 1, 3, 3};
}  // namespace

int rc_allgather_outputs(rc_handle* h, void* nccl_comm, const rc_outputs* local, int64_t n_local, const rc_outputs* full,
                         void* stream_v) {
  RC_TRY
  if (!h) return RC_ERR_INVALID_ARG;
  if (!nccl_comm || !local || !full) return fail(h, RC_ERR_INVALID_ARG, "rc_allgather_outputs: null argument");
  if (n_local < 0) return fail(h, RC_ERR_INVALID_ARG, "rc_allgather_outputs: negative n_local");
  if (n_local == 0) return RC_OK;
  if (!load_rccl()) return fail(h, RC_ERR_UNSUPPORTED, "rc_allgather_outputs: no RCCL in this process (librccl.so / RC_RCCL_LIBRARY)");
  RC_HIP(h, hipSetDevice(h->device));
  const int kNcclFloat = 7;                                   // ncclFloat32 (rccl.h)
  int rc = g_rccl.group_start();
  for (int i = 0; i < RC_OUT_COUNT && rc == 0; ++i)
    if (local->ptr[i] && full->ptr[i])
      rc = g_rccl.all_gather(local->ptr[i], full->ptr[i], (size_t)n_local * kOutWidth[i], kNcclFloat, nccl_comm, (hipStream_t)stream_v);
  const int rc_end = g_rccl.group_end();
  if (rc == 0) rc = rc_end;
  if (rc != 0) return fail(h, RC_ERR_HIP, std::string("rc_allgather_outputs: RCCL: ") + (g_rccl.error_string ? g_rccl.error_string(rc) : "error"));
  return RC_OK;
  RC_CATCH(h)
}

// The chunk loop of render_image in native code: n_chunks x rc_render_rays on alternating streams (include/rc_abi.h).
int rc_render_chunks(rc_handle* h, const rc_rays* rays, int64_t chunk, int64_t n_chunks, uint32_t pass_mask,
                     const rc_outputs* out0, int64_t out_stride, void* const* streams, int32_t n_streams) {
  if (!h) return RC_ERR_INVALID_ARG;
  if (!rays || !out0 || !streams) return fail(h, RC_ERR_INVALID_ARG, "rc_render_chunks: null rays/outputs/streams");
  if (chunk <= 0 || n_chunks < 0 || n_streams <= 0 || out_stride < 0)
    return fail(h, RC_ERR_INVALID_ARG, "rc_render_chunks: chunk, n_chunks, n_streams, out_stride out of range");
  if ((pass_mask & RC_PASS_SECONDARY) || (pass_mask & RC_PASS_RESAMPLE))
    return fail(h, RC_ERR_UNSUPPORTED, "rc_render_chunks: passes that draw random numbers go through rc_render_rays per chunk");
  for (int64_t i = 0; i < n_chunks; ++i) {
    rc_rays r = *rays;
    const int64_t o = i * chunk;
    auto adv = [&](const float*& p, int w) { if (p) p += o * w; };
    adv(r.origins, 3); adv(r.directions, 3); adv(r.viewdirs, 3); adv(r.near, 1); adv(r.far, 1); adv(r.lights, 3); adv(r.normals, 3);
    rc_outputs out = *out0;
    for (int k = 0; k < RC_OUT_COUNT; ++k)
      if (out.ptr[k]) out.ptr[k] += i * out_stride;
    const int rc = rc_render_rays(h, &r, chunk, nullptr, pass_mask, &out, streams[i % n_streams]);
    if (rc) return rc;
  }
  return RC_OK;
}

int rc_render_material(rc_handle* h, const rc_rays* rays, int64_t n, const rc_randoms* rnd,
                       const rc_material_randoms* mr, int32_t K, const rc_outputs* cache_out,
                       const rc_mat_outputs* mat_out, void* stream_v) {
  RC_TRY
  if (!h) return RC_ERR_INVALID_ARG;
  RoctxScope roctx_call("rc_render_material");
  if (h->transient) return fail(h, RC_ERR_UNSUPPORTED, "rc_render_material: this handle renders the time-resolved cache (rc_render_transient)");
  if (!rays || !mr || !cache_out || !mat_out) return fail(h, RC_ERR_INVALID_ARG, "rc_render_material: null argument");
  if (n < 0) return fail(h, RC_ERR_INVALID_ARG, "rc_render_material: negative n_rays");
  if (n == 0) return RC_OK;
  if (!rays->origins || !rays->directions || !rays->viewdirs || !rays->near || !rays->far)
    return fail(h, RC_ERR_INVALID_ARG, "rc_render_material: origins/directions/viewdirs/near/far are required");
  const rc_config& c = h->cfg;
  const int Ks = (int)lround(K * (1.0 - (double)c.diffuse_sample_fraction));
  const int Kd = (int)lround(K * (double)c.diffuse_sample_fraction);
  const int Kc = (int)lround(0.5 * Kd);
  if (K < 2 || Ks < 1 || Kd < 2 || Kc < 1 || Kd - Kc < 1 || Ks + Kd > 64)
    return fail(h, RC_ERR_UNSUPPORTED, "rc_render_material: num_secondary_samples must give 1 <= Ks, 2 <= Kd, Ks + Kd <= 64");
  if (c.num_vmf != 128) return fail(h, RC_ERR_UNSUPPORTED, "rc_render_material: num_vmf must be 128");
  if (!mr->vmf_noise || !mr->spec_u1 || !mr->spec_u2 || !mr->cos_u1 || !mr->cos_u2 || !mr->vmf_v || !mr->vmf_tmp ||
      !(mr->vmf_lobe || mr->vmf_lobe_gumbel))
    return fail(h, RC_ERR_INVALID_ARG, "rc_render_material: every sampler member of rc_material_randoms is required "
                                       "(vmf_lobe or vmf_lobe_gumbel)");
  if (!(mr->gumbel || mr->resample_inds) || !(mr->sec_gumbel || mr->sec_resample_inds))
    return fail(h, RC_ERR_INVALID_ARG, "rc_render_material: the categorical picks need gumbel or resample_inds (primary) and "
                                       "sec_gumbel or sec_resample_inds (secondary trace)");
  RC_HIP(h, hipSetDevice(h->device));
  hipStream_t st = (hipStream_t)stream_v;
  int rc;
  if (h->packed_dirty) {
    drop_graphs(h);
    if ((rc = repack(h))) return rc;
  }
  if (!h->have_material) return fail(h, RC_ERR_MISSING_WEIGHT, "missing weight: params/MaterialShader/* or params/LightSampler/*");
  if (!h->have_envmap) return fail(h, RC_ERR_MISSING_WEIGHT, "missing weight: params/Cache/EnvMap/*");
  const int NL = c.num_levels;
  const int S2 = c.num_samples[NL - 1];
  const int64_t np2 = n * S2, nsec = n * (Ks + Kd);
  h->ws_prefix = "";
  if ((rc = ws_enter(h, 0, st))) return rc;             // shares workspace set 0 (and "s:") with the other entry points
  struct LeaveGuard {
    rc_handle* h; hipStream_t st;
    ~LeaveGuard() { (void)ws_leave(h, 0, st); }
  } leave{h, st};
  if ((rc = ensure_workspace(h, n))) return rc;
  if ((rc = ws_alloc(h, "m_pts", 3 * n)) || (rc = ws_alloc(h, "m_nrm", 3 * n)) || (rc = ws_alloc(h, "m_feat", 32 * n)) ||
      (rc = ws_alloc(h, "m_mat", RC_MAT_CH * n)) || (rc = ws_alloc(h, "m_feat_all", 32 * np2)) ||
      (rc = ws_alloc(h, "m_mat_all", RC_MAT_CH * np2)) || (rc = ws_alloc(h, "l_feat", 32 * n)) ||
      (rc = ws_alloc(h, "l_vmf", (int64_t)128 * RC_VMF_CH * n)) || (rc = ws_alloc(h, "l_vmf_logit", (int64_t)128 * n)) || (rc = ws_alloc(h, "sec_origins", 3 * nsec)) ||
      (rc = ws_alloc(h, "sec_dirs", 3 * nsec)) || (rc = ws_alloc(h, "sec_near", nsec)) || (rc = ws_alloc(h, "sec_far", nsec)) ||
      (rc = ws_alloc(h, "sec_lights", 3 * nsec)) || (rc = ws_alloc(h, "sec_samples", RC_SMP_CH * nsec)) ||
      (rc = ws_alloc(h, "m_local_view", 3 * n)) || (rc = ws_alloc(h, "sec_rgb", 3 * nsec)) ||
      (rc = ws_alloc(h, "sec_acc", nsec)) || (rc = ws_alloc(h, "sec_env", 3 * nsec)))
    return rc;
  h->ws_prefix = "s:";
  rc = ensure_workspace(h, nsec);
  h->ws_prefix = "";
  if (rc) return rc;
  rc_shader_prepare();
  const std::string LL = std::to_string(NL - 1);

  // 1. cache pass on the primary rays (all samples shaded) -> cache_out
  RenderArgs A{};
  A.rays = *rays; A.have_rnd = rnd != nullptr; if (rnd) A.rnd = *rnd;
  A.n = n; A.mask = RC_PASS_CACHE; A.out = *cache_out; A.slot = -1;
  // the fused kernel when the handle runs it (rc_set_fused mode 1): one launch, its per-sample results exported for
  // steps 2-3 below; bitwise the launch-per-stage pass (tests/test_gpu_parity.py)
  A.fused = (h->fused_mode == 1 || h->fused_mode == 3) && h->fused_ok && !h->profiling && c.num_samples[NL - 1] == 32;
  A.export_samples = A.fused;
  enqueue_all(h, A, st);

  // 2. one shading sample per ray (MaterialModel.resample_render, models.py:1430-1439)
  {
    RcResampleArgs ra{};
    ra.n_rays = n; ra.S = S2; ra.tdist = W(h, "tdist" + LL); ra.density = W(h, "density" + LL);
    ra.directions = rays->directions; ra.gumbel = mr->gumbel; ra.inds_in = mr->resample_inds;
    ra.inds_out = (int32_t*)W(h, "inds"); ra.filt_weight = W(h, "filt_weight"); ra.weights = W(h, "weights" + LL);
    ra.src_out = (int32_t*)W(h, "src_idx");
    // position and predicted normal of the picked sample = the shading point
    ra.means = W(h, "means" + LL); ra.normals = W(h, "normals_pred"); ra.pts_out = W(h, "m_pts"); ra.nrm_out = W(h, "m_nrm");
    rc_launch_resample(ra, st);
  }
  auto raw = [&](const char* path, const char* leaf) { return h->packs[std::string("raw:") + path + "/" + leaf].p; };
  // Side stream: the material-only composite over all samples (step 3b) and the EnvMap along the secondary rays (step 6b)
  // feed only the outputs / the final integration, so they leave the critical path sampler -> trace -> integrate and
  // fill the gaps of its latency-bound kernels.  Forked from and joined to the caller's stream with events: to the
  // caller the call is still ordered on `st` alone.
  { int rcs = ensure_side_stream(h); if (rcs) return rcs; }
  hipStream_t side = h->side_stream;
  // Whatever way this function is left once work has been forked onto the side stream, the caller's stream is joined
  // to it again BEFORE the workspace set is released (declared after LeaveGuard: destroyed first): an early error
  // return must not leave side-stream kernels writing mat_out / sec_env behind a caller that believes `st` orders
  // everything of the call.
  struct SideJoin {
    rc_handle* h; hipStream_t st, side; bool forked;
    ~SideJoin() {
      if (!forked) return;
      if (hipEventRecord(h->ev_side[2], side) != hipSuccess || hipStreamWaitEvent(st, h->ev_side[2], 0) != hipSuccess)
        (void)hipStreamSynchronize(side);
    }
  } side_join{h, st, side, false};
  // 3a. material head and 4. light sampler (128 vMF lobes) at the shading point: one lookup launch + one head launch on the
  // caller's stream
  roctx_stage("material: heads + light sampler");
  {
    RcMatHeadArgs ma{};
    ma.w0 = raw("params/MaterialShader/bottleneck_layer", "kernel"); ma.b0 = raw("params/MaterialShader/bottleneck_layer", "bias");
    ma.w1 = raw("params/MaterialShader/pred_brdf_layer", "kernel"); ma.b1 = raw("params/MaterialShader/pred_brdf_layer", "bias");
    ma.min_roughness = c.min_roughness;
    RcLightHeadArgs la{};
    la.n = n; la.feat = W(h, "l_feat");
    la.w0 = raw("params/LightSampler/layers_0", "kernel"); la.b0 = raw("params/LightSampler/layers_0", "bias");
    la.w1 = raw("params/LightSampler/layers_1", "kernel"); la.b1 = raw("params/LightSampler/layers_1", "bias");
    la.w2 = raw("params/LightSampler/output_layer", "kernel"); la.b2 = raw("params/LightSampler/output_layer", "bias");
    la.pts = W(h, "m_pts"); la.noise = mr->vmf_noise; la.vmf_scale = c.vmf_scale; la.vmf = W(h, "l_vmf"); la.vmf_logit = W(h, "l_vmf_logit");
    // caller's stream, the critical path: both grids at the shading points in one launch, both heads in one launch (as
    // four launches on two streams the BRDF sampler waited ~12 us for the event behind the light head)
    rc_launch_hashgrid_two(h->grids[4].dev, h->grids[5].dev, W(h, "m_pts"), n, W(h, "m_feat"), W(h, "l_feat"), c.contract_radius, st);
    ma.n = n; ma.feat = W(h, "m_feat"); ma.mat = W(h, "m_mat");
    rc_launch_shading_heads(ma, la, st);
  }
  // 5. BRDF importance sampling -> secondary rays
  roctx_stage("material: brdf sample");
  {
    RcBrdfSampleArgs sa{};
    sa.n = n; sa.Ks = Ks; sa.Kd = Kd; sa.Kc = Kc;
    sa.pts = W(h, "m_pts"); sa.nrm = W(h, "m_nrm"); sa.viewdirs = rays->viewdirs; sa.lights = rays->lights;
    sa.mat = W(h, "m_mat"); sa.vmf = W(h, "l_vmf"); sa.vmf_logit = W(h, "l_vmf_logit");
    sa.spec_u1 = mr->spec_u1; sa.spec_u2 = mr->spec_u2; sa.cos_u1 = mr->cos_u1; sa.cos_u2 = mr->cos_u2;
    sa.vmf_lobe = mr->vmf_lobe; sa.vmf_v = mr->vmf_v; sa.vmf_tmp = mr->vmf_tmp; sa.vmf_lobe_gumbel = mr->vmf_lobe_gumbel;
    sa.normal_eps = c.secondary_normal_eps; sa.near = c.secondary_near; sa.far = c.secondary_far;
    sa.sec_origins = W(h, "sec_origins"); sa.sec_dirs = W(h, "sec_dirs"); sa.sec_near = W(h, "sec_near");
    sa.sec_far = W(h, "sec_far"); sa.sec_lights = W(h, "sec_lights"); sa.samples = W(h, "sec_samples");
    sa.local_view = W(h, "m_local_view");
    rc_launch_brdf_sample(sa, st);
  }
  // 3b. side stream: the material head on ALL samples and the material-only composite (outputs only).  Forked HERE, behind
  // the BRDF sampler: beside the small latency-bound kernels above they doubled those kernels' times (heads 18 -> 35 us,
  // sampler 19 -> 33 us); beside the first level of the trace they fit into what its workgroups leave of a CU (no LDS /
  // 11 KB) and cost it little.
  {
    RC_HIP(h, hipEventRecord(h->ev_side[0], st));
    RC_HIP(h, hipStreamWaitEvent(side, h->ev_side[0], 0));
    side_join.forked = true;
    RcMatHeadArgs ma{};
    ma.w0 = raw("params/MaterialShader/bottleneck_layer", "kernel"); ma.b0 = raw("params/MaterialShader/bottleneck_layer", "bias");
    ma.w1 = raw("params/MaterialShader/pred_brdf_layer", "kernel"); ma.b1 = raw("params/MaterialShader/pred_brdf_layer", "bias");
    ma.min_roughness = c.min_roughness;
    rc_launch_hashgrid(h->grids[4].dev, W(h, "means" + LL), 1, np2, W(h, "m_feat_all"), 0, 32, c.contract_radius, nullptr, side);
    ma.n = np2; ma.feat = W(h, "m_feat_all"); ma.mat = W(h, "m_mat_all");
    rc_launch_material_head(ma, side);
    rc_launch_material_composite_all(n, S2, W(h, "weights" + LL), W(h, "m_mat_all"), mat_out->ptr[RC_MOUT_MATERIAL_ALBEDO],
                                     mat_out->ptr[RC_MOUT_MATERIAL_ROUGHNESS], mat_out->ptr[RC_MOUT_MATERIAL_METALNESS],
                                     mat_out->ptr[RC_MOUT_MATERIAL_F_0], c.default_F_0, side);
  }
  // 6. ONE batched secondary trace through the cache (is_secondary, resample, use_env_map=False;
  //    ref_rays.normals = None since MaterialMLP.shadow_eps_indirect = False) + EnvMap along the same rays
  {
    RenderArgs B{};
    B.rays.origins = W(h, "sec_origins"); B.rays.directions = W(h, "sec_dirs"); B.rays.viewdirs = W(h, "sec_dirs");
    B.rays.near = W(h, "sec_near"); B.rays.far = W(h, "sec_far"); B.rays.lights = W(h, "sec_lights"); B.rays.normals = nullptr;
    B.have_rnd = true;
    for (int l = 0; l < RC_MAX_LEVELS; ++l) B.rnd.jitter[l] = mr->sec_jitter[l];
    B.rnd.gumbel = mr->sec_gumbel; B.rnd.resample_inds = mr->sec_resample_inds;
    B.n = nsec; B.mask = RC_PASS_CACHE | RC_PASS_SECONDARY | RC_PASS_NO_ENVMAP; B.slot = -1;
    // every secondary ray of this trace has the same (near, far) (k_brdf_sample writes the two constants): the
    // power-ladder image of the pair is computed once instead of by each of the 3 x 32 768 sampler waves
    // ... and once per handle: the five inputs are constants of the configuration (same device function, a buffer of
    // the handle's own; 5 us + a 6 us gap in every step before)
    if (!h->sec_sbounds) {
      RC_HIP(h, hipMalloc((void**)&h->sec_sbounds, 2 * sizeof(float)));
      rc_launch_ladder_bounds(c.secondary_near, c.secondary_far, c.env_map_distance, c.raydist_p, c.raydist_premult, h->sec_sbounds, st);
      RC_HIP(h, hipStreamSynchronize(st));       // later calls may come on other streams
    }
    B.s_bounds = h->sec_sbounds;
    memset(&B.out, 0, sizeof(B.out));
    B.out.ptr[RC_OUT_RGB] = W(h, "sec_rgb"); B.out.ptr[RC_OUT_ACC] = W(h, "sec_acc");
    float* sec_dirs = W(h, "sec_dirs"); float* sec_env = W(h, "sec_env");
    RcEnvMapArgs ea{};
    ea.n = nsec; ea.viewdirs = sec_dirs; ea.wstream = h->packs["envmap"].p; ea.rgb_bias = c.env_rgb_bias; ea.env_rgb = sec_env;
    // 6b. EnvMap of the secondary directions, on the side stream.  It is MFMA-bound and keeps a CU's LDS to itself, as the
    // level kernels of the trace do: next to a level kernel it only gets the CUs that kernel's persistent workgroups
    // leave.  Beside the first two levels (F = 1: bound by their own instruction issue) every CU it takes is a CU they
    // miss; the LAST level (F = 4) is bound by the fabric's random-sector rate, not by CUs.  So the EnvMap is released when
    // that level is launched, and that launch leaves it a quarter of the CUs (RC_ENV_RESERVE overrides; 0 = beside the
    // first level as before): 1.48 -> 1.45 ms per step (reserve 32 / 64 / 96 of 256: 1.463 / 1.447-1.456 / 1.508).
    // (split-MFMA build: the EnvMap takes 0.7 of the time on the bf16 pipe and an eighth of the CUs is enough -- 16 ... 40 of
    // 256 measure 1.315-1.33 ms per step, 64: 1.36, 80: 1.38)
    static const int env_reserve = getenv("RC_ENV_RESERVE") ? atoi(getenv("RC_ENV_RESERVE")) : rc_device_cus() / (kRcSplit ? 8 : 4);
    const bool beside_last = env_reserve > 0 && nsec >= 24576 && (h->fused_mode == 1 || h->fused_mode == 3);
    bool env_released = false;
    if (beside_last) {
      B.env = &ea; B.env_side = side; B.env_ready = h->ev_side[1]; B.env_done = h->ev_side[2]; B.env_reserve = env_reserve;
      B.env_released = &env_released;
    } else {
      RC_HIP(h, hipEventRecord(h->ev_side[1], st));               // secondary rays are in place
      RC_HIP(h, hipStreamWaitEvent(side, h->ev_side[1], 0));
      rc_launch_envmap(ea, side);
      RC_HIP(h, hipEventRecord(h->ev_side[2], side));
      env_released = true;
    }
    h->ws_prefix = "s:";
    enqueue_all(h, B, st);
    h->ws_prefix = "";
    if (!env_released) {
      // the trace took a launch plan without that spot (per-stage profiling on, a grid layout the level kernels do not
      // cover): the EnvMap behind the trace
      RC_HIP(h, hipEventRecord(h->ev_side[1], st));
      RC_HIP(h, hipStreamWaitEvent(side, h->ev_side[1], 0));
      rc_launch_envmap(ea, side);
      RC_HIP(h, hipEventRecord(h->ev_side[2], side));
    }
    RC_HIP(h, hipStreamWaitEvent(st, h->ev_side[2], 0));          // join: everything of this call is ordered on st again
    side_join.forked = false;
  }
  // 7. Monte-Carlo BRDF integration + MaterialIntegrator composite
  roctx_stage("material: integrate");
  {
    RcMatIntegrateArgs ia{};
    ia.n = n; ia.Ks = Ks; ia.Kd = Kd; ia.S = S2;
    ia.mat = W(h, "m_mat"); ia.samples = W(h, "sec_samples"); ia.local_view = W(h, "m_local_view");
    ia.sec_rgb = W(h, "sec_rgb"); ia.sec_acc = W(h, "sec_acc"); ia.sec_env = W(h, "sec_env");
    ia.weights = W(h, "weights" + LL); ia.filt_weight = W(h, "filt_weight");
    ia.pts = W(h, "m_pts"); ia.nrm = W(h, "m_nrm"); ia.origins = rays->origins; ia.lights = rays->lights;
    ia.f0 = c.default_F_0; ia.rgb_max = c.rgb_max; ia.bg = c.bg_intensity;
    ia.out = *mat_out;
    rc_launch_material_integrate(ia, st);
  }
  RC_HIP(h, hipGetLastError());
  return RC_OK;
  RC_CATCH(h)
}

}  // extern "C"

#include "rc_transient_host.inc"
#include "rc_train_host.inc"
